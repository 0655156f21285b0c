"""Loads libzbra_hip.so (the HIP kernels + C ABI) and declares its entry points.

There is no fallback: if the shared library is missing, importing this module raises.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZBRA_HIP_LIB") or os.path.join(_HERE, "libzbra_hip.so")  # override: tuning builds


class DevResult(C.Structure):
    """zbra_dev_result of include/zbra_hip.h."""
    _fields_ = [("status", C.c_uint32), ("width", C.c_uint32), ("out_count", C.c_uint64),
                ("midpoint", C.c_int64), ("packed_len", C.c_uint64)]


class ArrayDesc(C.Structure):
    """zbra_array_desc of include/zbra_hip.h."""
    _fields_ = [("data", C.c_void_p), ("len", C.c_uint64), ("count", C.c_uint64), ("out", C.c_void_p),
                ("kind", C.c_uint32), ("reserved", C.c_uint32)]


class EncodeDesc(C.Structure):
    """zbra_encode_desc of include/zbra_hip.h."""
    _fields_ = [("values", C.c_void_p), ("n", C.c_uint64), ("out", C.c_void_p), ("cap", C.c_uint64),
                ("kind", C.c_uint32), ("reserved", C.c_uint32)]


class EnumDesc(C.Structure):
    """zbra_enum_desc of include/zbra_hip.h."""
    _fields_ = [("tags", C.c_void_p), ("n", C.c_uint64), ("variant", C.c_void_p), ("rank", C.c_void_p), ("counts", C.c_void_p)]


class FileArray(C.Structure):
    """zbra_file_array of include/zbra_hip.h."""
    _fields_ = [("offset", C.c_uint64), ("len", C.c_uint64), ("count", C.c_uint64), ("kind", C.c_uint32),
                ("role", C.c_uint32), ("block", C.c_uint32), ("reserved", C.c_uint32)]


class FileInfo(C.Structure):
    """zbra_file_info of include/zbra_hip.h."""
    _fields_ = [("schema_offset", C.c_uint64), ("schema_len", C.c_uint64), ("compression_offset", C.c_uint64),
                ("compression_len", C.c_uint64), ("blocks_offset", C.c_uint64), ("block_count", C.c_uint32),
                ("reserved", C.c_uint32), ("n_arrays", C.c_uint64), ("total_values", C.c_uint64),
                ("end_offset", C.c_uint64)]


class FileJob(C.Structure):
    """zbra_file_job of include/zbra_hip.h."""
    _fields_ = [("device", C.c_int32), ("status", C.c_int32), ("image", C.c_void_p), ("len", C.c_uint64),
                ("arrays", C.POINTER(FileArray)), ("n", C.c_uint64), ("outs", C.POINTER(C.c_void_p)), ("statuses", C.POINTER(C.c_int))]


_p, _sz = C.c_void_p, C.c_size_t
_psz = C.POINTER(C.c_size_t)

# name -> (restype, argtypes); one line per declaration of include/zbra_hip.h
SIGNATURES = {
    "zbra_hip_version": (C.c_char_p, []),
    "zbra_hip_strerror": (C.c_char_p, [C.c_int]),
    "zbra_hip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "zbra_hip_set_device": (C.c_int, [C.c_int]),
    "zbra_hip_init": (C.c_int, []),
    "zbra_hip_encode_bound": (_sz, [_sz]),
    "zbra_hip_decoded_count": (C.c_int, [_p, _sz, _sz, _psz]),
    "zbra_hip_encode_i64": (C.c_int, [_p, _sz, _p, _sz, _psz]),
    "zbra_hip_encode_u32": (C.c_int, [_p, _sz, _p, _sz, _psz]),
    "zbra_hip_decode_i64": (C.c_int, [_p, _sz, _sz, _p, _psz]),
    "zbra_hip_decode_u32": (C.c_int, [_p, _sz, _sz, _p, _psz]),
    "zbra_hip_frame_of_reference_encode": (C.c_int, [_p, _sz, C.POINTER(C.c_int64), _p]),
    "zbra_hip_frame_of_reference_decode": (C.c_int, [C.c_int64, _p, _sz, _p]),
    "zbra_hip_zig_zag_encode": (C.c_int, [_p, _sz, _p]),
    "zbra_hip_zig_zag_decode": (C.c_int, [_p, _sz, _p]),
    "zbra_hip_bp64_pack": (C.c_int, [_p, _sz, _p, _sz, _psz]),
    "zbra_hip_bp64_unpack": (C.c_int, [_p, _sz, _sz, _p, _psz]),
    "zbra_hip_decode_dev": (C.c_int, [_p, _sz, _sz, _p, C.c_uint32, _p, _p]),
    "zbra_hip_decode_offsets_dev": (C.c_int, [_p, _sz, _sz, _p, _p, _p]),
    "zbra_hip_enum_rows_dev": (C.c_int, [_p, _sz, _p, C.c_uint32, _p, _p, _p, _p, _p]),
    "zbra_hip_decode_offsets_batch_dev": (C.c_int, [C.POINTER(ArrayDesc), _sz, _p, _p]),
    "zbra_hip_enum_rows_batch_dev": (C.c_int, [C.POINTER(EnumDesc), _sz, _p, C.c_uint32, _p, _p]),
    "zbra_hip_decode_batch_dev": (C.c_int, [C.POINTER(ArrayDesc), _sz, _p, _p]),
    "zbra_hip_decode_batch_dev_noalloc": (C.c_int, [C.POINTER(ArrayDesc), _sz, _p, _p]),
    "zbra_hip_encode_dev": (C.c_int, [_p, _sz, C.c_uint32, _p, _sz, _p, _p]),
    "zbra_hip_encode_batch_dev": (C.c_int, [C.POINTER(EncodeDesc), _sz, _p, _p]),
    "zbra_hip_decode_batch": (C.c_int, [C.POINTER(ArrayDesc), _sz, C.POINTER(C.c_int)]),
    "zbra_hip_encode_frames_dev": (C.c_int, [C.POINTER(EncodeDesc), _sz, _p, _sz, _p, _p, _p]),
    "zbra_hip_encode_frames": (C.c_int, [C.POINTER(EncodeDesc), _sz, _p, _sz, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "zbra_hip_file_index": (C.c_int, [_p, _sz, C.POINTER(FileArray), _sz, C.POINTER(FileInfo), _psz]),
    "zbra_hip_file_decode_multi": (C.c_int, [C.POINTER(FileJob), _sz]),
    "zbra_hip_file_decode": (C.c_int, [_p, _sz, C.POINTER(FileArray), _sz, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C zbra_amd/csrc` (or __graft_entry__.build()). "
            "zbra_amd has no CPU fallback.")
    # When torch is (or will be) in the process, its bundled HIP runtime must be the one both use,
    # or device pointers made by torch are foreign to this library: load torch's first.
    if "torch" not in sys.modules and os.environ.get("ZBRA_HIP_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()

OK = 0
ERR_SHORT_HEADER = 1
ERR_SHORT_PACKED = 2
ERR_TRUNCATED = 3
ERR_CAPACITY = 4
ERR_INVALID_ARG = 5
ERR_NO_DEVICE = 6
ERR_FORMAT = 7
ERR_UNKNOWN_TAG = 8
ERR_NOALLOC_LIMIT = 9
NOALLOC_MAX_ARRAYS = 4096
NOALLOC_MAX_TILES = 1 << 20
ERR_HIP_BASE = 100
ELEM_I64 = 0
ELEM_U32 = 1


def strerror(status: int) -> str:
    return lib.zbra_hip_strerror(status).decode()
