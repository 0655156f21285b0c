"""ctypes binding of oracle/libzbra_oracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py as the checker / the timed CPU port — never by the
shipped package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzbra_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    p = C.c_void_p
    lib.zo_frame_of_reference_encode.argtypes = [p, C.c_size_t, C.POINTER(C.c_int64), p]
    lib.zo_frame_of_reference_encode.restype = C.c_int
    lib.zo_frame_of_reference_decode.argtypes = [C.c_int64, p, C.c_size_t, p]
    lib.zo_frame_of_reference_decode.restype = None
    lib.zo_zig_zag_encode.argtypes = [p, C.c_size_t, p]
    lib.zo_zig_zag_encode.restype = None
    lib.zo_zig_zag_decode.argtypes = [p, C.c_size_t, p]
    lib.zo_zig_zag_decode.restype = None
    lib.zo_bp64_pack.argtypes = [p, C.c_size_t, p]
    lib.zo_bp64_pack.restype = C.c_size_t
    lib.zo_bp64_unpack.argtypes = [p, C.c_size_t, C.c_size_t, p, C.POINTER(C.c_size_t)]
    lib.zo_bp64_unpack.restype = C.c_int
    lib.zo_compress_bound.argtypes = [C.c_size_t]
    lib.zo_compress_bound.restype = C.c_size_t
    lib.zo_compress_int_array.argtypes = [p, C.c_size_t, p]
    lib.zo_compress_int_array.restype = C.c_size_t
    lib.zo_decompress_int_array.argtypes = [p, C.c_size_t, C.c_size_t, p, C.POINTER(C.c_size_t)]
    lib.zo_decompress_int_array.restype = C.c_int
    lib.zo_write_int_array_compressed.argtypes = [p, C.c_size_t, p]
    lib.zo_write_int_array_compressed.restype = C.c_size_t
    lib.zo_sm.argtypes = [C.c_uint64, C.c_uint64]
    lib.zo_sm.restype = C.c_uint64
    lib.zo_gen_narrow.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, p]
    lib.zo_gen_narrow.restype = None
    return lib


lib = _load()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def frame_of_reference_encode(values):
    v = np.ascontiguousarray(values, dtype=np.int64)
    d = np.empty_like(v)
    mid = C.c_int64(0)
    rc = lib.zo_frame_of_reference_encode(_ptr(v), v.size, C.byref(mid), _ptr(d))
    assert rc == 0
    return mid.value, d


def frame_of_reference_decode(midpoint, deltas):
    d = np.ascontiguousarray(deltas, dtype=np.int64)
    out = np.empty_like(d)
    lib.zo_frame_of_reference_decode(midpoint, _ptr(d), d.size, _ptr(out))
    return out


def zig_zag_encode(values):
    v = np.ascontiguousarray(values, dtype=np.int64)
    out = np.empty(v.size, dtype=np.uint64)
    lib.zo_zig_zag_encode(_ptr(v), v.size, _ptr(out))
    return out


def zig_zag_decode(values):
    u = np.ascontiguousarray(values, dtype=np.uint64)
    out = np.empty(u.size, dtype=np.int64)
    lib.zo_zig_zag_decode(_ptr(u), u.size, _ptr(out))
    return out


def bp64_pack(values) -> bytes:
    u = np.ascontiguousarray(values, dtype=np.uint64)
    out = np.empty(1 + 8 * u.size, dtype=np.uint8)
    n = lib.zo_bp64_pack(_ptr(u), u.size, _ptr(out))
    return out[:n].tobytes()


def bp64_unpack(packed: bytes, count: int):
    buf = np.frombuffer(bytes(packed), dtype=np.uint8)
    out = np.empty(max(count, 1), dtype=np.uint64)
    got = C.c_size_t(0)
    rc = lib.zo_bp64_unpack(_ptr(buf) if buf.size else None, buf.size, count, _ptr(out), C.byref(got))
    return rc, out[: got.value].copy()


def compress_int_array(values) -> bytes:
    v = np.ascontiguousarray(values, dtype=np.int64)
    out = np.empty(lib.zo_compress_bound(v.size) or 1, dtype=np.uint8)
    n = lib.zo_compress_int_array(_ptr(v), v.size, _ptr(out))
    assert n != C.c_size_t(-1).value
    return out[:n].tobytes()


def compress_int_array_into(values: np.ndarray, out: np.ndarray) -> int:
    """No-copy variant for timing: `out` is a uint8 array of >= compress_bound bytes."""
    return lib.zo_compress_int_array(_ptr(values), values.size, _ptr(out))


def decompress_int_array(data: bytes, count: int):
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    out = np.empty(max(count, 1), dtype=np.int64)
    got = C.c_size_t(0)
    rc = lib.zo_decompress_int_array(_ptr(buf) if buf.size else None, buf.size, count, _ptr(out), C.byref(got))
    return rc, out[: got.value].copy()


def decompress_int_array_into(data: np.ndarray, count: int, out: np.ndarray):
    """No-copy variant for timing."""
    got = C.c_size_t(0)
    rc = lib.zo_decompress_int_array(_ptr(data), data.size, count, _ptr(out), C.byref(got))
    return rc, got.value


def write_int_array_compressed(values) -> bytes:
    v = np.ascontiguousarray(values, dtype=np.int64)
    out = np.empty(8 + (lib.zo_compress_bound(v.size) or 1), dtype=np.uint8)
    n = lib.zo_write_int_array_compressed(_ptr(v), v.size, _ptr(out))
    return out[:n].tobytes()


def sm(seed: int, i: int) -> int:
    return lib.zo_sm(seed, i)


def gen_narrow(seed: int, first: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.int64)
    lib.zo_gen_narrow(seed, first, n, _ptr(out))
    return out


def compress_int_array_stable_sort_into(values: np.ndarray, out: np.ndarray) -> int:
    """The timed CPU encode baseline: the same stages with the median's sort done by an
    inlined-compare stable merge sort, as Rust's Vec::sort is (oracle/zbra_baseline_sort.cpp).
    Same bytes as compress_int_array_into.  Loads libzbra_baseline.so on first use."""
    global _baseline
    if _baseline is None:
        so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libzbra_baseline.so")
        if not os.path.exists(so):
            raise ImportError(f"{so} is missing: run `make -C oracle`")
        _baseline = C.CDLL(so)
        _baseline.zb_compress_int_array_stable_sort.restype = C.c_size_t
        _baseline.zb_compress_int_array_stable_sort.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    return _baseline.zb_compress_int_array_stable_sort(_ptr(values), values.size, _ptr(out))


_baseline = None
