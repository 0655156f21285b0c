"""Host-side mirror of `zbra_core::compression` (zbra-core/src/compression.rs:43-330).

Same eight public names, same argument meaning, same results and error behaviour as the Rust
module, with numpy arrays for slices/Vecs and `bytes` for `Vec<u8>`.  Every function runs the HIP
kernels of libzbra_hip.so through its C ABI (include/zbra_hip.h); nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib


class BinaryError(Exception):
    """zbra_core::error::BinaryError (zbra-core/src/error.rs)."""


class DecompressionError(BinaryError):
    """BinaryError::DecompressionError(String): the only error this path raises
    (compression.rs:301-303, 313-315)."""


class DeviceError(BinaryError):
    """HIP runtime failure or unusable device: no counterpart in the reference (it has no device)."""


def _raise(status: int):
    msg = _lib.strerror(status)
    if status in (_lib.ERR_SHORT_HEADER, _lib.ERR_SHORT_PACKED, _lib.ERR_TRUNCATED):
        raise DecompressionError(msg)
    if status in (_lib.ERR_INVALID_ARG, _lib.ERR_NOALLOC_LIMIT):
        raise ValueError(msg)
    raise DeviceError(f"{msg} (status {status})")


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p) if a.size else None


def _i64(values) -> np.ndarray:
    return np.ascontiguousarray(values, dtype=np.int64)


def _u64(values) -> np.ndarray:
    return np.ascontiguousarray(values, dtype=np.uint64)


def _bytes_arr(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def frame_of_reference_encode(values):
    """frame_of_reference_encode(values: &[i64]) -> (i64, Vec<i64>)   compression.rs:52-74"""
    v = _i64(values)
    deltas = np.empty_like(v)
    mid = C.c_int64(0)
    st = lib.zbra_hip_frame_of_reference_encode(_ptr(v), v.size, C.byref(mid), _ptr(deltas))
    if st:
        _raise(st)
    return mid.value, deltas


def frame_of_reference_decode(midpoint: int, deltas):
    """frame_of_reference_decode(midpoint, deltas) -> Vec<i64>          compression.rs:77-82"""
    d = _i64(deltas)
    out = np.empty_like(d)
    st = lib.zbra_hip_frame_of_reference_decode(midpoint, _ptr(d), d.size, _ptr(out))
    if st:
        _raise(st)
    return out


def zig_zag_encode(values):
    """zig_zag_encode(values: &[i64]) -> Vec<u64>                       compression.rs:88-98"""
    v = _i64(values)
    out = np.empty(v.size, dtype=np.uint64)
    st = lib.zbra_hip_zig_zag_encode(_ptr(v), v.size, _ptr(out))
    if st:
        _raise(st)
    return out


def zig_zag_decode(values):
    """zig_zag_decode(values: &[u64]) -> Vec<i64>                       compression.rs:101-111"""
    u = _u64(values)
    out = np.empty(u.size, dtype=np.int64)
    st = lib.zbra_hip_zig_zag_decode(_ptr(u), u.size, _ptr(out))
    if st:
        _raise(st)
    return out


def bp64_pack(values) -> bytes:
    """bp64_pack(values: &[u64]) -> Result<Vec<u8>>                     compression.rs:117-178"""
    u = _u64(values)
    out = np.empty(1 + 8 * u.size, dtype=np.uint8)
    n = C.c_size_t(0)
    st = lib.zbra_hip_bp64_pack(_ptr(u), u.size, _ptr(out), out.size, C.byref(n))
    if st:
        _raise(st)
    return out[: n.value].tobytes()


def bp64_unpack(packed, count: int):
    """bp64_unpack(packed: &[u8], count) -> Result<Vec<u64>>            compression.rs:181-233"""
    p = _bytes_arr(packed)
    out = np.empty(count, dtype=np.uint64)
    n = C.c_size_t(0)
    st = lib.zbra_hip_bp64_unpack(_ptr(p), p.size, count, _ptr(out), C.byref(n))
    if st:
        _raise(st)
    return out[: n.value]


def compress_int_array(values) -> bytes:
    """compress_int_array(values: &[i64]) -> Result<Vec<u8>>            compression.rs:271-292"""
    v = _i64(values)
    out = np.empty(lib.zbra_hip_encode_bound(v.size), dtype=np.uint8)
    n = C.c_size_t(0)
    st = lib.zbra_hip_encode_i64(_ptr(v), v.size, _ptr(out), out.size, C.byref(n))
    if st:
        _raise(st)
    return out[: n.value].tobytes()


def decoded_count(data, count: int) -> int:
    """Length of the Vec decompress_int_array(data, count) returns (zbra_hip_decoded_count); raises
    what the decode would raise.  Host only."""
    d = _bytes_arr(data)
    n = C.c_size_t(0)
    st = lib.zbra_hip_decoded_count(_ptr(d), d.size, count, C.byref(n))
    if st:
        _raise(st)
    return n.value


def decompress_int_array(data, count: int):
    """decompress_int_array(data: &[u8], count) -> Result<Vec<i64>>     compression.rs:295-330"""
    d = _bytes_arr(data)
    out = np.empty(count, dtype=np.int64)
    n = C.c_size_t(0)
    st = lib.zbra_hip_decode_i64(_ptr(d), d.size, count, _ptr(out), C.byref(n))
    if st:
        _raise(st)
    return out[: n.value]


def compress_u32_array(values) -> bytes:
    """write_u32_array_compressed's widen `as i64` + compress_int_array   binary.rs:666-669"""
    v = np.ascontiguousarray(values, dtype=np.uint32)
    out = np.empty(lib.zbra_hip_encode_bound(v.size), dtype=np.uint8)
    n = C.c_size_t(0)
    st = lib.zbra_hip_encode_u32(_ptr(v), v.size, _ptr(out), out.size, C.byref(n))
    if st:
        _raise(st)
    return out[: n.value].tobytes()


def decompress_u32_array(data, count: int):
    """read_u32_array_compressed's decompress + narrow `as u32`           binary.rs:671-674"""
    d = _bytes_arr(data)
    out = np.empty(count, dtype=np.uint32)
    n = C.c_size_t(0)
    st = lib.zbra_hip_decode_u32(_ptr(d), d.size, count, _ptr(out), C.byref(n))
    if st:
        _raise(st)
    return out[: n.value]


def write_int_arrays_compressed(arrays, as_bytes=True):
    """write_int_array_compressed / write_u32_array_compressed (binary.rs:639-645, 666-669) for a
    list of arrays in one submission: returns (frames, offsets) where frames[offsets[i]:offsets[i+1]]
    is `u32 count | u32 clen | E` of arrays[i].  uint32 arrays go the enum-tag way, everything else
    is taken as int64.  as_bytes=False returns the frames as a uint8 array (no copy)."""
    from ._lib import ELEM_I64, ELEM_U32, EncodeDesc
    vs = [np.ascontiguousarray(a, dtype=np.uint32) if getattr(a, "dtype", None) == np.uint32 else _i64(a) for a in arrays]
    n = len(vs)
    descs = (EncodeDesc * max(1, n))()
    bound = 0
    for i, v in enumerate(vs):
        descs[i].values = v.ctypes.data if v.size else None
        descs[i].n = v.size
        descs[i].kind = ELEM_U32 if v.dtype == np.uint32 else ELEM_I64
        bound += 8 + lib.zbra_hip_encode_bound(v.size)
    out = np.empty(bound, dtype=np.uint8)
    offsets = (C.c_uint64 * (n + 1))()
    statuses = (C.c_int * max(1, n))()
    st = lib.zbra_hip_encode_frames(descs, n, _ptr(out) if bound else None, bound, offsets, statuses)
    if st:
        _raise(st)
    offs = np.frombuffer(offsets, dtype=np.uint64).astype(np.int64)
    return (out[: offs[n]].tobytes() if as_bytes else out[: offs[n]]), offs


def read_int_arrays_compressed(arrays):
    """read_int_array_compressed / read_u32_array_compressed (binary.rs:648-654, 671-674) for a
    list of (compressed bytes, count, dtype) in one submission: the compressed arrays sit in
    separate host buffers, as after the reference's read_exact per array.  dtype is np.int64 or
    np.uint32 (enum tags).  Returns the list of decoded numpy arrays; raises on the first array
    the reference would reject."""
    from ._lib import ELEM_I64, ELEM_U32, ArrayDesc
    n = len(arrays)
    descs = (ArrayDesc * max(1, n))()
    keep, outs = [], []
    for i, (data, count, dtype) in enumerate(arrays):
        d = _bytes_arr(data)
        out = np.empty(count, dtype=np.uint32 if np.dtype(dtype) == np.uint32 else np.int64)
        keep.append(d)
        outs.append(out)
        descs[i].data = d.ctypes.data if d.size else None
        descs[i].len = d.size
        descs[i].count = count
        descs[i].out = out.ctypes.data if out.size else None
        descs[i].kind = ELEM_U32 if out.dtype == np.uint32 else ELEM_I64
    statuses = (C.c_int * max(1, n))()
    st = lib.zbra_hip_decode_batch(descs, n, statuses)
    if st:
        _raise(st)
    # decompress_int_array returns an empty Vec for empty data and for packed_len == 0 whatever
    # count says (compression.rs:296-298, :182-184): the device wrote nothing there
    return [o[:decoded_count(k, o.size)] for o, k in zip(outs, keep)]
