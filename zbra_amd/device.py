"""Device-resident entry points of libzbra_hip.so on torch tensors.

torch supplies device memory and the stream; every byte of codec work is done by the library's
HIP kernels.  Calls are stream-ordered on torch's current stream and do not synchronise.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ArrayDesc, DevResult, EncodeDesc, EnumDesc, lib
from .compression import _raise

RESULT_BYTES = C.sizeof(DevResult)  # 32


def _stream_handle(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def encode_bound(n: int) -> int:
    return lib.zbra_hip_encode_bound(n)


def new_results(n: int, device) -> torch.Tensor:
    """Device buffer for n zbra_dev_result records."""
    return torch.zeros(n * RESULT_BYTES, dtype=torch.uint8, device=device)


def read_results(results: torch.Tensor):
    """Copies result records to the host (synchronises) and returns a list of DevResult."""
    raw = results.cpu().numpy().tobytes()
    n = len(raw) // RESULT_BYTES
    return [DevResult.from_buffer_copy(raw, i * RESULT_BYTES) for i in range(n)]


def _kind_of(dtype) -> int:
    if dtype == torch.int64:
        return _lib.ELEM_I64
    if dtype in (torch.uint32, torch.int32):
        return _lib.ELEM_U32
    raise TypeError(f"unsupported element dtype {dtype}")


def decode_dev(encoded: torch.Tensor, count: int, out: torch.Tensor, result: torch.Tensor = None, length: int = None,
               stream=None):
    """decompress_int_array (compression.rs:295-330) on bytes resident in HBM.

    encoded: uint8 CUDA tensor holding E; out: int64 (or int32/uint32 for tag arrays) CUDA tensor
    of `count` elements; result: optional uint8 tensor of RESULT_BYTES bytes."""
    assert encoded.is_cuda and encoded.dtype == torch.uint8 and encoded.is_contiguous()
    assert out.is_cuda and out.is_contiguous() and out.numel() >= count
    n = encoded.numel() if length is None else length
    st = lib.zbra_hip_decode_dev(encoded.data_ptr(), n, count, out.data_ptr(), _kind_of(out.dtype),
                                 result.data_ptr() if result is not None else None, _stream_handle(stream))
    if st:
        _raise(st)


def decode_offsets_dev(encoded: torch.Tensor, count: int, offsets: torch.Tensor, result: torch.Tensor = None, length: int = None,
                       stream=None):
    """zbra_hip_decode_offsets_dev: a lengths array (binary.rs:661-664) decoded straight into its
    running sum; `offsets` is an int64/uint64 CUDA tensor of count + 1 words."""
    assert encoded.is_cuda and encoded.dtype == torch.uint8 and offsets.is_cuda and offsets.element_size() == 8
    assert offsets.numel() >= count + 1
    n = encoded.numel() if length is None else length
    st = lib.zbra_hip_decode_offsets_dev(encoded.data_ptr(), n, count, offsets.data_ptr(),
                                         result.data_ptr() if result is not None else None, _stream_handle(stream))
    if st:
        _raise(st)


def enum_rows_dev(tags: torch.Tensor, variant_tags, variant: torch.Tensor, rank: torch.Tensor, counts: torch.Tensor,
                  result: torch.Tensor = None, stream=None):
    """zbra_hip_enum_rows_dev: the row loop of Column::from_values for an Enum column
    (striped.rs:340-387) from the tag column alone.  tags: int32/uint32 CUDA tensor; variant_tags:
    host sequence of at most 32 tags (the schema's variants, in order); variant (4-byte), rank
    (8-byte) CUDA tensors of len(tags); counts: 8-byte CUDA tensor of len(variant_tags)."""
    import ctypes as C
    n = tags.numel()
    assert tags.is_cuda and tags.element_size() == 4 and variant.is_cuda and variant.element_size() == 4
    assert rank.is_cuda and rank.element_size() == 8 and variant.numel() >= n and rank.numel() >= n
    vt = (C.c_uint32 * max(1, len(variant_tags)))(*[int(t) & 0xFFFFFFFF for t in variant_tags])
    assert counts.is_cuda and counts.element_size() == 8 and counts.numel() >= len(variant_tags)
    st = lib.zbra_hip_enum_rows_dev(tags.data_ptr(), n, vt, len(variant_tags), variant.data_ptr(), rank.data_ptr(), counts.data_ptr(),
                                    result.data_ptr() if result is not None else None, _stream_handle(stream))
    if st:
        _raise(st)


def decode_offsets_batch_dev(descs, results: torch.Tensor = None, stream=None):
    """zbra_hip_decode_offsets_batch_dev: many lengths arrays, each decoded straight into its own
    running sum, in one submission.  descs from make_descs(): `out` of array i is the device address
    of count_i + 1 eight-byte words, kind ELEM_I64."""
    st = lib.zbra_hip_decode_offsets_batch_dev(descs, len(descs), results.data_ptr() if results is not None else None,
                                               _stream_handle(stream))
    if st:
        _raise(st)


def enum_rows_batch_dev(arrays, variant_tags, results: torch.Tensor = None, stream=None):
    """zbra_hip_enum_rows_batch_dev: the Enum row split of many tag arrays that share one variant
    list, in one submission.  arrays: sequence of (tags, variant, rank, counts) CUDA tensors as in
    enum_rows_dev."""
    arrays = list(arrays)
    descs = (EnumDesc * max(1, len(arrays)))()
    for i, (tags, variant, rank, counts) in enumerate(arrays):
        n = tags.numel()
        assert tags.is_cuda and tags.element_size() == 4 and variant.element_size() == 4 and rank.element_size() == 8
        assert variant.numel() >= n and rank.numel() >= n and counts.element_size() == 8 and counts.numel() >= len(variant_tags)
        descs[i].tags, descs[i].n = tags.data_ptr(), n
        descs[i].variant, descs[i].rank, descs[i].counts = variant.data_ptr(), rank.data_ptr(), counts.data_ptr()
    vt = (C.c_uint32 * max(1, len(variant_tags)))(*[int(t) & 0xFFFFFFFF for t in variant_tags])
    st = lib.zbra_hip_enum_rows_batch_dev(descs, len(arrays), vt, len(variant_tags), results.data_ptr() if results is not None else None,
                                          _stream_handle(stream))
    if st:
        _raise(st)


def make_descs(items):
    """items: iterable of (data_ptr, len, count, out_ptr, kind) -> ctypes array of zbra_array_desc."""
    items = list(items)
    arr = (ArrayDesc * len(items))()
    for i, (dp, ln, cnt, op, kind) in enumerate(items):
        arr[i].data = dp
        arr[i].len = ln
        arr[i].count = cnt
        arr[i].out = op
        arr[i].kind = kind
    return arr


def decode_batch_dev(descs, results: torch.Tensor = None, stream=None):
    """Decodes many independent arrays in one submission; descs from make_descs()."""
    st = lib.zbra_hip_decode_batch_dev(descs, len(descs), results.data_ptr() if results is not None else None,
                                       _stream_handle(stream))
    if st:
        _raise(st)


def decode_batch_dev_noalloc(descs, results: torch.Tensor = None, stream=None):
    """zbra_hip_decode_batch_dev_noalloc: decode_batch_dev for callers that cannot allocate (kernel
    launches only, fixed scratch inside the library; at most _lib.NOALLOC_MAX_ARRAYS arrays and
    _lib.NOALLOC_MAX_TILES tiles per call; calls that overlap on one GPU race)."""
    st = lib.zbra_hip_decode_batch_dev_noalloc(descs, len(descs), results.data_ptr() if results is not None else None,
                                               _stream_handle(stream))
    if st:
        _raise(st)


def encode_dev(values: torch.Tensor, out: torch.Tensor, result: torch.Tensor, stream=None):
    """compress_int_array (compression.rs:271-292) on values resident in HBM; the encoded length
    lands in result (out_count)."""
    assert values.is_cuda and values.is_contiguous() and out.is_cuda and out.dtype == torch.uint8
    st = lib.zbra_hip_encode_dev(values.data_ptr(), values.numel(), _kind_of(values.dtype), out.data_ptr(), out.numel(),
                                 result.data_ptr(), _stream_handle(stream))
    if st:
        _raise(st)


def make_encode_descs(items):
    """items: iterable of (values_ptr, n, out_ptr, cap, kind) -> ctypes array of zbra_encode_desc."""
    items = list(items)
    arr = (EncodeDesc * len(items))()
    for i, (vp, n, op, cap, kind) in enumerate(items):
        arr[i].values = vp
        arr[i].n = n
        arr[i].out = op
        arr[i].cap = cap
        arr[i].kind = kind
    return arr


def encode_frames_dev(descs, out: torch.Tensor, offsets: torch.Tensor, results=None, stream=None):
    """zbra_hip_encode_frames_dev: frames `u32 count | u32 clen | E` back to back in `out` (uint8);
    `offsets` is an int64/uint64 device tensor of len(descs) + 1 words."""
    assert offsets.numel() >= len(descs) + 1 and offsets.element_size() == 8
    st = lib.zbra_hip_encode_frames_dev(descs, len(descs), out.data_ptr(), out.numel(), offsets.data_ptr(),
                                        results.data_ptr() if results is not None else None, _stream_handle(stream))
    if st:
        _raise(st)


def encode_batch_dev(descs, results: torch.Tensor, stream=None):
    st = lib.zbra_hip_encode_batch_dev(descs, len(descs), results.data_ptr(), _stream_handle(stream))
    if st:
        _raise(st)
