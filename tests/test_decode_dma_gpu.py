"""The experimental decode shape — two tiles per workgroup, packed bytes brought into LDS by
global_load_lds_dwordx4 with hand-counted waits — is not the default, but it is in the library and
selectable, so it is held to the same bar: every width, every size class incl. ragged last tiles
and odd tile counts, every byte alignment, int64 and u32 outputs, batches that mix packed, raw,
fill, empty and rejected arrays; all against the oracle, with guard words around every output."""
import struct

import numpy as np
import pytest

from conftest import KINDS, column

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    from oracle import zbra_oracle_c as oc
    from zbra_amd import _lib, device
    return torch, device, oc, _lib


@pytest.fixture(params=[(3, 2), (2, 0), (3, 1)], ids=["nt-by-width", "plain-loads", "nt-chunked"])
def dma_shape(env, request):
    _, _, _, L = env
    nt, order = request.param
    assert L.lib.zbra_tuning_set_decode_variant(2048, 512, nt, 2, order) == 0
    yield
    assert L.lib.zbra_tuning_set_decode_variant(2048, 512, 3, 0, 2) == 0  # the default


def _dev_bytes(torch, b, pad):
    buf = torch.zeros(pad + len(b) + 64, dtype=torch.uint8, device="cuda")
    if len(b):
        buf[pad:pad + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
    return buf[pad:pad + len(b)] if len(b) else buf[:0]


@pytest.mark.parametrize("kind", KINDS)
def test_single_arrays(env, dma_shape, kind):
    torch, device, oc, L = env
    # 1 tile, 2 tiles, 3 tiles (a workgroup with one tile only), ragged ends, many tiles
    for n in (1, 8, 2047, 2048, 2049, 4096, 4097, 6144, 50001, 300007):
        v = column(kind, n, seed=23 + n)
        e = oc.compress_int_array(v)
        for pad in (0, 1, 7, 13):
            d_e = _dev_bytes(torch, e, pad)
            out = torch.full((n + 2,), -7, dtype=torch.int64, device="cuda")
            res = device.new_results(1, "cuda")
            device.decode_dev(d_e, n, out[1:], res, length=len(e))
            r = device.read_results(res)[0]
            assert r.status == 0 and r.out_count == n and r.width == e[12]
            got = out.cpu().numpy()
            assert np.array_equal(got[1:n + 1], v), (kind, n, pad)
            assert got[0] == -7 and got[n + 1] == -7, "wrote outside the output"


def test_every_width_i64_and_u32(env, dma_shape):
    torch, device, oc, L = env
    rng = np.random.default_rng(14)
    for w in range(1, 32):
        n = int(rng.integers(5000, 40000))
        z = rng.integers(0, 1 << w, n, dtype=np.uint64)
        z[0] = (1 << w) - 1                                  # the width is exactly w
        d = (z >> np.uint64(1)).astype(np.int64) ^ -(z & np.uint64(1)).astype(np.int64)
        d[1] = 0
        v = d - int(np.sort(d)[(n - 1) // 2]) + 1000
        e = oc.compress_int_array(v)
        want = oc.decompress_int_array(e, n)[1]
        d_e = _dev_bytes(torch, e, 5)
        out = torch.full((n + 2,), -1, dtype=torch.int64, device="cuda")
        device.decode_dev(d_e, n, out[1:], None, length=len(e))
        got = out.cpu().numpy()
        assert np.array_equal(got[1:n + 1], want) and got[0] == -1 and got[n + 1] == -1, w
        out32 = torch.full((n + 5,), -1, dtype=torch.int32, device="cuda")
        device.decode_dev(d_e, n, out32[3:], None, length=len(e))   # 12 bytes past a 16-byte boundary
        got = out32.cpu().numpy()
        assert np.array_equal(got[3:n + 3].view(np.uint32), want.astype(np.uint64).astype(np.uint32)), w
        assert (got[:3] == -1).all() and (got[n + 3:] == -1).all(), w


def test_batch_mixing_every_kind_of_tile(env, dma_shape):
    """Pairs of tiles that belong to different arrays: packed next to raw, fill, empty, rejected; u32
    tiles (twice the values) next to int64 tiles; arrays of one tile, so that every pair straddles."""
    torch, device, oc, L = env
    rng = np.random.default_rng(8)
    specs = []
    for k in range(60):
        kind = KINDS[k % len(KINDS)]
        n = int(rng.choice([1, 700, 2048, 2049, 4096, 5000, 12289, 40000]))
        specs.append((kind, n, k % 7 == 3))            # every seventh array is a u32 tag array
    cols, encs, kinds = [], [], []
    for i, (kind, n, as_u32) in enumerate(specs):
        if as_u32:
            v = rng.choice(np.array([0, 1, 2, 7, 1000], dtype=np.uint32), n)
            cols.append(v); encs.append(oc.compress_int_array(v.astype(np.int64))); kinds.append(L.ELEM_U32)
        else:
            v = column(kind, n, seed=900 + i)
            cols.append(v); encs.append(oc.compress_int_array(v)); kinds.append(L.ELEM_I64)
    # crafted: width byte 0 (fill), empty data, a truncated packed array, a short raw payload
    fill = struct.pack("<qIB", -5, 1, 0)
    good = oc.compress_int_array(column("narrow", 3000, 1))
    raw = oc.compress_int_array(column("w40", 3000, 2))
    short_raw = bytearray(raw[:13 + 8 * 1000 + 3]); short_raw[8:12] = struct.pack("<I", len(short_raw) - 12)
    crafted = [(fill, 5000, 0), (b"", 77, 0), (good[:-1], 3000, L.ERR_SHORT_PACKED), (bytes(short_raw), 3000, 0)]
    blob, offs = bytearray(), []
    all_encs = encs + [c[0] for c in crafted]
    for e in all_encs:
        blob += b"\xAA" * int(rng.integers(0, 9))
        offs.append(len(blob))
        blob += e
    blob += b"\0" * 64
    d_blob = torch.frombuffer(blob, dtype=torch.uint8).cuda()
    counts = [c.size for c in cols] + [c[1] for c in crafted]
    kinds += [L.ELEM_I64] * len(crafted)
    outs = [torch.full((cnt + 2,), -3, dtype=torch.int64 if k == L.ELEM_I64 else torch.int32, device="cuda") for cnt, k in zip(counts, kinds)]
    descs = device.make_descs([(d_blob.data_ptr() + o if len(e) else 0, len(e), cnt, out.data_ptr() + out.element_size(), k)
                               for o, e, cnt, out, k in zip(offs, all_encs, counts, outs, kinds)])
    res = device.new_results(len(all_encs), "cuda")
    device.decode_batch_dev(descs, res)
    torch.cuda.synchronize()
    rs = device.read_results(res)
    for i, v in enumerate(cols):
        got = outs[i].cpu().numpy()
        want = v if v.dtype == np.int64 else v.view(np.int32)
        assert rs[i].status == 0 and np.array_equal(got[1:-1], want) and got[0] == -3 and got[-1] == -3, (i, specs[i])
    base = len(cols)
    for j, (e, cnt, status) in enumerate(crafted):
        got = outs[base + j].cpu().numpy()
        st, want = oc.decompress_int_array(e, cnt)
        assert rs[base + j].status == status == st, j
        assert got[0] == -3 and got[-1] == -3
        if status == 0 and want.size:
            assert np.array_equal(got[1:-1], want), j
        else:
            assert (got == -3).all(), j
