"""Narrow submissions are decoded chunk by chunk behind a warm-up of each chunk's packed bytes, by
the LDS-free form of the tile kernel (plain loads, each lane reading the dwords that hold its own
values).  By default only submissions of 8 MiB or more whose arrays all have at most 6 bits per value take that path;
these tests send every packed array down it (any width, any size, tiny chunks) and compare with the
oracle, then check the default thresholds on a submission that is large enough."""
import ctypes as C

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
    f = _lib.lib.zbra_tuning_set_decode_warm
    f.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
    f.restype = C.c_int
    return torch, device, oc, _lib


@pytest.fixture()
def warm_everything(env):
    _, _, _, L = env
    L.lib.zbra_tuning_set_decode_warm(64 << 10, 31, 0)  # 64 KiB chunks, every packed width, any size
    yield
    L.lib.zbra_tuning_set_decode_warm(64 << 20, 6, 8 << 20)  # the defaults


def _dev_bytes(torch, b, pad):
    buf = torch.zeros(pad + len(b) + 64, dtype=torch.uint8, device="cuda")
    if len(b):
        buf[pad:pad + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
    return buf[pad:pad + len(b)] if len(b) else buf[:0]


@pytest.mark.parametrize("kind", KINDS)
def test_single_arrays_any_width(env, warm_everything, kind):
    torch, device, oc, L = env
    for n in (8, 9, 2047, 2049, 50001, 300007):
        v = column(kind, n, seed=17 + n)
        e = oc.compress_int_array(v)
        for pad in (0, 1, 3, 13):
            d_e = _dev_bytes(torch, e, pad)
            out = torch.full((n + 2,), -7, dtype=torch.int64, device="cuda")
            res = device.new_results(1, "cuda")
            device.decode_dev(d_e, n, out[1:], res, length=len(e))
            r = device.read_results(res)[0]
            assert r.status == 0 and r.out_count == n and r.width == e[12]
            got = out.cpu().numpy()
            assert np.array_equal(got[1:n + 1], v), (kind, n, pad)
            assert got[0] == -7 and got[n + 1] == -7, "wrote outside the output"


def test_every_width_i64_and_u32(env, warm_everything):
    torch, device, oc, L = env
    rng = np.random.default_rng(4)
    for w in range(1, 32):
        n = int(rng.integers(5000, 40000))
        z = rng.integers(0, 1 << w, n, dtype=np.uint64)
        z[0] = (1 << w) - 1                                  # the width is exactly w
        d = (z >> np.uint64(1)).astype(np.int64) ^ -(z & np.uint64(1)).astype(np.int64)
        d[1] = 0
        v = d - int(np.sort(d)[(n - 1) // 2]) + 1000         # any column will do: the oracle says what its bytes are
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
        assert np.array_equal(got[3:n + 3], want.astype(np.int32)) and (got[:3] == -1).all() and (got[n + 3:] == -1).all(), w


def test_batch_with_tags_empty_and_rejected(env, warm_everything):
    torch, device, oc, L = env
    rng = np.random.default_rng(8)
    specs = [("binary", 70000), ("small_signed", 33333), ("constant", 9000), ("narrow", 50000), ("small_signed", 0), ("binary", 7),
             ("w31", 4100), ("small_signed", 123456)]
    cols = [column(k, n, seed=i) for i, (k, n) in enumerate(specs)]
    kinds = [L.ELEM_I64] * len(cols)
    tags = rng.choice(np.array([0, 1, 2, 7], dtype=np.uint32), 200001, p=[0.5, 0.25, 0.15, 0.1])
    cols.append(tags); kinds.append(L.ELEM_U32)
    encs = [oc.compress_int_array(c.astype(np.int64)) for c in cols]
    bad = 3                                                   # its payload is cut short: ZBRA_ERR_TRUNCATED, neighbours untouched
    blob, offs = bytearray(), []
    for e in encs:
        blob += b"\xAA" * int(rng.integers(0, 5))
        offs.append(len(blob))
        blob += e
    d_blob = _dev_bytes(torch, bytes(blob), 0)
    outs, items = [], []
    for i, (v, e) in enumerate(zip(cols, encs)):
        o = torch.full((v.size + 2,), -3, dtype=torch.int64 if kinds[i] == L.ELEM_I64 else torch.int32, device="cuda")
        outs.append(o)
        e_len = len(e)
        if i == bad:
            cut = bytearray(e[:len(e) - 40]); cut[8:12] = (len(cut) - 12).to_bytes(4, "little")
            d_blob[offs[i]:offs[i] + len(cut)] = torch.frombuffer(cut, dtype=torch.uint8).cuda()
            e_len = len(cut)
        items.append((d_blob.data_ptr() + offs[i], e_len, v.size, o.data_ptr() + o.element_size(), kinds[i]))
    res = device.new_results(len(items), "cuda")
    device.decode_batch_dev(device.make_descs(items), res)
    rs = device.read_results(res)
    for i, v in enumerate(cols):
        got = outs[i].cpu().numpy()
        if i == bad:
            assert rs[i].status == L.ERR_TRUNCATED and (got == -3).all()
            continue
        assert rs[i].status == L.OK and rs[i].out_count == v.size, i
        expect = v if kinds[i] == L.ELEM_I64 else v.view(np.int32)
        assert np.array_equal(got[1:v.size + 1], expect) and got[0] == -3 and got[-1] == -3, i


def test_default_thresholds_on_a_large_narrow_column(env):
    """2^25 tags of width 4 (16 MiB packed): takes the warm-up path with the default settings."""
    torch, device, oc, L = env
    n = 1 << 25
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    v = torch.randint(0, 7, (n,), dtype=torch.int64, device="cuda", generator=g)
    enc = torch.empty(device.encode_bound(n), dtype=torch.uint8, device="cuda")
    res = device.new_results(1, "cuda")
    device.encode_dev(v, enc, res)
    r = device.read_results(res)[0]
    assert r.status == 0 and r.width <= 4 and r.out_count >= 8 << 20
    out = torch.empty_like(v)
    device.decode_dev(enc, n, out, None, length=int(r.out_count))
    assert torch.equal(out, v)
    out32 = torch.empty(n, dtype=torch.int32, device="cuda")
    device.decode_dev(enc, n, out32, None, length=int(r.out_count))
    assert torch.equal(out32.to(torch.int64), v)


def test_mixed_batch_puts_narrow_arrays_first(env):
    """Narrow and wide arrays in one submission: the device sees the narrow ones first (their tiles
    take the warm-up route, the others the default kernel); results and status words must still
    land at the caller's indices."""
    torch, device, oc, L = env
    L.lib.zbra_tuning_set_decode_warm(64 << 10, 6, 0)  # the default width limit, any size
    try:
        rng = np.random.default_rng(12)
        specs = [("w31", 5000), ("binary", 70000), ("full", 4097), ("small_signed", 33333), ("narrow", 50000), ("constant", 9000),
                 ("small_signed", 0), ("time_series", 6143), ("binary", 7), ("w32", 2049), ("small_signed", 123456), ("clustered", 2047)]
        cols = [column(k, n, seed=40 + i) for i, (k, n) in enumerate(specs)]
        kinds = [L.ELEM_I64] * len(cols)
        for n_tags, vals in ((200001, [0, 1]), (70001, [0, 1, 2, 7])):     # u32: width 1-2 is narrow, width 4 is not
            cols.append(rng.choice(np.array(vals, dtype=np.uint32), n_tags)); kinds.append(L.ELEM_U32)
        encs = [oc.compress_int_array(c.astype(np.int64)) for c in cols]
        widths = [e[12] if len(e) > 12 else None for e in encs]
        assert any(w is not None and w <= 6 for w in widths) and any(w is not None and w > 6 for w in widths)
        bad = {1: L.ERR_SHORT_PACKED, 4: L.ERR_TRUNCATED}                   # a narrow and a wide array rejected
        blob, offs, lens = bytearray(), [], []
        for i, e in enumerate(encs):
            blob += b"\\xAA" * int(rng.integers(0, 5))
            offs.append(len(blob))
            if bad.get(i) == L.ERR_TRUNCATED:
                cut = bytearray(e[:len(e) - 40]); cut[8:12] = (len(cut) - 12).to_bytes(4, "little"); e = bytes(cut)
            blob += e
            lens.append(len(e) - 1 if bad.get(i) == L.ERR_SHORT_PACKED else len(e))
        d_blob = _dev_bytes(torch, bytes(blob), 3)
        outs, items = [], []
        for i, v in enumerate(cols):
            o = torch.full((v.size + 2,), -3, dtype=torch.int64 if kinds[i] == L.ELEM_I64 else torch.int32, device="cuda")
            outs.append(o)
            items.append((d_blob.data_ptr() + offs[i], lens[i], v.size, o.data_ptr() + o.element_size(), kinds[i]))
        res = device.new_results(len(items), "cuda")
        device.decode_batch_dev(device.make_descs(items), res)
        rs = device.read_results(res)
        for i, v in enumerate(cols):
            got = outs[i].cpu().numpy()
            if i in bad:
                assert rs[i].status == bad[i] and (got == -3).all(), i
                continue
            assert rs[i].status == L.OK and rs[i].out_count == v.size, i
            if v.size:
                assert rs[i].width == widths[i], i
            expect = v if kinds[i] == L.ELEM_I64 else v.view(np.int32)
            assert np.array_equal(got[1:v.size + 1], expect) and got[0] == -3 and got[-1] == -3, (i, specs[i] if i < len(specs) else "tags")
    finally:
        L.lib.zbra_tuning_set_decode_warm(64 << 20, 6, 8 << 20)
