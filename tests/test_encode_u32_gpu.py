"""Encode of u32 (enum tag) arrays, device-resident: write_u32_array_compressed widens the tags to
i64 and runs the same codec (binary.rs:666-669), so the bytes must equal the oracle's encoding of
the widened values.  A single u32 array and a batch of equally long u32 arrays are packed by a
kernel of their own (4096 tags per workgroup, 32-bit arithmetic, tiles numbered per array), which
this file drives through every width incl. the raw ones, lengths either side of its tile size,
4-byte-misaligned inputs, byte-misaligned outputs, capacity errors and framed output; ragged u32
batches and batches that mix element kinds must keep taking the generic kernel."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    from oracle import zbra_oracle_c as oc
    from zbra_amd import _lib, device
    return torch, device, oc, _lib


def _tags_of_width(rng, n, w):
    """n >= 3 u32 values whose encoding has bit width w (1..33).  More than half of them equal the
    median, so the midpoint is known; the extremes give the zig-zag maximum its bit length."""
    if w == 33:                                              # the median at one end of the u32 range
        v = np.zeros(n, dtype=np.uint32)
        v[::3] = rng.integers(0, 1 << 32, v[::3].size, dtype=np.uint64).astype(np.uint32)
        v[0] = (1 << 32) - 1                                 # delta 2^32 - 1 -> zig-zag 2^33 - 2
        v[1] = v[2] = 0
        return v
    base, reach = 1 << 31, 1 << (w - 1)
    v = np.full(n, base, dtype=np.int64)
    if w > 1:
        v[::3] = base + rng.integers(-reach + 1, reach, v[::3].size, dtype=np.int64)
    v[0] = base - reach                                      # zig-zag 2^w - 1
    v[1] = v[2] = base
    return v.astype(np.uint32)


def _encode_one(torch, device, L, tags, pad_in=0, pad_out=3, cap=None):
    n = tags.size
    buf = torch.zeros(n + pad_in + 8, dtype=torch.int32, device="cuda")
    buf[pad_in:pad_in + n] = torch.from_numpy(tags.view(np.int32)).cuda()
    d_v = buf[pad_in:pad_in + n]
    bound = device.encode_bound(n)
    cap = bound if cap is None else cap
    out = torch.full((pad_out + bound + 32,), 0xEE, dtype=torch.uint8, device="cuda")
    res = device.new_results(1, "cuda")
    device.encode_dev(d_v, out[pad_out:pad_out + cap], res)
    return device.read_results(res)[0], out.cpu().numpy()


def test_every_width_and_ragged_length(env):
    torch, device, oc, L = env
    rng = np.random.default_rng(31)
    for w in range(1, 34):
        n = int(rng.choice([3, 4095, 4096, 4097, 8191, 8192, 8193, 20000, 70001]))
        tags = _tags_of_width(rng, n, w)
        expect = oc.compress_int_array(tags.astype(np.int64))
        pad_out = int(rng.integers(1, 16))
        r, got = _encode_one(torch, device, L, tags, pad_in=int(rng.integers(0, 4)), pad_out=pad_out)
        assert expect[12] == w, (w, expect[12])
        assert r.status == 0 and r.out_count == len(expect) and r.width == w, (w, n, r.width)
        assert got[pad_out:pad_out + len(expect)].tobytes() == expect, (w, n)
        assert (got[:pad_out] == 0xEE).all() and (got[pad_out + len(expect):] == 0xEE).all(), (w, n)


def test_bytes_guards_and_alignments(env):
    torch, device, oc, L = env
    rng = np.random.default_rng(32)
    for n in (1, 2, 7, 255, 4095, 4096, 4097, 12288, 12289, 100003, (1 << 20) + 5):
        for kind in ("tags", "wide20", "raw"):
            tags = (rng.choice(np.array([0, 1, 2, 7], dtype=np.uint32), n, p=[0.5, 0.25, 0.15, 0.1]) if kind == "tags"
                    else rng.integers(5_000_000, 5_000_000 + (1 << 19), n, dtype=np.uint32) if kind == "wide20"
                    else rng.integers(0, 1 << 32, n, dtype=np.uint32))
            expect = oc.compress_int_array(tags.astype(np.int64))
            for pad_in, pad_out in ((0, 3), (1, 16), (3, 13)):
                r, got = _encode_one(torch, device, L, tags, pad_in, pad_out)
                assert r.status == 0 and r.out_count == len(expect), (n, kind)
                assert got[pad_out:pad_out + len(expect)].tobytes() == expect, (n, kind, pad_in, pad_out)
                assert (got[:pad_out] == 0xEE).all() and (got[pad_out + len(expect):] == 0xEE).all(), "wrote outside E"


def _shapes(rng, n):
    """u32 columns that take each route of the median selection: brackets of one, two and a few
    keys, a bracket below 4096 keys, wide brackets (random and sorted, whose candidates are one
    stretch of the array), a candidate list that overflows, a run of one value inside wide data."""
    ar = np.arange(n, dtype=np.uint64)
    u = rng.integers(0, 100, n)
    return {
        "one_key": np.where(u < 70, 5, rng.integers(0, 1 << 20, n)).astype(np.uint32),
        "two_keys": rng.integers(1000, 1002, n).astype(np.uint32),
        "few_keys": rng.choice(np.array([0, 1, 2, 7], dtype=np.uint32), n, p=[0.5, 0.25, 0.15, 0.1]),
        "heavy_ends": np.where(u < 45, 0, np.where(u < 48, 1, np.where(u < 51, 2, 3))).astype(np.uint32),
        "below_4096_keys": (3_000_000_000 + rng.integers(0, 3000, n)).astype(np.uint32),
        "wide": rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32),
        "wide_24": (1 << 30) + rng.integers(0, 1 << 24, n, dtype=np.uint64).astype(np.uint32),
        "sorted": (ar * 1000 % (1 << 32)).astype(np.uint32) if n * 1000 < (1 << 32) else (ar * 3).astype(np.uint32),
        "reversed": ((n - ar) * 7).astype(np.uint32),
        "three_far": rng.choice(np.array([0, 1 << 31, (1 << 32) - 1], dtype=np.uint32), n),
        "run_in_wide": np.where(u < 4, 2_000_000_000, rng.integers(0, 1 << 32, n, dtype=np.uint64)).astype(np.uint32),
        "sawtooth": ((ar % 4096) * 17).astype(np.uint32),
    }


@pytest.mark.parametrize("n", [5000, 70001, (1 << 21) + 3])
def test_every_selection_route_single_and_batched(env, n):
    torch, device, oc, L = env
    rng = np.random.default_rng(n + 1)
    shapes = _shapes(rng, n)
    expects = {k: oc.compress_int_array(v.astype(np.int64)) for k, v in shapes.items()}
    for k, v in shapes.items():                                   # one array per call
        r, got = _encode_one(torch, device, L, v, pad_in=1, pad_out=5)
        e = expects[k]
        assert r.status == 0 and r.out_count == len(e), (k, n)
        assert r.midpoint == struct.unpack("<q", e[:8])[0] and r.width == e[12], (k, n, r.midpoint, r.width)
        assert got[5:5 + len(e)].tobytes() == e, (k, n)
    # all of them as one submission (equal lengths), then with a shorter array added (ragged: the
    # generic pack kernel behind the u32 form of the bracket pass)
    for extra in ([], [shapes["wide"][:n // 2 + 1]]):
        cols = list(shapes.values()) + extra
        want = [expects[k] for k in shapes] + [oc.compress_int_array(c.astype(np.int64)) for c in extra]
        d_cols = [torch.from_numpy(c.view(np.int32).copy()).cuda() for c in cols]
        outs = [torch.full((device.encode_bound(c.size) + 16,), 0xEE, dtype=torch.uint8, device="cuda") for c in cols]
        res = device.new_results(len(cols), "cuda")
        device.encode_batch_dev(device.make_encode_descs(
            [(d.data_ptr(), c.size, o.data_ptr() + 1, device.encode_bound(c.size), L.ELEM_U32) for d, c, o in zip(d_cols, cols, outs)]), res)
        for i, (o, e, r) in enumerate(zip(outs, want, device.read_results(res))):
            got = o.cpu().numpy()
            assert r.status == 0 and r.out_count == len(e), (i, n)
            assert got[1:1 + len(e)].tobytes() == e and got[0] == 0xEE and (got[1 + len(e):] == 0xEE).all(), (i, n)


def test_capacity_error_writes_nothing(env):
    torch, device, oc, L = env
    tags = np.random.default_rng(33).integers(0, 9, 50000, dtype=np.uint32)
    expect = oc.compress_int_array(tags.astype(np.int64))
    r, got = _encode_one(torch, device, L, tags, cap=len(expect) - 1)
    assert r.status == L.ERR_CAPACITY and r.out_count == 0
    assert (got == 0xEE).all()
    r, got = _encode_one(torch, device, L, tags, cap=len(expect))
    assert r.status == 0 and got[3:3 + len(expect)].tobytes() == expect


@pytest.mark.parametrize("n", [1, 4096, 4097, 30000, (1 << 18) + 1])
def test_equal_length_batches(env, n):
    """Several u32 arrays of one length: tiles numbered per array.  One array is rejected for
    capacity and must leave its output alone; widths differ between the arrays."""
    torch, device, oc, L = env
    rng = np.random.default_rng(n)
    na = 9
    cols = [(rng.integers(0, 1 << int(rng.integers(1, 33)), n, dtype=np.uint64)).astype(np.uint32) for _ in range(na)]
    d_cols = [torch.from_numpy(c.view(np.int32)).cuda() for c in cols]
    bound = device.encode_bound(n)
    outs = [torch.full((bound + 40,), 0xEE, dtype=torch.uint8, device="cuda") for _ in cols]
    caps = [bound] * na
    caps[4] = 12                                             # cannot hold any encoding
    res = device.new_results(na, "cuda")
    device.encode_batch_dev(device.make_encode_descs([(d.data_ptr(), n, o.data_ptr() + 5, cap, L.ELEM_U32) for d, o, cap in zip(d_cols, outs, caps)]), res)
    rs = device.read_results(res)
    for i, c in enumerate(cols):
        got = outs[i].cpu().numpy()
        if i == 4:
            assert rs[i].status == L.ERR_CAPACITY and (got == 0xEE).all()
            continue
        expect = oc.compress_int_array(c.astype(np.int64))
        assert rs[i].status == 0 and rs[i].out_count == len(expect), i
        assert got[5:5 + len(expect)].tobytes() == expect, i
        assert (got[:5] == 0xEE).all() and (got[5 + len(expect):] == 0xEE).all(), i


def test_framed_equal_length_tag_arrays(env):
    """The tag arrays of equal Blocks as frames (write_u32_array_compressed, binary.rs:666-669)."""
    torch, device, oc, L = env
    rng = np.random.default_rng(35)
    n, na = 50001, 12
    cols = [rng.choice(np.array([0, 1, 2, 7], dtype=np.uint32), n) for _ in range(na)]
    d_cols = [torch.from_numpy(c.view(np.int32)).cuda() for c in cols]
    want = b"".join(struct.pack("<II", n, len(e)) + e for e in (oc.compress_int_array(c.astype(np.int64)) for c in cols))
    out = torch.full((len(want) + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    offs = torch.zeros(na + 1, dtype=torch.int64, device="cuda")
    res = device.new_results(na, "cuda")
    device.encode_frames_dev(device.make_encode_descs([(d.data_ptr(), n, 0, 0, L.ELEM_U32) for d in d_cols]), out[:len(want) + 7], offs, res)
    torch.cuda.synchronize()
    assert all(r.status == 0 for r in device.read_results(res))
    got = out.cpu().numpy()
    assert int(offs[na]) == len(want) and got[:len(want)].tobytes() == want and (got[len(want):] == 0xEE).all()
    # a buffer that ends inside frame 7: frames 0..6 as before, 7 and later reported and not written
    ends = np.cumsum([8 + len(oc.compress_int_array(c.astype(np.int64))) for c in cols])
    cap = int(ends[6]) + 5
    out.fill_(0xEE)
    device.encode_frames_dev(device.make_encode_descs([(d.data_ptr(), n, 0, 0, L.ELEM_U32) for d in d_cols]), out[:cap], offs, res)
    torch.cuda.synchronize()
    rs = device.read_results(res)
    assert [r.status for r in rs] == [0] * 7 + [L.ERR_CAPACITY] * (na - 7)
    got = out.cpu().numpy()
    assert got[:ends[6]].tobytes() == want[:ends[6]] and (got[ends[6]:] == 0xEE).all()
    assert [int(o) for o in offs.cpu()] == [0] + [int(e) for e in ends[:7]] + [int(ends[6])] * (na - 7)


def test_ragged_and_mixed_batches_keep_the_generic_kernel(env):
    torch, device, oc, L = env
    rng = np.random.default_rng(36)
    tags = [rng.integers(0, 5, n, dtype=np.uint32) for n in (5000, 5001, 4096, 1, 70000)]
    ints = [1_700_000_000_000 + rng.integers(0, 4000, 5000, dtype=np.int64)]
    for group in (tags, tags[:1] + ints + tags[1:2], [tags[0], tags[0].copy(), ints[0]]):
        d = [torch.from_numpy(c.view(np.int32) if c.dtype == np.uint32 else c).cuda() for c in group]
        outs = [torch.full((device.encode_bound(c.size) + 8,), 0xEE, dtype=torch.uint8, device="cuda") for c in group]
        res = device.new_results(len(group), "cuda")
        device.encode_batch_dev(device.make_encode_descs(
            [(t.data_ptr(), c.size, o.data_ptr(), device.encode_bound(c.size), L.ELEM_U32 if c.dtype == np.uint32 else L.ELEM_I64)
             for t, c, o in zip(d, group, outs)]), res)
        for c, o, r in zip(group, outs, device.read_results(res)):
            expect = oc.compress_int_array(c.astype(np.int64))
            assert r.status == 0 and o.cpu().numpy()[:len(expect)].tobytes() == expect
