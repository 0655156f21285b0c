"""GPU parity of the fused lengths -> offsets decode: the oracle decodes the lengths
(read_int_array_usize_compressed, binary.rs:661-664), numpy forms the running sum the reference's
consumers compute (`offset += length`, striped.rs:523-583), the device must produce the same
count + 1 words and touch nothing else."""
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


def _want(lengths):
    out = np.zeros(lengths.size + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        np.cumsum(lengths.astype(np.uint64), out=out[1:])
    return out


def _run(torch, device, e: bytes, count: int, pad: int = 0):
    buf = torch.zeros(pad + len(e) + 64, dtype=torch.uint8, device="cuda")
    if len(e):
        buf[pad:pad + len(e)] = torch.frombuffer(bytearray(e), dtype=torch.uint8).cuda()
    offs = torch.full((count + 3,), -7, dtype=torch.int64, device="cuda")
    res = device.new_results(1, "cuda")
    device.decode_offsets_dev(buf[pad:pad + len(e)] if len(e) else buf[:0], count, offs[1:], res, length=len(e))
    return offs.cpu().numpy(), device.read_results(res)[0]


@pytest.mark.parametrize("n", [1, 2, 3, 2047, 2048, 2049, 4097, 100003, (1 << 21) + 5, 40_000_000])
def test_offsets_match_cumsum(env, n):
    torch, device, oc, _ = env
    rng = np.random.default_rng(n)
    shapes = [rng.integers(0, 50, n, dtype=np.int64),                       # short strings: w <= 6
              rng.integers(0, 1 << 20, n, dtype=np.int64)]                  # w = 21
    if n <= 100003:
        shapes += [np.zeros(n, dtype=np.int64),                             # all empty: w = 1, midpoint 0
                   np.full(n, 17, dtype=np.int64),                          # constant
                   rng.integers(0, 1 << 40, n, dtype=np.int64),             # raw fallback
                   rng.integers(-(1 << 62), 1 << 62, n, dtype=np.int64)]    # not lengths at all: wrapping sums
    for lengths in shapes:
        e = oc.compress_int_array(lengths)
        got, r = _run(torch, device, e, n, pad=n % 5)
        assert r.status == 0 and r.out_count == n
        assert got[0] == -7 and got[n + 2] == -7, "wrote outside the offsets"
        assert np.array_equal(got[1:n + 2].view(np.uint64), _want(lengths)), (n, int(e[12]))


@pytest.mark.parametrize("w", list(range(1, 32)))
def test_offsets_every_width(env, w):
    """Every packed width, signed deltas around a non-zero midpoint, several tiles and a ragged last
    one, payload at every byte alignment the test's padding reaches."""
    torch, device, oc, _ = env
    rng = np.random.default_rng(100 + w)
    for n in (257, 2048 * 3 + 1, 20011):
        half = 1 << (w - 1)
        x = 1000 + rng.integers(-(half // 2), half - half // 2, n, dtype=np.int64)
        x[rng.integers(0, n)] = 1000 - half // 2          # both ends of the range are present
        x[rng.integers(0, n)] = 1000 + half - half // 2 - 1
        e = oc.compress_int_array(x)
        assert abs(int(e[12]) - w) <= 1
        got, r = _run(torch, device, e, n, pad=(w + n) % 16)
        assert r.status == 0 and r.out_count == n
        assert got[0] == -7 and got[n + 2] == -7
        assert np.array_equal(got[1:n + 2].view(np.uint64), _want(x)), (n, w, int(e[12]))


def test_offsets_empty_rejected_and_crafted(env):
    torch, device, oc, L = env
    e = oc.compress_int_array(np.arange(10, dtype=np.int64))
    # empty data: empty Vec whatever count says -> only offsets[0]
    got, r = _run(torch, device, b"", 5)
    assert r.status == 0 and r.out_count == 0 and got[1] == 0 and (got[2:] == -7).all() and got[0] == -7
    # count 0
    got, r = _run(torch, device, e, 0)
    assert r.status == 0 and r.out_count == 0 and got[1] == 0 and (got[2:] == -7).all()
    # rejected: short header, short packed slice, truncated payload
    for bad, status in ((e[:11], L.ERR_SHORT_HEADER), (e[:-1], L.ERR_SHORT_PACKED),
                        (e[:8] + struct.pack("<I", len(e) - 13) + e[12:-1], L.ERR_TRUNCATED)):
        got, r = _run(torch, device, bad, 10)
        assert r.status == status and r.out_count == 0 and got[1] == 0 and (got[2:] == -7).all()
    # width byte 0: every length equals the midpoint (compression.rs:187-189)
    fill = struct.pack("<qIB", 3, 1, 0)
    got, r = _run(torch, device, fill, 5000)
    assert r.status == 0 and np.array_equal(got[1:5002], np.arange(5001, dtype=np.int64) * 3)
    # raw payload that stops early: the missing values read as zig-zag 0 -> midpoint
    wide = np.random.default_rng(1).integers(0, 1 << 40, 3000, dtype=np.int64)
    ew = oc.compress_int_array(wide)
    short = bytearray(ew[: 13 + 8 * 1000 + 3])
    short[8:12] = struct.pack("<I", len(short) - 12)
    st, vals = oc.decompress_int_array(bytes(short), 3000)
    assert st == 0
    got, r = _run(torch, device, bytes(short), 3000)
    assert r.status == 0 and np.array_equal(got[1:3002].view(np.uint64), _want(vals))


def _run_batch(torch, device, L, encs, counts, seed=0):
    """Encoded arrays back to back in one device buffer at odd offsets; one batched submission; returns
    each array's offsets buffer (two guard words in front, at least two behind) and the result words."""
    rng = np.random.default_rng(seed)
    blob, at = bytearray(), []
    for e in encs:
        blob += b"\x55" * int(rng.integers(0, 7))
        at.append(len(blob))
        blob += e
    blob += b"\0" * 64
    d_blob = torch.frombuffer(blob, dtype=torch.uint8).cuda()
    outs = [torch.full((c + 4,), -7, dtype=torch.int64, device="cuda") for c in counts]
    descs = device.make_descs([(d_blob.data_ptr() + a if len(e) else 0, len(e), c, o.data_ptr() + 16, L.ELEM_I64)
                               for a, e, c, o in zip(at, encs, counts, outs)])
    res = device.new_results(len(encs), "cuda")
    device.decode_offsets_batch_dev(descs, res)
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in outs], device.read_results(res)


def test_offsets_batch_ragged_with_empty_and_rejected_arrays(env):
    """Many lengths arrays in one submission: every width class, counts from 0 to several tiles, an
    array with empty data, one with count 0, three rejected ones between good ones.  Every array's
    offsets must equal the cumsum of the oracle's decode of THAT array (each sum starts at 0), the
    rejected and empty ones get offsets[0] = 0 only, nothing is written outside."""
    torch, device, oc, L = env
    rng = np.random.default_rng(77)
    arrays = []
    for k, n in enumerate([1, 2047, 2048, 2049, 5, 70001, 300, 6145, 40000, 3, 1 << 18, 9000, 12289, 64, 100003, 4097]):
        kind = k % 5
        if kind == 0: v = rng.integers(0, 50, n, dtype=np.int64)
        elif kind == 1: v = rng.integers(0, 1 << 20, n, dtype=np.int64)
        elif kind == 2: v = np.full(n, 9, dtype=np.int64)
        elif kind == 3: v = rng.integers(0, 1 << 40, n, dtype=np.int64)            # raw
        else: v = rng.integers(-(1 << 62), 1 << 62, n, dtype=np.int64)             # wrapping sums
        arrays.append(v)
    encs = [oc.compress_int_array(v) for v in arrays]
    counts = [v.size for v in arrays]
    want = [_want(v) for v in arrays]
    status = [0] * len(arrays)
    # empty data with a count; count 0 with data; short header; short packed slice; truncated payload
    good = oc.compress_int_array(np.arange(10, dtype=np.int64))
    extra = [(b"", 5, 0), (good, 0, 0), (good[:11], 10, L.ERR_SHORT_HEADER), (good[:-1], 10, L.ERR_SHORT_PACKED),
             (good[:8] + struct.pack("<I", len(good) - 13) + good[12:-1], 10, L.ERR_TRUNCATED)]
    for pos, (e, c, st) in zip((2, 5, 9, 12, 16), extra):
        encs.insert(pos, e); counts.insert(pos, c); want.insert(pos, np.zeros(1, dtype=np.uint64)); status.insert(pos, st)
    outs, rs = _run_batch(torch, device, L, encs, counts, seed=3)
    for i, (o, r) in enumerate(zip(outs, rs)):
        m = want[i].size
        assert r.status == status[i], i
        assert r.out_count == m - 1, i
        assert o[0] == -7 and o[1] == -7 and (o[2 + m:] == -7).all(), (i, "wrote outside the offsets")
        assert np.array_equal(o[2:2 + m].view(np.uint64), want[i]), (i, counts[i])


def test_offsets_batch_equals_single_calls(env):
    """The batched call and one single call per array give the same words, on a batch of the
    width sweep (every width 1..31 in one submission)."""
    torch, device, oc, L = env
    rng = np.random.default_rng(5)
    arrays = []
    for w in range(1, 32):
        half = 1 << (w - 1)
        n = int(rng.integers(1, 9000))
        arrays.append(1000 + rng.integers(-(half // 2), half - half // 2, n, dtype=np.int64))
    encs = [oc.compress_int_array(v) for v in arrays]
    outs, rs = _run_batch(torch, device, L, encs, [v.size for v in arrays], seed=9)
    for i, v in enumerate(arrays):
        single, r1 = _run(torch, device, encs[i], v.size, pad=i % 16)
        assert rs[i].status == 0 and r1.status == 0 and rs[i].width == r1.width
        assert np.array_equal(outs[i][2:3 + v.size], single[1:2 + v.size]), i
        assert np.array_equal(outs[i][2:3 + v.size].view(np.uint64), _want(v)), i


def test_offsets_batch_invalid_arguments(env):
    torch, device, oc, L = env
    e = torch.zeros(64, dtype=torch.uint8, device="cuda")
    o = torch.zeros(8, dtype=torch.int64, device="cuda")
    assert L.lib.zbra_hip_decode_offsets_batch_dev(None, 0, None, None) == 0
    no_out = device.make_descs([(e.data_ptr(), 20, 3, 0, L.ELEM_I64)])
    assert L.lib.zbra_hip_decode_offsets_batch_dev(no_out, 1, None, None) == L.ERR_INVALID_ARG
    narrow = device.make_descs([(e.data_ptr(), 20, 3, o.data_ptr(), L.ELEM_U32)])      # offsets are 8-byte words
    assert L.lib.zbra_hip_decode_offsets_batch_dev(narrow, 1, None, None) == L.ERR_INVALID_ARG
