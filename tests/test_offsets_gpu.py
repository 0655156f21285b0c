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
