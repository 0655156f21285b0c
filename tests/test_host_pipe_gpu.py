"""GPU parity of the host-pointer entry points at sizes where they move data through the pinned
ring with worker threads, and where decode is cut into pieces that are uploaded, decoded and
downloaded overlapped.  Results are compared with the oracle, as everywhere else."""
import struct
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    from oracle import zbra_oracle_c as oc
    from zbra_amd import compression as zc
    return oc, zc


PIECE = 1 << 23  # i64 values per piece of a piecewise host decode (64 MiB of output)


def _narrow(n, seed):
    rng = np.random.default_rng(seed)
    return 1_700_000_000_000 + rng.integers(0, 4000, n, dtype=np.int64)


@pytest.mark.parametrize("n", [(1 << 17) + 1, (1 << 20) - 3, 3 * (1 << 19) + 4097])
def test_ring_sizes_roundtrip(env, n):
    """Transfers of 1 to 12 MiB: whole chunks, a ragged last chunk, a ragged last piece."""
    oc, zc = env
    v = _narrow(n, n)
    e = zc.compress_int_array(v)
    assert e == oc.compress_int_array(v)
    assert np.array_equal(zc.decompress_int_array(e, n), v)
    wide = np.random.default_rng(n).integers(-(1 << 62), 1 << 62, n, dtype=np.int64)
    e = zc.compress_int_array(wide)
    assert e == oc.compress_int_array(wide)
    assert np.array_equal(zc.decompress_int_array(e, n), wide)


@pytest.mark.parametrize("n", [2 * PIECE, 2 * PIECE + 777, 3 * PIECE - 8])
def test_piecewise_decode_packed(env, n):
    oc, zc = env
    v = _narrow(n, 5)
    v[n // 3] = 1_700_000_004_000 + (1 << 20)  # width 22: pieces start at bytes that are not multiples of 4
    e = oc.compress_int_array(v)
    assert e[12] == 22
    got = zc.decompress_int_array(e, n)
    assert np.array_equal(got, v)
    # the same bytes with fewer values asked for: pieces end inside the payload
    m = 2 * PIECE + 5
    if m < n:
        assert np.array_equal(zc.decompress_int_array(e, m), v[:m])


def test_piecewise_decode_raw_fill_and_short_tail(env):
    oc, zc = env
    n = 2 * PIECE + 4099
    wide = np.random.default_rng(9).integers(-(1 << 62), 1 << 62, n, dtype=np.int64)
    e = oc.compress_int_array(wide)
    assert e[12] >= 32
    assert np.array_equal(zc.decompress_int_array(e, n), wide)
    # raw payload that stops 5 bytes early: the last value is zero-extended (compression.rs:196-205)
    short = bytearray(e[:-5])
    short[8:12] = struct.pack("<I", len(short) - 12)
    st, want = oc.decompress_int_array(bytes(short), n)
    assert st == 0
    assert np.array_equal(zc.decompress_int_array(bytes(short), n), want)
    # ... and one that stops inside an earlier piece: every later value is 0 -> un-zig-zag 0 + midpoint
    cut = 13 + 8 * (PIECE + 100) + 3
    short = bytearray(e[:cut])
    short[8:12] = struct.pack("<I", len(short) - 12)
    st, want = oc.decompress_int_array(bytes(short), n)
    assert st == 0
    assert np.array_equal(zc.decompress_int_array(bytes(short), n), want)
    # width byte 0: count copies of the midpoint (compression.rs:187-189)
    fill = struct.pack("<qIB", -123456789, 1, 0)
    got = zc.decompress_int_array(fill, n)
    assert got.shape == (n,) and got.min() == got.max() == -123456789


def test_piecewise_decode_u32(env):
    oc, zc = env
    n = 4 * PIECE + 12345  # u32 pieces hold twice the values
    tags = (np.random.default_rng(3).integers(0, 8, n, dtype=np.int64) * 37 % 11).astype(np.uint32)
    e = zc.compress_u32_array(tags)
    assert e == oc.compress_int_array(tags.astype(np.int64))
    assert np.array_equal(zc.decompress_u32_array(e, n), tags)


def test_host_calls_from_threads(env):
    """Host-pointer calls of one device run on a few lanes at a time (each with its own streams, rings
    and device buffers) and queue beyond that; concurrent callers must all get their own bytes."""
    oc, zc = env
    n = (1 << 20) + 17
    cols = [_narrow(n, 100 + t) + t * 1_000_003 for t in range(4)]
    want = [oc.compress_int_array(c) for c in cols]
    errs = []

    def work(t):
        try:
            for _ in range(3):
                e = zc.compress_int_array(cols[t])
                assert e == want[t]
                assert np.array_equal(zc.decompress_int_array(e, n), cols[t])
        except Exception as ex:  # noqa: BLE001
            errs.append((t, ex))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_every_host_entry_point_from_more_threads_than_lanes(env):
    """Six threads, each a different mix of host-pointer calls (single arrays of several sizes incl.
    a piecewise decode, stage functions, framed batches both ways), all at once: more callers than
    the device has host lanes, so lanes are shared in turn and every call must still see only its
    own buffers.  Everything against the oracle."""
    oc, zc = env
    rng = np.random.default_rng(77)
    big_n = 2 * PIECE + 1234                       # piecewise decode: two streams and four events of a lane
    big = _narrow(big_n, 9)
    big_e = oc.compress_int_array(big)
    mids = [_narrow((1 << 18) + 13 * t, 200 + t) - t * 999 for t in range(6)]
    wides = [rng.integers(-(1 << 62), 1 << 62, (1 << 16) + t, dtype=np.int64) for t in range(6)]
    tags = [rng.choice(np.array([0, 1, 2, 7], dtype=np.uint32), (1 << 19) + t) for t in range(6)]
    batch = [_narrow(5000 + 777 * k, 300 + k) for k in range(40)] + [np.zeros(0, dtype=np.int64), wides[0][:3001]]
    batch_frames = b"".join(struct.pack("<II", v.size, len(e)) + e for v, e in ((v, oc.compress_int_array(v)) for v in batch))
    errs = []

    def single(t):
        for _ in range(3):
            e = zc.compress_int_array(mids[t]); assert e == oc.compress_int_array(mids[t])
            assert np.array_equal(zc.decompress_int_array(e, mids[t].size), mids[t])
            e = zc.compress_int_array(wides[t]); assert e == oc.compress_int_array(wides[t])
            assert np.array_equal(zc.decompress_int_array(e, wides[t].size), wides[t])
            e = zc.compress_u32_array(tags[t]); assert e == oc.compress_int_array(tags[t].astype(np.int64))
            assert np.array_equal(zc.decompress_u32_array(e, tags[t].size), tags[t])

    def piecewise(t):
        for _ in range(2):
            assert np.array_equal(zc.decompress_int_array(big_e, big_n), big)

    def stages(t):
        for _ in range(3):
            mid, deltas = zc.frame_of_reference_encode(mids[t])
            omid, odeltas = oc.frame_of_reference_encode(mids[t])
            assert mid == omid and np.array_equal(deltas, odeltas)
            zz = zc.zig_zag_encode(deltas); assert np.array_equal(zz, oc.zig_zag_encode(odeltas))
            packed = zc.bp64_pack(zz); assert packed == oc.bp64_pack(zz)
            assert np.array_equal(zc.bp64_unpack(packed, zz.size), zz)
            assert np.array_equal(zc.frame_of_reference_decode(mid, zc.zig_zag_decode(zz)), mids[t])

    def framed(t):
        for _ in range(3):
            frames, offs = zc.write_int_arrays_compressed(batch)
            assert frames == batch_frames
            got = zc.read_int_arrays_compressed([(frames[offs[k] + 8:offs[k + 1]], batch[k].size, np.int64) for k in range(len(batch))])
            for g, v in zip(got, batch):
                assert np.array_equal(g, v)

    jobs = [single, piecewise, stages, framed, single, framed]

    def work(t):
        try:
            jobs[t](t)
        except Exception as ex:  # noqa: BLE001
            errs.append((t, jobs[t].__name__, repr(ex)))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(len(jobs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_short_raw_payload_first_call_in_fresh_process(tmp_path):
    """A raw array (width byte >= 32) whose payload ends before later 64-MiB pieces begin, decoded as
    the FIRST call of a process: the input buffer on the device is then only as large as the few
    bytes of E, and pieces that start past the payload must not read outside it.  Every value past
    the payload is 0 -> midpoint (compression.rs:196-205)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "fresh.py"
    script.write_text(
        "import struct, sys\n"
        "import numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from oracle import zbra_oracle_c as oc\n"
        "from zbra_amd import compression as zc\n"
        "n = (1 << 24) + 3\n"
        "e = struct.pack('<qIB', 41, 9, 40) + bytes(range(1, 9))\n"
        "st, want = oc.decompress_int_array(e, n)\n"
        "assert st == 0 and want.shape == (n,)\n"
        "got = zc.decompress_int_array(e, n)\n"
        "assert np.array_equal(got, want)\n"
        "got32 = zc.decompress_u32_array(e, 2 * n)\n"
        "st, want = oc.decompress_int_array(e, 2 * n)\n"
        "assert np.array_equal(got32, want.astype(np.uint32))\n"
        "print('fresh ok')\n")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ))
    assert out.returncode == 0 and "fresh ok" in out.stdout, out.stderr[-2000:]
