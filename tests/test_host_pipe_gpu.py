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
    """Host-pointer calls are serialised per device; concurrent callers must all get their own bytes."""
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
