"""The C++ tree writer / reader (include/zbra/binary_file.hpp): tests/cpp/test_binary_file.cpp checks
it against a reference-style walk with the oracle and reads the image back; here the image it leaves
behind is compared, byte for byte, with the image the Python writer builds for the same table with
the CPU oracle as codec.  Both sides generate the table from the same counter-based generator."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "tests", "cpp", "test_binary_file")
M64 = (1 << 64) - 1


def sm(seed: int, n: int) -> np.ndarray:
    """splitmix64 finaliser of seed + (i + 1) * golden ratio, i = 0..n-1, as uint64."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z ^= z >> np.uint64(30); z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27); z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return z


def rich_table(zb, rows: int, s: int):
    """tests/cpp/test_binary_file.cpp::rich_table, value for value."""
    u = np.uint64
    ints = 1_700_000_000_000 + (sm(s, rows) % u(4000)).astype(np.int64)
    wide = sm(s + 1, rows).view(np.int64)
    dbl = sm(s + 2, rows).view(np.float64)
    strlens = sm(s + 3, rows) % u(4)
    arrlens = sm(s + 4, rows) % u(5)
    which = (sm(s + 6, rows) % u(4)).astype(np.int64)
    tags = np.array([0, 1, 2, 7], dtype=np.uint32)[which]
    counts = [int((which == v).sum()) for v in range(4)]
    strdata = bytes((ord("a") + k % 26) for k in range(int(strlens.sum())))
    elems = (sm(s + 5, int(arrlens.sum())) % u(2000)).astype(np.int64) - 1000
    bx = 5_000_000 + (sm(s + 7, counts[1]) % u(1000)).astype(np.int64)
    by = 1_640_000_000_000 + np.arange(counts[1], dtype=np.int64) * 1000
    cz = (sm(s + 8, counts[2]) % u(1 << 30)).astype(np.int64) - (1 << 29)
    keys = np.arange(2 * counts[3], dtype=np.int64)
    mapvals = np.arange(2 * counts[3], dtype=np.uint64).view(np.float64)
    return zb.ArrayTable(zb.Struct([
        ("ints", zb.Int(ints)),
        ("wide", zb.Int(wide)),
        ("dbl", zb.Double(dbl)),
        ("str", zb.Binary(strlens, strdata)),
        ("arr", zb.Array(arrlens, zb.Int(elems, "Date"))),
        ("sum", zb.Enum(tags, [("a", 0, zb.Unit(counts[0])),
                               ("b", 1, zb.Struct([("x", zb.Int(bx)), ("y", zb.Int(by, "TimeMilliseconds"))])),
                               ("c", 2, zb.Reversed(zb.Int(cz))),
                               ("d", 7, zb.Nested(np.full(counts[3], 2, dtype=np.uint64), zb.MapTable(zb.Int(keys), zb.Double(mapvals))))])),
        ("blob", zb.Nested(np.ones(rows, dtype=np.uint64), zb.BinaryTable(b"\x00\x01" * rows))),
    ]))


def _ensure_built():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])


def test_generator_matches_the_oracle():
    from oracle import zbra_oracle_c as oc
    got = sm(0x5EED0002, 50)
    assert [int(x) for x in got] == [oc.sm(0x5EED0002, i) & M64 for i in range(50)]


@pytest.mark.gpu
def test_cpp_image_equals_python_oracle_image(tmp_path):
    from test_binary_file import OracleCodec
    from zbra_amd import binary as zb
    _ensure_built()
    dump = tmp_path / "cpp.zbra"
    p = subprocess.run([EXE, "--dump", str(dump)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and " 0 failures" in p.stdout, p.stdout + p.stderr
    blocks = [zb.Block(n, rich_table(zb, n, 100 + 10 * i)) for i, n in enumerate([0, 1, 257, 5000, 70001])]
    f = zb.BinaryFile("{}", blocks)
    want = f.to_bytes(OracleCodec)
    got = dump.read_bytes()
    assert len(got) == len(want) and got == want
    assert f.to_bytes() == want          # and the Python writer on the GPU agrees with both


def test_cpp_tree_writer_refuses_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    _ensure_built()
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=60)
    assert p.returncode == 2 and "need a GPU" in p.stdout
