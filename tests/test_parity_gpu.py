"""GPU parity: every entry point of the host-pointer C ABI (through zbra_amd.compression, the
mirror of zbra_core::compression) against the CPU oracle, bit for bit, on the golden vectors, on
seeded columns of every width class, and on the edge cases the reference defines."""
import numpy as np
import pytest

from conftest import KINDS, SIZES, column

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zc():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    from zbra_amd import compression
    return compression


@pytest.fixture(scope="module")
def oc():
    from oracle import zbra_oracle_c
    return zbra_oracle_c


def test_golden_pipeline(zc, golden):
    for vec in golden["pipeline"]:
        values = np.array([int(v) for v in vec["values"]], dtype=np.int64)
        e = zc.compress_int_array(values)
        assert e.hex() == vec["hex"], vec["name"]
        back = zc.decompress_int_array(bytes.fromhex(vec["hex"]), values.size)
        assert np.array_equal(back, values), vec["name"]


def test_golden_pack(zc, golden):
    for vec in golden["pack"]:
        values = np.array([int(v) for v in vec["values"]], dtype=np.uint64)
        assert zc.bp64_pack(values).hex() == vec["hex"], vec["name"]
        assert np.array_equal(zc.bp64_unpack(bytes.fromhex(vec["hex"]), values.size), values), vec["name"]


def test_golden_stages_published_by_the_reference(zc, golden):
    """docs/compression-pipeline.md:27-82: median, deltas, zig-zag values and bit width as the
    reference's documentation states them, through the stage functions on the device."""
    for vec in golden["stages"]:
        values = np.array([int(v) for v in vec["values"]], dtype=np.int64)
        pub = vec["published"]
        mid, deltas = zc.frame_of_reference_encode(values)
        assert int(mid) == int(pub["median"]) and [int(d) for d in deltas] == [int(d) for d in pub["deltas"]]
        zz = zc.zig_zag_encode(deltas)
        assert [int(z) for z in zz] == [int(z) for z in pub["zigzag"]]
        packed = zc.bp64_pack(zz)
        assert packed[0] == pub["w"]
        assert np.array_equal(zc.zig_zag_decode(zc.bp64_unpack(packed, values.size)), deltas)
        assert zc.compress_int_array(values).hex() == vec["hex"]


@pytest.mark.parametrize("kind", KINDS)
def test_pipeline_matches_oracle(zc, oc, kind):
    for n in SIZES:
        v = column(kind, n, seed=1000 + n)
        expect = oc.compress_int_array(v)
        got = zc.compress_int_array(v)
        assert got == expect, (kind, n, got[:16].hex(), expect[:16].hex())
        back = zc.decompress_int_array(expect, n)
        assert np.array_equal(back, v), (kind, n)


@pytest.mark.parametrize("seed", range(40))
def test_every_width(zc, oc, seed):
    for n in (1, 100, 2049, 70001):
        v = column("width_sweep", n, seed)
        expect = oc.compress_int_array(v)
        assert zc.compress_int_array(v) == expect, (seed, n)
        assert np.array_equal(zc.decompress_int_array(expect, n), v), (seed, n)


@pytest.mark.parametrize("kind", KINDS)
def test_stage_functions_match_oracle(zc, oc, kind):
    for n in (1, 2, 65, 2049, 30011):
        v = column(kind, n, seed=7 + n)
        mid_o, d_o = oc.frame_of_reference_encode(v)
        mid_g, d_g = zc.frame_of_reference_encode(v)
        assert mid_g == mid_o and np.array_equal(d_g, d_o), (kind, n)
        assert np.array_equal(zc.frame_of_reference_decode(mid_o, d_o), v)
        z_o = oc.zig_zag_encode(d_o)
        assert np.array_equal(zc.zig_zag_encode(d_o), z_o)
        assert np.array_equal(zc.zig_zag_decode(z_o), d_o)
        p_o = oc.bp64_pack(z_o)
        assert zc.bp64_pack(z_o) == p_o, (kind, n)
        assert np.array_equal(zc.bp64_unpack(p_o, n), z_o), (kind, n)


def test_reference_proptest_shapes(zc, oc):
    # compression.rs:378-406: vec(any::<i64>(), 0..100) and vec(0..1000u64, 0..100)
    rng = np.random.default_rng(20250718)
    for _ in range(60):
        n = int(rng.integers(0, 100))
        v = rng.integers(-(1 << 63), (1 << 63) - 1, n, dtype=np.int64, endpoint=True)
        e = zc.compress_int_array(v)
        assert e == oc.compress_int_array(v)
        assert np.array_equal(zc.decompress_int_array(e, n), v)
        u = rng.integers(0, 1000, n, dtype=np.uint64)
        p = zc.bp64_pack(u)
        assert p == oc.bp64_pack(u)
        assert np.array_equal(zc.bp64_unpack(p, n), u)


def test_u32_tag_arrays(zc, oc):
    # binary.rs:666-674: tags are widened to i64 before the codec and narrowed `as u32` after
    rng = np.random.default_rng(5)
    for n in (1, 5, 4096, 100001):
        tags = rng.choice(np.array([0, 1, 2, 7], dtype=np.uint32), n, p=[0.5, 0.25, 0.15, 0.1])
        expect = oc.compress_int_array(tags.astype(np.int64))
        assert zc.compress_u32_array(tags) == expect
        assert np.array_equal(zc.decompress_u32_array(expect, n), tags)
    big = rng.integers(0, 1 << 32, 5000, dtype=np.uint32)  # needs the raw fallback
    expect = oc.compress_int_array(big.astype(np.int64))
    assert zc.compress_u32_array(big) == expect
    assert np.array_equal(zc.decompress_u32_array(expect, big.size), big)
    # narrowing truncates: an i64 column decoded as u32 keeps the low 32 bits
    wide = np.array([1 << 40, (1 << 33) + 5, -1], dtype=np.int64)
    e = oc.compress_int_array(wide)
    assert np.array_equal(zc.decompress_u32_array(e, 3), wide.astype(np.uint32))


def test_double_columns_as_bit_patterns(zc, oc):
    # binary.rs:299-300 / 378-382: f64::to_bits as i64 through the same codec
    rng = np.random.default_rng(11)
    x = rng.normal(size=5001)
    bits = x.view(np.int64)
    e = zc.compress_int_array(bits)
    assert e == oc.compress_int_array(bits)
    assert np.array_equal(zc.decompress_int_array(e, bits.size).view(np.float64), x)


def test_empty_and_error_conventions(zc, oc):
    assert zc.compress_int_array(np.zeros(0, dtype=np.int64)) == b""
    assert zc.bp64_pack(np.zeros(0, dtype=np.uint64)) == b""
    assert zc.decompress_int_array(b"", 5).size == 0
    assert zc.bp64_unpack(b"", 5).size == 0
    mid, d = zc.frame_of_reference_encode(np.zeros(0, dtype=np.int64))
    assert mid == 0 and d.size == 0
    with pytest.raises(zc.DecompressionError, match="Invalid compressed data length"):
        zc.decompress_int_array(b"\0" * 11, 1)
    e = oc.compress_int_array([1, 2, 3, 4, 5])
    with pytest.raises(zc.DecompressionError, match="Insufficient data for packed array"):
        zc.decompress_int_array(e[:-1], 5)
    assert zc.decompress_int_array(b"\x05" + b"\0" * 11, 4).size == 0
    w0 = (9).to_bytes(8, "little") + (1).to_bytes(4, "little") + b"\0"
    assert zc.decompress_int_array(w0, 3).tolist() == [9, 9, 9]
    raw = (0).to_bytes(8, "little") + (12).to_bytes(4, "little") + bytes([40]) + (6).to_bytes(8, "little") + b"\1\1\1"
    assert zc.decompress_int_array(raw, 3).tolist() == [3, 0, 0]
    raw200 = (0).to_bytes(8, "little") + (9).to_bytes(4, "little") + bytes([200]) + (8).to_bytes(8, "little")
    assert zc.decompress_int_array(raw200, 1).tolist() == [4]
    assert zc.decompress_int_array(e + b"\xff\xff", 5).tolist() == [1, 2, 3, 4, 5]
    assert zc.decompress_int_array(e, 0).size == 0
    short = e[:8] + (len(e) - 13).to_bytes(4, "little") + e[12:-1]
    with pytest.raises(zc.DecompressionError):
        zc.decompress_int_array(short, 5)
    # decoding fewer values than were encoded is legal (count comes from the frame, not from E)
    v = column("narrow", 5000, 3)
    ev = oc.compress_int_array(v)
    assert np.array_equal(zc.decompress_int_array(ev, 1234), v[:1234])


def test_large_column_roundtrip_and_oracle(zc, oc):
    n = (1 << 22) + 12345
    v = oc.gen_narrow(0x5EED0002, 0, n)
    expect = oc.compress_int_array(v)
    got = zc.compress_int_array(v)
    assert got == expect
    assert got[12] == 12
    assert np.array_equal(zc.decompress_int_array(got, n), v)


def test_encode_select_paths(zc, oc):
    """Arrays long enough for every route of the median selection: bracket kept as a histogram
    (narrow), as a candidate list (wide), rejected because the list overflows (few distinct
    values far apart) or because the sample misleads (periodic data), medians on a run of equal
    values, and the even-length case whose upper median lies outside the bracket."""
    rng = np.random.default_rng(77)
    I64_MIN, I64_MAX = -(1 << 63), (1 << 63) - 1
    for n in (4097, 131071, 300001, 300002, (1 << 20) + 3):
        cases = {
            "narrow": 1_700_000_000_000 + rng.integers(0, 4000, n, dtype=np.int64),
            "wide": rng.integers(-(1 << 45), 1 << 45, n, dtype=np.int64),
            "full": rng.integers(I64_MIN, I64_MAX, n, dtype=np.int64, endpoint=True),
            "two_far": np.where(np.arange(n) % 2 == 0, I64_MIN, I64_MAX).astype(np.int64),
            "two_near": rng.integers(0, 2, n, dtype=np.int64) * 5000 + 17,
            "mostly_constant": np.where(rng.random(n) < 0.99, 42, rng.integers(-(1 << 50), 1 << 50, n, dtype=np.int64)),
            "sorted": np.sort(rng.integers(-(1 << 40), 1 << 40, n, dtype=np.int64)),
            "reverse_sorted": np.sort(rng.integers(-(1 << 40), 1 << 40, n, dtype=np.int64))[::-1].copy(),
            "periodic": (np.arange(n, dtype=np.int64) % 73) * (1 << 33),
            "half_min": np.where(np.arange(n) < n // 2 + 1, I64_MIN, rng.integers(0, 1 << 62, n, dtype=np.int64)),
            "step": np.where(np.arange(n) < n // 2, -(1 << 61), 1 << 61).astype(np.int64),
        }
        for name, v in cases.items():
            v = np.ascontiguousarray(v, dtype=np.int64)
            expect = oc.compress_int_array(v)
            got = zc.compress_int_array(v)
            assert got[:13] == expect[:13], (name, n, got[:13].hex(), expect[:13].hex())
            assert got == expect, (name, n)
