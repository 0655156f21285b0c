"""CPU tests of the oracle itself: the C restatement and the numpy restatement of
zbra-core/src/compression.rs against the golden vectors, against each other, and against the
properties the reference's own tests assert.  No GPU, no HIP library."""
import numpy as np
import pytest

from conftest import KINDS, SIZES, column
from oracle import zbra_oracle_c as oc
from oracle import zbra_oracle_np as on

ORACLES = [("c", oc), ("numpy", on)]


@pytest.mark.parametrize("name,o", ORACLES)
def test_golden_pipeline(golden, name, o):
    assert [v["name"] for v in golden["pipeline"] if not v["hand"]["hex"]] == ["M"]  # 643 bytes: too long to type
    for vec in golden["pipeline"]:
        values = [int(v) for v in vec["values"]]
        e = o.compress_int_array(values)
        assert e.hex() == vec["hex"], vec["name"]
        hand = vec["hand"]
        assert int.from_bytes(e[:8], "little", signed=True) == int(hand["mid"]), vec["name"]
        assert e[12] == hand["w"] and len(e) == hand["len"], vec["name"]
        if hand["hex"]:
            assert e.hex() == hand["hex"], vec["name"]
        st, back = o.decompress_int_array(e, len(values))
        assert st == 0 and back.tolist() == values, vec["name"]


@pytest.mark.parametrize("name,o", ORACLES)
def test_golden_pack(golden, name, o):
    for vec in golden["pack"]:
        values = [int(v) for v in vec["values"]]
        p = o.bp64_pack(values)
        assert p.hex() == vec["hex"] and p[0] == vec["hand"]["w"] and len(p) == vec["hand"]["packed_len"], vec["name"]
        st, back = o.bp64_unpack(p, len(values))
        assert st == 0 and back.tolist() == values


@pytest.mark.parametrize("name,o", ORACLES)
def test_golden_stages_published_by_the_reference(golden, name, o):
    """docs/compression-pipeline.md:27-82: the one example whose intermediate results the reference states."""
    assert golden["stages"]
    for vec in golden["stages"]:
        values = [int(v) for v in vec["values"]]
        pub = vec["published"]
        mid, deltas = o.frame_of_reference_encode(values)
        assert int(mid) == int(pub["median"]) and [int(d) for d in deltas] == [int(d) for d in pub["deltas"]]
        zz = o.zig_zag_encode(deltas)
        assert [int(z) for z in zz] == [int(z) for z in pub["zigzag"]]
        assert o.bp64_pack(zz)[0] == pub["w"]
        assert o.compress_int_array(values).hex() == vec["hex"] == vec["hand"]["hex"]  # bytes derived by hand from the published numbers


@pytest.mark.parametrize("kind", KINDS)
def test_c_and_numpy_agree(kind):
    for n in [s for s in SIZES if s <= 10000]:
        v = column(kind, n, seed=n)
        ec, en = oc.compress_int_array(v), on.compress_int_array(v)
        assert ec == en, (kind, n)
        mc, dc = oc.frame_of_reference_encode(v)
        mn, dn = on.frame_of_reference_encode(v)
        assert mc == mn and np.array_equal(dc, dn)
        assert np.array_equal(oc.zig_zag_encode(dc), on.zig_zag_encode(dn))
        st, back = oc.decompress_int_array(ec, n)
        assert st == 0 and np.array_equal(back, v)
        st, back = on.decompress_int_array(ec, n)
        assert st == 0 and np.array_equal(back, v)


@pytest.mark.parametrize("name,o", ORACLES)
def test_reference_unit_vectors(name, o):
    # compression.rs:337-367
    v = np.array([100, 102, 98, 101, 99, 103, 97], dtype=np.int64)
    mid, d = o.frame_of_reference_encode(v)
    assert np.array_equal(o.frame_of_reference_decode(mid, d), v)
    v = np.array([-5, -1, 0, 1, 5, -100, 100], dtype=np.int64)
    assert np.array_equal(o.zig_zag_decode(o.zig_zag_encode(v)), v)
    u = np.array([0, 1, 2, 15, 255, 1000], dtype=np.uint64)
    st, back = o.bp64_unpack(o.bp64_pack(u), u.size)
    assert st == 0 and np.array_equal(back, u)


@pytest.mark.parametrize("name,o", ORACLES)
def test_reference_pipeline_assertions(name, o):
    # tests/compression-pipeline.rs:7-90, 208-279
    ts = np.array([4102444800000 + i * 3600000 for i in range(168)], dtype=np.int64)
    mid, d = o.frame_of_reference_encode(ts)
    assert int(np.abs(d).max()).bit_length() <= 30
    z = o.zig_zag_encode(d)
    assert int(z.max()).bit_length() <= 32
    packed = o.bp64_pack(z)
    assert ts.size * 8 / len(packed) > 1.5
    e = o.compress_int_array(ts)
    assert ts.size * 8 / len(e) > 1.5
    # the figures docs/compression-pipeline.md:119-127 states for this week of hourly timestamps:
    # 29-bit deltas, 30-bit zig-zag values, 30 bits per value, "~53% compression from bit-packing alone"
    assert int(np.abs(d).max()).bit_length() == 29 and int(z.max()).bit_length() == 30 and packed[0] == 30
    assert round(100 * (1 - (len(packed) - 1) / (ts.size * 8))) == 53
    # and docs/compression-pipeline.md:77-82 for its three-value example: 23 bits instead of 64, "~64% size reduction"
    assert round(100 * (1 - o.bp64_pack(np.array([7199999, 0, 6400000], dtype=np.uint64))[0] / 64)) == 64
    # tests/compression-pipeline.rs:93-121
    small = o.bp64_pack(np.arange(64, dtype=np.uint64) * 1000)
    large = o.bp64_pack((1 << 40) + np.arange(64, dtype=np.uint64))
    assert 64 * 8 / len(small) > 2.0 and 64 * 8 / len(large) < 1.1
    # tests/compression-pipeline.rs:172-204: scattered timestamps exceed 32 bits
    _, sd = o.frame_of_reference_encode(np.array([1577836800000, 4102444800000, 7258118400000], dtype=np.int64))
    assert int(np.abs(sd).max()).bit_length() > 32


@pytest.mark.parametrize("name,o", ORACLES)
def test_empty_and_error_conventions(name, o):
    assert o.compress_int_array(np.zeros(0, dtype=np.int64)) == b""          # compression.rs:272-274
    assert o.bp64_pack(np.zeros(0, dtype=np.uint64)) == b""                  # :118-120
    st, out = o.decompress_int_array(b"", 5)                                  # :296-298
    assert st == 0 and out.size == 0
    st, out = o.bp64_unpack(b"", 5)                                           # :182-184
    assert st == 0 and out.size == 0
    assert o.decompress_int_array(b"\0" * 11, 1)[0] == 1                      # :300-304
    e = o.compress_int_array([1, 2, 3, 4, 5])
    assert o.decompress_int_array(e[:-1], 5)[0] == 2                          # :312-316
    # packed_len == 0: bp64_unpack sees an empty slice -> empty Vec
    st, out = o.decompress_int_array(b"\x05" + b"\0" * 11, 4)
    assert st == 0 and out.size == 0
    # width byte 0 -> zeros -> every value is the midpoint                    :187-189
    st, out = o.decompress_int_array((9).to_bytes(8, "little") + (1).to_bytes(4, "little") + b"\0", 3)
    assert st == 0 and out.tolist() == [9, 9, 9]
    # raw fallback with missing bytes -> zero                                  :196-205
    raw = (0).to_bytes(8, "little") + (1 + 8 + 3).to_bytes(4, "little") + bytes([40]) + (6).to_bytes(8, "little") + b"\1\1\1"
    st, out = o.decompress_int_array(raw, 3)
    assert st == 0 and out.tolist() == [3, 0, 0]
    # width bytes 65..255 take the raw path too (>= 32)
    raw200 = (0).to_bytes(8, "little") + (9).to_bytes(4, "little") + bytes([200]) + (8).to_bytes(8, "little")
    st, out = o.decompress_int_array(raw200, 1)
    assert st == 0 and out.tolist() == [4]
    # trailing bytes after 12 + packed_len are ignored
    st, out = o.decompress_int_array(e + b"\xff\xff", 5)
    assert st == 0 and out.tolist() == [1, 2, 3, 4, 5]
    # short packed payload: the reference underflows; both restatements report ERR_TRUNCATED
    short = e[:8] + (len(e) - 12 - 1).to_bytes(4, "little") + e[12:-1]
    assert o.decompress_int_array(short, 5)[0] == 3


def test_splitmix_agrees():
    i = np.arange(1000, dtype=np.uint64)
    a = on.sm(0x5EED0002, i)
    assert all(int(a[k]) == oc.sm(0x5EED0002, k) for k in range(0, 1000, 37))
    v = oc.gen_narrow(0x5EED0002, 0, 1000)
    assert np.array_equal(v, 1_700_000_000_000 + (a % np.uint64(4000)).astype(np.int64))


def test_headline_column_width_is_12():
    v = oc.gen_narrow(0x5EED0002, 0, 1 << 16)
    e = oc.compress_int_array(v)
    assert e[12] == 12 and len(e) == 13 + (v.size * 12 + 7) // 8


@pytest.mark.parametrize("kind", KINDS)
def test_timed_baseline_with_stable_sort_gives_the_oracles_bytes(kind):
    """bench.py times the CPU encode with an inlined-compare stable sort (oracle/zbra_baseline_sort.cpp):
    same bytes as the statement-by-statement port, whatever the sort."""
    for n in (1, 2, 7, 4096, 10001):
        v = column(kind, n, seed=n + 3)
        want = oc.compress_int_array(v)
        buf = np.empty(13 + 8 * n, dtype=np.uint8)
        got = oc.compress_int_array_stable_sort_into(np.ascontiguousarray(v), buf)
        assert buf[:got].tobytes() == want, (kind, n)


def test_big_integer_statement_of_the_layout(golden):
    """A third formulation with no bit buffer (oracle/zbra_oracle_spec.py: the payload is
    sum(z[i] << (i * w)) as little-endian bytes) gives the bytes of every golden vector, the 643
    bytes of vector M included, and of proptest-shaped random arrays (compression.rs:378-406)."""
    from oracle import zbra_oracle_spec as osp
    for vec in golden["pipeline"] + golden["stages"]:
        values = [int(v) for v in vec["values"]]
        assert osp.compress_int_array(values).hex() == vec["hex"], vec["name"]
        assert osp.decompress_int_array(bytes.fromhex(vec["hex"]), len(values)) == values, vec["name"]
    rng = np.random.default_rng(2026)
    for trial in range(300):
        n = int(rng.integers(0, 100))
        shape = trial % 4
        if shape == 0:
            v = rng.integers(-(1 << 63), (1 << 63) - 1, n, dtype=np.int64, endpoint=True)   # any::<i64>()
        elif shape == 1:
            v = rng.integers(0, 1000, n, dtype=np.int64)                                      # 0..1000
        elif shape == 2:
            bits = int(rng.integers(1, 63))
            v = rng.integers(-(1 << bits), 1 << bits, n, dtype=np.int64) + int(rng.integers(-(1 << 40), 1 << 40))
        else:
            v = np.where(rng.integers(0, 2, n) == 0, np.int64(-(1 << 63)), np.int64((1 << 63) - 1))  # wrap-around in the deltas
        e = oc.compress_int_array(v)
        assert e == osp.compress_int_array(v.tolist()), (trial, v.tolist())
        assert e == on.compress_int_array(v), trial
        assert osp.decompress_int_array(e, n) == v.tolist(), trial
