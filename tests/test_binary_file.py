""".zbra container tests.  CPU part: the host-only walker (zbra_hip_file_index) on images built
with the oracle as codec — offsets, roles, kinds, error cases, every truncation.  GPU part: the
writer and the two-phase reader (index + whole-file decode) against oracle-built images."""
import numpy as np
import pytest

from conftest import column


class OracleCodec:
    """Test-side codec: the CPU oracle stands in for the GPU when expected images are built."""
    @staticmethod
    def compress_int_array(v):
        from oracle import zbra_oracle_c as oc
        return oc.compress_int_array(v)


def _time_series_table(zb, rows=16, seed=0):
    """The CPU-runnable configuration: array<struct{attribute_id:int, entity_id:int,
    time:time_milliseconds, value:int}> (struct keys alphabetical, as the reference CLI emits)."""
    rng = np.random.default_rng(seed)
    return zb.ArrayTable(zb.Struct([
        ("attribute_id", zb.Int(rng.integers(1, 8, rows, dtype=np.int64))),
        ("entity_id", zb.Int(1000 + rng.integers(0, 50, rows, dtype=np.int64))),
        ("time", zb.Int(1_700_000_000_000 + np.arange(rows, dtype=np.int64) * 60_000, "TimeMilliseconds")),
        ("value", zb.Int(rng.integers(-500, 500, rows, dtype=np.int64))),
    ]))


def _rich_table(zb, rows, seed):
    """Every column kind that reaches the int codec (binary.rs:366-434) plus the ones that do not."""
    rng = np.random.default_rng(seed)
    tags = rng.choice(np.array([0, 1, 2, 7], dtype=np.uint32), rows, p=[0.5, 0.25, 0.15, 0.1])
    counts = [int((tags == t).sum()) for t in (0, 1, 2, 7)]
    lengths = rng.integers(0, 5, rows).astype(np.uint64)
    strlens = rng.integers(0, 4, rows).astype(np.uint64)
    return zb.ArrayTable(zb.Struct([
        ("ints", zb.Int(column("narrow", rows, seed))),
        ("wide", zb.Int(column("full", rows, seed + 1))),
        ("dbl", zb.Double(rng.normal(size=rows))),
        ("str", zb.Binary(strlens, bytes(rng.integers(97, 123, int(strlens.sum()), dtype=np.uint8)))),
        ("arr", zb.Array(lengths, zb.Int(column("small_signed", int(lengths.sum()), seed + 2), "Date"))),
        ("sum", zb.Enum(tags, [("a", 0, zb.Unit(counts[0])),
                               ("b", 1, zb.Struct([("x", zb.Int(column("clustered", counts[1], 3))),
                                                   ("y", zb.Int(column("time_series", counts[1], 4), "TimeMilliseconds"))])),
                               ("c", 2, zb.Reversed(zb.Int(column("w31", counts[2], 5)))),
                               ("d", 7, zb.Nested(np.full(counts[3], 2, dtype=np.uint64),
                                                  zb.MapTable(zb.Int(np.arange(2 * counts[3], dtype=np.int64)),
                                                              zb.Double(np.linspace(0, 1, 2 * counts[3])))))])),
        ("blob", zb.Nested(np.ones(rows, dtype=np.uint64), zb.BinaryTable(b"\x00\x01" * rows))),
    ]))


def _int_arrays_in_order(zb, t, out):
    """(values as the codec sees them, kind, role) in file order, straight from the tree."""
    def col(c):
        if isinstance(c, zb.Int):
            out.append((np.asarray(c.values, dtype=np.int64), 0, 0))
        elif isinstance(c, zb.Double):
            out.append((np.asarray(c.values, dtype=np.float64).view(np.int64), 0, 1))
        elif isinstance(c, zb.Binary):
            out.append((np.asarray(c.lengths, dtype=np.uint64).view(np.int64), 0, 2))
        elif isinstance(c, zb.Array):
            out.append((np.asarray(c.lengths, dtype=np.uint64).view(np.int64), 0, 3)); col(c.element)
        elif isinstance(c, zb.Struct):
            for _, x in c.fields:
                col(x)
        elif isinstance(c, zb.Enum):
            out.append((np.asarray(c.tags, dtype=np.uint32).astype(np.int64), 1, 4))
            for _, _, x in c.variants:
                col(x)
        elif isinstance(c, zb.Nested):
            out.append((np.asarray(c.lengths, dtype=np.uint64).view(np.int64), 0, 5)); tab(c.table)
        elif isinstance(c, zb.Reversed):
            col(c.inner)

    def tab(t):
        if isinstance(t, zb.ArrayTable):
            col(t.column)
        elif isinstance(t, zb.MapTable):
            col(t.key_column); col(t.value_column)
    tab(t)
    return out


def _trees_equal(zb, a, b):
    if type(a) is not type(b):
        return False
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_trees_equal(zb, x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return a.dtype == b.dtype and np.array_equal(a.view(np.uint8) if a.dtype == np.float64 else a,
                                                      b.view(np.uint8) if b.dtype == np.float64 else b)
    if hasattr(a, "__dataclass_fields__"):
        return all(_trees_equal(zb, getattr(a, f), getattr(b, f)) for f in a.__dataclass_fields__)
    return a == b


# ---- CPU: walker only ------------------------------------------------------------------------

def test_index_finds_every_array_in_order():
    from oracle import zbra_oracle_c as oc
    from zbra_amd import binary as zb
    blocks = [zb.Block(300, _rich_table(zb, 300, 1)), zb.Block(16, _time_series_table(zb)), zb.Block(0, _rich_table(zb, 0, 2))]
    img = zb.BinaryFile('{"Array":{}}', blocks).to_bytes(OracleCodec)
    info, arrays = zb.file_index(img)
    expect = []
    for bi, b in enumerate(blocks):
        for item in _int_arrays_in_order(zb, b.table, []):
            expect.append((bi,) + item)
    assert info.block_count == 3 and info.n_arrays == len(expect) and info.end_offset == len(img)
    assert img[info.schema_offset:info.schema_offset + info.schema_len] == b'{"Array":{}}'
    assert img[info.compression_offset:info.compression_offset + info.compression_len].decode() == zb.COMPRESSION_NONE
    assert info.total_values == sum(e[1].size for e in expect)
    for a, (bi, v, kind, role) in zip(arrays, expect):
        assert (a.block, a.count, a.kind, a.role) == (bi, v.size, kind, role)
        assert img[a.offset:a.offset + a.len] == oc.compress_int_array(v)
        assert int.from_bytes(img[a.offset - 8:a.offset - 4], "little") == v.size       # u32 count
        assert int.from_bytes(img[a.offset - 4:a.offset], "little") == a.len            # u32 clen


def test_index_rejects_bad_images():
    from zbra_amd import binary as zb
    img = zb.BinaryFile("{}", [zb.Block(50, _rich_table(zb, 50, 3))]).to_bytes(OracleCodec)
    with pytest.raises(zb.BinaryFormatError):
        zb.file_index(b"||_ZBRA||00002||" + img[16:])          # InvalidMagicNumber
    for cut in list(range(0, 120)) + list(range(120, len(img), 7)):
        with pytest.raises(zb.BinaryFormatError):              # UnexpectedEof at every truncation
            zb.file_index(img[:cut])
    info, arrays = zb.file_index(img)
    first_col_tag = info.blocks_offset + 4 + 4 + 1 + 1         # block_count, row_count, table tag, default
    assert img[first_col_tag] == 5                              # the Struct column
    bad = bytearray(img); bad[first_col_tag] = 9
    with pytest.raises(zb.BinaryFormatError, match=f"offset {first_col_tag}"):
        zb.file_index(bytes(bad))                               # InvalidColumnTag
    bad = bytearray(img); bad[info.blocks_offset + 8] = 3
    with pytest.raises(zb.BinaryFormatError):
        zb.file_index(bytes(bad))                               # InvalidTableTag
    rng = np.random.default_rng(0)
    for _ in range(300):                                        # corrupted bytes never crash the walker
        bad = bytearray(img)
        for p in rng.integers(16, len(img), 3):
            bad[p] = int(rng.integers(0, 256))
        try:
            zb.file_index(bytes(bad))
        except zb.BinaryFormatError:
            pass


def _golden_file():
    import json
    import os
    from conftest import ROOT
    with open(os.path.join(ROOT, "tests", "golden", "known_answers.json")) as f:
        return json.load(f)["file"][0]


def _golden_file_tree(zb, vec):
    values = np.array([int(v) for v in vec["values"]], dtype=np.int64)
    return zb.BinaryFile(vec["schema_json"], [zb.Block(len(values), zb.ArrayTable(zb.Int(values)))], vec["compression_json"])


def test_hand_built_image_of_the_reference_round_trip_test_cpu():
    """The image of binary.rs:708-741 (BinaryFile::new of an Int column 1..5, default compression
    config), typed by hand in scripts/make_golden.py: the writer's tree walk with the oracle as
    codec produces exactly these 228 bytes, and the walker finds the one array in it."""
    from zbra_amd import binary as zb
    vec = _golden_file()
    want = bytes.fromhex(vec["hand"]["hex"])
    assert len(want) == vec["hand"]["len"] == 228
    assert _golden_file_tree(zb, vec).to_bytes(OracleCodec) == want
    info, arrays = zb.file_index(want)
    assert info.block_count == 1 and info.n_arrays == 1 and info.end_offset == len(want)
    assert (arrays[0].count, arrays[0].len, arrays[0].role, arrays[0].kind) == (5, 15, 0, 0)
    assert want[arrays[0].offset:arrays[0].offset + 15].hex() == "030000000000000003000000030b44"
    assert want[info.schema_offset:info.schema_offset + info.schema_len].decode() == vec["schema_json"]
    # a table with byte blobs cannot be written under a zstd config here: zstd is another path
    blob = zb.BinaryFile("{}", [zb.Block(1, zb.ArrayTable(zb.Binary(np.array([2], dtype=np.uint64), b"hi")))], vec["compression_json"])
    with pytest.raises(NotImplementedError):
        blob.to_bytes(OracleCodec)


# ---- GPU: writer + two-phase reader -------------------------------------------------------------

@pytest.mark.gpu
def test_hand_built_image_of_the_reference_round_trip_test_gpu():
    from zbra_amd import binary as zb
    vec = _golden_file()
    want = bytes.fromhex(vec["hand"]["hex"])
    f = _golden_file_tree(zb, vec)
    assert f.to_bytes() == want
    back = zb.BinaryFile.from_bytes(want)
    assert back.schema_json == vec["schema_json"] and back.compression_json == vec["compression_json"]
    assert len(back.blocks) == 1 and back.blocks[0].row_count == 5
    assert _trees_equal(zb, back.blocks[0].table, f.blocks[0].table)


@pytest.mark.gpu
def test_small_time_series_roundtrip():
    from zbra_amd import binary as zb
    t = _time_series_table(zb)
    f = zb.BinaryFile('{"Array":{"default":"Allow"}}', [zb.Block(16, t)])
    img = f.to_bytes()
    assert img == f.to_bytes(OracleCodec)                       # same image as with the CPU codec
    back = zb.BinaryFile.from_bytes(img)
    assert back.schema_json == f.schema_json and back.compression_json == zb.COMPRESSION_NONE
    assert len(back.blocks) == 1 and back.blocks[0].row_count == 16
    assert _trees_equal(zb, back.blocks[0].table, t)


@pytest.mark.gpu
def test_rich_multi_block_roundtrip():
    from zbra_amd import binary as zb
    blocks = [zb.Block(n, _rich_table(zb, n, 10 + i)) for i, n in enumerate([0, 1, 257, 5000, 70001])]
    f = zb.BinaryFile("{}", blocks)
    img = f.to_bytes()
    assert img == f.to_bytes(OracleCodec)
    back = zb.BinaryFile.from_bytes(img)
    assert len(back.blocks) == len(blocks)
    for a, b in zip(back.blocks, blocks):
        assert a.row_count == b.row_count and _trees_equal(zb, a.table, b.table)


@pytest.mark.gpu
def test_file_decode_groups_and_statuses():
    """Enough bytes that the whole-file decode splits into several groups on both lanes, plus a
    corrupted array whose status must not disturb the others."""
    import ctypes as C
    from oracle import zbra_oracle_c as oc
    from zbra_amd import _lib
    from zbra_amd import binary as zb
    cols = [("full", 1 << 21), ("narrow", 1 << 22), ("w40", (1 << 21) + 5), ("small_signed", 3), ("full", 1 << 22),
            ("time_series", 1 << 22), ("w32", 1 << 21), ("narrow", 12345)]
    blocks = [zb.Block(n, zb.ArrayTable(zb.Int(column(k, n, 50 + i)))) for i, (k, n) in enumerate(cols)]
    img = bytearray(zb.BinaryFile("{}", blocks).to_bytes(OracleCodec))
    info, arrays = zb.file_index(bytes(img))
    assert info.n_arrays == len(cols) and len(img) > (100 << 20)
    # corrupt array 3's packed_len so that it claims more bytes than its frame holds
    a3 = arrays[3]
    img[a3.offset + 8:a3.offset + 12] = (a3.len).to_bytes(4, "little")
    outs = [np.full(arrays[i].count, -1, dtype=np.int64) for i in range(info.n_arrays)]
    ptrs = (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
    statuses = (C.c_int * len(outs))()
    buf = np.frombuffer(bytes(img), dtype=np.uint8)
    st = _lib.lib.zbra_hip_file_decode(buf.ctypes.data_as(C.c_void_p), buf.size, arrays, len(outs), ptrs, statuses)
    assert st == _lib.ERR_SHORT_PACKED and list(statuses) == [0, 0, 0, _lib.ERR_SHORT_PACKED, 0, 0, 0, 0]
    assert (outs[3] == -1).all()
    for i, (k, n) in enumerate(cols):
        if i != 3:
            assert np.array_equal(outs[i], column(k, n, 50 + i)), i


def test_walker_fuzz_under_sanitizers(tmp_path):
    """The walker is host code that reads untrusted bytes: random trees, every truncation and random
    corruptions under AddressSanitizer + UBSan (tests/cpp/fuzz_walker.cpp)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fuzz_walker")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-static-libasan",
                    "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "fuzz_walker.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe, "4000", "7"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "4000 valid images indexed exactly" in r.stdout


@pytest.mark.gpu
def test_frames_that_produce_nothing():
    """decompress_int_array returns an empty Vec for empty data and for packed_len == 0 whatever the
    frame's count says (compression.rs:296-298, 182-184): the reader hands back empty vectors and
    sizes nothing by such a count (a corrupt frame may say 2^32 - 1)."""
    from zbra_amd import binary as zb

    class HollowCodec:
        @staticmethod
        def compress_int_array(values):
            return bytes(8) + (0).to_bytes(4, "little")      # midpoint 0, packed_len 0

    real = column("narrow", 1000, 3)
    blocks = [zb.Block(9, zb.ArrayTable(zb.Struct([("a", zb.Int(np.arange(9, dtype=np.int64)))]))),
              zb.Block(1000, zb.ArrayTable(zb.Int(real)))]
    hollow = bytearray(zb.BinaryFile("{}", blocks[:1]).to_bytes(HollowCodec))
    info, arrays = zb.file_index(bytes(hollow))
    assert info.n_arrays == 1 and arrays[0].count == 9 and arrays[0].len == 12
    back = zb.BinaryFile.from_bytes(bytes(hollow))
    assert back.blocks[0].table.column.fields[0][1].values.size == 0
    # an empty frame whose count field says 2^32 - 1, next to a real array
    img = bytearray(zb.BinaryFile("{}", [zb.Block(0, zb.ArrayTable(zb.Int(np.zeros(0, dtype=np.int64)))), blocks[1]]).to_bytes())
    info, arrays = zb.file_index(bytes(img))
    assert arrays[0].count == 0 and arrays[0].len == 0
    img[arrays[0].offset - 8:arrays[0].offset - 4] = (0xFFFFFFFF).to_bytes(4, "little")
    back = zb.BinaryFile.from_bytes(bytes(img))
    assert back.blocks[0].table.column.values.size == 0
    assert np.array_equal(back.blocks[1].table.column.values, real)


@pytest.mark.gpu
def test_files_on_every_device_from_one_process():
    """zbra_hip_file_decode_multi: one image per job, jobs dealt to the devices of the box round robin
    (file f on GPU f mod N; one device here means every job runs there, one after another), plus a
    job that names a device that does not exist: it alone fails, the others decode exactly."""
    import ctypes as C
    import torch
    from zbra_amd import _lib
    from zbra_amd import binary as zb
    ndev = torch.cuda.device_count()
    jobs_data = []
    for f in range(5):
        blocks = [zb.Block(n, _rich_table(zb, n, 300 + 10 * f + i)) for i, n in enumerate([3, 4097, 30001 + f])]
        img = zb.BinaryFile("{}", blocks).to_bytes(OracleCodec)
        want = []
        for b in blocks:
            _int_arrays_in_order(zb, b.table, want)
        info, arrays = zb.file_index(img)
        assert info.n_arrays == len(want)
        outs = [np.full(arrays[k].count, 0x55, dtype=np.int64 if arrays[k].kind == _lib.ELEM_I64 else np.uint32) for k in range(info.n_arrays)]
        jobs_data.append((np.frombuffer(img, dtype=np.uint8), arrays, outs, want, (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs]),
                          (C.c_int * len(outs))()))
    jobs = (_lib.FileJob * (len(jobs_data) + 1))()
    for f, (buf, arrays, outs, want, ptrs, statuses) in enumerate(jobs_data):
        jobs[f].device, jobs[f].status = f % ndev, -1
        jobs[f].image, jobs[f].len = buf.ctypes.data, buf.size
        jobs[f].arrays, jobs[f].n = arrays, len(outs)
        jobs[f].outs, jobs[f].statuses = ptrs, statuses
    bad = len(jobs_data)
    jobs[bad].device = ndev + 3
    jobs[bad].image, jobs[bad].len, jobs[bad].n = jobs[0].image, jobs[0].len, 0
    assert _lib.lib.zbra_hip_file_decode_multi(jobs, len(jobs)) == _lib.ERR_INVALID_ARG      # the first failing job's status
    assert jobs[bad].status == _lib.ERR_INVALID_ARG
    for f, (buf, arrays, outs, want, ptrs, statuses) in enumerate(jobs_data):
        assert jobs[f].status == 0 and list(statuses) == [0] * len(outs), f
        for k, (values, kind, role) in enumerate(want):
            expect = values if kind == 0 else values.astype(np.uint32)
            assert np.array_equal(outs[k], expect), (f, k)
    assert _lib.lib.zbra_hip_file_decode_multi(None, 0) == 0
    assert torch.cuda.current_device() == 0                     # the caller's own device is untouched
