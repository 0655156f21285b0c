#!/usr/bin/env python3
"""Writes tests/golden/known_answers.json.

The reference ships no golden bytes and cannot be run here, so each vector has
two layers:
  * "hand": midpoint / bit width / encoded length (and, for short vectors, the
    full hex) derived BY HAND from the text of zbra-core/src/compression.rs —
    typed in below, not computed;
  * "hex": the full encoded bytes, written only when the C restatement, the
    numpy restatement and the big-integer statement of the layout
    (oracle/zbra_oracle_spec.py) produce identical bytes that also satisfy the
    hand layer.
The script refuses to write the file if any of them disagree.

Inputs come from the reference's own tests:
  compression.rs:339,347,355,363           unit-test vectors
  binary.rs:708-741                        the int-array file of the round-trip test, as a whole image
  proptest-regressions/compression.txt:7-11 overflow seeds
  tests/compression-pipeline.rs:9-16,96,101,172-176 deterministic inputs
  zbra-cli/src/main.rs:384                  numbers.json example
A third layer, "stages", holds the one worked example the reference PUBLISHES with its
intermediate results (docs/compression-pipeline.md:27-82: median, deltas, zig-zag values, bit
width): numbers stated by the reference's own documentation, not derived here.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import zbra_oracle_c as oc  # noqa: E402
from oracle import zbra_oracle_np as on  # noqa: E402
from oracle import zbra_oracle_spec as osp  # noqa: E402

I64_MIN, I64_MAX = -(1 << 63), (1 << 63) - 1

# name, input, hand-derived (mid, w, len, hex-or-None), origin
PIPELINE = [
    ("A", [1, 2, 3, 4, 5], 3, 3, 15, "030000000000000003000000030b44", "odd length; smallest case"),
    ("B", [100, 102, 98, 101, 99, 103, 97, -5, -1, 0], 98, 8, 23,
     "62000000000000000b0000000804080006020a01cdc5c3", "compression.rs:363"),
    ("C", [0], 0, 1, 14, "0000000000000000020000000100", "all-zero deltas give width 1, not 0"),
    ("D", [7, 7, 7, 7], 7, 1, 14, "0700000000000000020000000100", "constant column"),
    ("E", [1 << 62], 1 << 62, 1, 14, "0000000000000040020000000100", "proptest-regressions/compression.txt:7"),
    ("F", [0, -(1 << 60)], -(1 << 59), 61, 29,
     "00000000000000f8110000003d0000000000000010ffffffffffffff0f", "proptest-regressions/compression.txt:8"),
    ("G", [0, -((1 << 60) - 1)], -(1 << 59), 61, 29,
     "00000000000000f8110000003d0000000000000010fdffffffffffff0f", "proptest-regressions/compression.txt:11"),
    ("H", [5823061857575085472, -3400310179279690336, -3400310179279690336], -3400310179279690336, 64, 37,
     "a0a5e355e8abcfd01900000040ffffffffffffffff00000000000000000000000000000000",
     "proptest-regressions/compression.txt:9"),
    # a = -6818581005699659960, b = -2404791031155115849, b - a = 4413789974544544111, trunc(/2) =
    # 2206894987272272055, mid = a + that = -(2^62 + 1) = 0xBFFFFFFFFFFFFFFF; zig-zag of 0 - mid =
    # 2 * (2^62 + 1) = 0x8000000000000002; of a - mid = -2206894987272272055: 4413789974544544109 =
    # 0x3D40EE98B9E8716D; of b - mid = +2206894987272272056: that + 3; 64 bits -> raw, 1 + 6 * 8 = 0x31
    ("I", [0, -6818581005699659960, -6818581005699659960, -2404791031155115849, -6818581005699659960, 0],
     -4611686018427387905, 64, 61,
     "ffffffffffffffbf" "31000000" "40" "0200000000000080" "6d71e8b998ee403d" "6d71e8b998ee403d" "7071e8b998ee403d"
     "6d71e8b998ee403d" "0200000000000080", "proptest-regressions/compression.txt:10"),
    ("J", [I64_MIN, I64_MAX], I64_MIN, 1, 14, "0000000000000080020000000102",
     "truncating (not floor) division in the even-length median"),
    # mid = 4102444800000 = 0x03BB2CC3D800; deltas -2524608000000, 0, +3155673600000; zig-zag
    # 5049215999999 = 0x04979CB9DFFF, 0, 6311347200000 = 0x05BD799BA000 (43 bits -> raw, 1 + 3 * 8 = 0x19)
    ("K", [1577836800000, 4102444800000, 7258118400000], 4102444800000, 43, 37,
     "00d8c32cbb030000" "19000000" "2b" "ffdfb99c97040000" "0000000000000000" "00a09b79bd050000",
     "tests/compression-pipeline.rs:172-176 (raw fallback)"),
    ("L", [1, 2, 3, 4, 5, 10, 20, 30], 4, 6, 19, "04000000000000000700000006c510000203d2",
     "zbra-cli/src/main.rs:384"),
    ("M", [4102444800000 + i * 3600000 for i in range(168)], 4102745400000, 30, 643, None,
     "tests/compression-pipeline.rs:211-212"),
    # sorted 97..103, median 100; deltas 0 2 -2 1 -1 3 -3 -> zig-zag 0 4 3 2 1 6 5, 3 bits each:
    # 0 | 4<<3 | 3<<6 | 2<<9 | 1<<12 | 6<<15 | 5<<18 = 1512672 = 0x1714E0
    ("R", [100, 102, 98, 101, 99, 103, 97], 100, 3, 16, "6400000000000000" "04000000" "03" "e01417",
     "compression.rs:339 through compress_int_array"),
    # sorted -100 -5 -1 0 1 5 100, median 0; zig-zag 9 1 0 2 10 199 200 -> 8 bits each
    ("S", [-5, -1, 0, 1, 5, -100, 100], 0, 8, 20, "0000000000000000" "08000000" "08" "090100020ac7c8",
     "compression.rs:347 through compress_int_array"),
]

# bp64_pack-only vectors: name, input, hand (w, packed_len), origin
PACK = [
    ("N", [i * 1000 for i in range(64)], 16, 129, "tests/compression-pipeline.rs:96"),
    ("O", [(1 << 40) + i for i in range(64)], 41, 513, "tests/compression-pipeline.rs:101"),
    ("P", [0, 1, 2, 15, 255, 1000], 10, 9, "compression.rs:355"),
]


# worked example of docs/compression-pipeline.md:27-33 (median, deltas), :52-56 (zig-zag), :77-82 (width)
# The hex is derived by hand from the published numbers: mid 4102448400000 = 0x03BB2CFAC680; 23-bit
# fields at bits 0, 23, 46: 7199999 = 0x6DDCFF fills bytes 0-2, the zero field leaves bytes 3-4 clear,
# 6400000 = 0x61A800 has its bits 10-17 (0x6A) in byte 7 and bits 18-22 (0x18) in byte 8; 9 payload bytes.
STAGES = [
    ("Q", [4102444800000, 4102448400000, 4102451600000],
     {"median": 4102448400000, "deltas": [-3600000, 0, 3200000], "zigzag": [7199999, 0, 6400000], "w": 23},
     "80c6fa2cbb030000" "0a000000" "17" "ffdc6d000000006a18",
     "docs/compression-pipeline.md:27-33, 52-56, 77-82"),
]

# A whole .zbra image typed by hand: the file of the reference's own round-trip test
# (zbra-core/src/binary.rs:708-741): BinaryFile::new(Array<Int>, values 1..5), hence the default
# compression config.  Pieces in file order (binary.rs:103-130, 184-192, 220-224, 285-294, 639-645);
# the JSON is serde's externally tagged form of TableSchema (logical.rs:8-24, 27-33) and
# CompressionConfig (compression.rs:13-41), 92 and 67 bytes; E is vector A.
FILE_IMAGE = [
    ("ascii", "||_ZBRA||00001||"),
    ("hex", "5c000000" "5c000000"),
    ("ascii", '{"Array":{"default":"Allow","element":{"Int":{"default":"Allow","encoding":{"Int":"Int"}}}}}'),
    ("hex", "43000000" "43000000"),
    ("ascii", '{"binary_data":{"Zstd":{"level":3}},"strings":{"Zstd":{"level":3}}}'),
    ("hex", "01000000"),            # block_count
    ("hex", "05000000"),            # Block.row_count
    ("hex", "01" "00"),             # Table::Array, Default::Allow
    ("hex", "01" "00" "00" "00"),   # Column::Int, Default::Allow, Encoding::Int, IntEncoding::Int
    ("hex", "05000000" "0f000000"), # count, clen
    ("hex", "030000000000000003000000030b44"),
]
FILE_IMAGE_LEN = 228


def main():
    out = {"note": "hand = derived by hand from compression.rs; hex = bytes on which the C and numpy "
                   "restatements agree. Not produced by running the reference.",
           "pipeline": [], "pack": [], "stages": [], "file": []}
    for name, values, mid, w, length, hexs, origin in PIPELINE:
        ec = oc.compress_int_array(values)
        en = on.compress_int_array(values)
        assert ec == en, f"{name}: C and numpy restatements disagree"
        assert ec == osp.compress_int_array(values), f"{name}: the big-integer statement of the layout disagrees"
        assert osp.decompress_int_array(ec, len(values)) == values, name
        got_mid = int.from_bytes(ec[0:8], "little", signed=True)
        assert got_mid == mid, (name, got_mid, mid)
        assert ec[12] == w, (name, ec[12], w)
        assert len(ec) == length, (name, len(ec), length)
        if hexs is not None:
            assert ec.hex() == hexs, (name, ec.hex(), hexs)
        # round trip through both decoders
        rc, back = oc.decompress_int_array(ec, len(values))
        assert rc == 0 and back.tolist() == values, name
        rn, backn = on.decompress_int_array(en, len(values))
        assert rn == 0 and backn.tolist() == values, name
        out["pipeline"].append({"name": name, "origin": origin, "values": [str(v) for v in values],
                                "hand": {"mid": str(mid), "w": w, "len": length, "hex": hexs},
                                "hex": ec.hex()})
    for name, values, w, plen, origin in PACK:
        pc = oc.bp64_pack(values)
        pn = on.bp64_pack(values)
        assert pc == pn, name
        assert pc[0] == w and len(pc) == plen, (name, pc[0], len(pc))
        rc, back = oc.bp64_unpack(pc, len(values))
        assert rc == 0 and back.tolist() == values
        out["pack"].append({"name": name, "origin": origin, "values": [str(v) for v in values],
                            "hand": {"w": w, "packed_len": plen}, "hex": pc.hex()})
    for name, values, pub, hand_hex, origin in STAGES:
        for o in (oc, on):
            mid, deltas = o.frame_of_reference_encode(values)
            assert int(mid) == pub["median"] and [int(d) for d in deltas] == pub["deltas"], name
            zz = o.zig_zag_encode(deltas)
            assert [int(z) for z in zz] == pub["zigzag"], name
            assert o.bp64_pack(zz)[0] == pub["w"], name
        out["stages"].append({"name": name, "origin": origin, "values": [str(v) for v in values],
                              "published": {"median": str(pub["median"]), "deltas": [str(d) for d in pub["deltas"]],
                                            "zigzag": [str(z) for z in pub["zigzag"]], "w": pub["w"]},
                              "hand": {"hex": hand_hex},
                              "hex": oc.compress_int_array(values).hex()})
        assert oc.compress_int_array(values).hex() == hand_hex == on.compress_int_array(values).hex(), name
        assert osp.compress_int_array(values).hex() == hand_hex, name
    image = b"".join(p.encode() if kind == "ascii" else bytes.fromhex(p) for kind, p in FILE_IMAGE)
    assert len(image) == FILE_IMAGE_LEN, len(image)
    out["file"] = [{"name": "T", "origin": "zbra-core/src/binary.rs:708-741 (test_binary_roundtrip_simple)",
                    "schema_json": FILE_IMAGE[2][1], "compression_json": FILE_IMAGE[4][1], "values": ["1", "2", "3", "4", "5"],
                    "hand": {"len": FILE_IMAGE_LEN, "hex": image.hex()}}]
    path = os.path.join(ROOT, "tests", "golden", "known_answers.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(out["pipeline"]), "pipeline +", len(out["pack"]), "pack +", len(out["stages"]), "stage vectors")


if __name__ == "__main__":
    main()
