#!/usr/bin/env python3
"""Writes tests/golden/known_answers.json.

The reference ships no golden bytes and cannot be run here, so each vector has
two layers:
  * "hand": midpoint / bit width / encoded length (and, for short vectors, the
    full hex) derived BY HAND from the text of zbra-core/src/compression.rs —
    typed in below, not computed;
  * "hex": the full encoded bytes, written only when the C restatement and the
    numpy restatement produce identical bytes that also satisfy the hand layer.
The script refuses to write the file if any of the three disagree.

Inputs come from the reference's own tests:
  compression.rs:339,347,355,363           unit-test vectors
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
    ("I", [0, -6818581005699659960, -6818581005699659960, -2404791031155115849, -6818581005699659960, 0],
     -4611686018427387905, 64, 61, None, "proptest-regressions/compression.txt:10"),
    ("J", [I64_MIN, I64_MAX], I64_MIN, 1, 14, "0000000000000080020000000102",
     "truncating (not floor) division in the even-length median"),
    ("K", [1577836800000, 4102444800000, 7258118400000], 4102444800000, 43, 37, None,
     "tests/compression-pipeline.rs:172-176 (raw fallback)"),
    ("L", [1, 2, 3, 4, 5, 10, 20, 30], 4, 6, 19, "04000000000000000700000006c510000203d2",
     "zbra-cli/src/main.rs:384"),
    ("M", [4102444800000 + i * 3600000 for i in range(168)], 4102745400000, 30, 643, None,
     "tests/compression-pipeline.rs:211-212"),
]

# bp64_pack-only vectors: name, input, hand (w, packed_len), origin
PACK = [
    ("N", [i * 1000 for i in range(64)], 16, 129, "tests/compression-pipeline.rs:96"),
    ("O", [(1 << 40) + i for i in range(64)], 41, 513, "tests/compression-pipeline.rs:101"),
    ("P", [0, 1, 2, 15, 255, 1000], 10, 9, "compression.rs:355"),
]


# worked example of docs/compression-pipeline.md:27-33 (median, deltas), :52-56 (zig-zag), :77-82 (width)
STAGES = [
    ("Q", [4102444800000, 4102448400000, 4102451600000],
     {"median": 4102448400000, "deltas": [-3600000, 0, 3200000], "zigzag": [7199999, 0, 6400000], "w": 23},
     "docs/compression-pipeline.md:27-33, 52-56, 77-82"),
]


def main():
    out = {"note": "hand = derived by hand from compression.rs; hex = bytes on which the C and numpy "
                   "restatements agree. Not produced by running the reference.",
           "pipeline": [], "pack": [], "stages": []}
    for name, values, mid, w, length, hexs, origin in PIPELINE:
        ec = oc.compress_int_array(values)
        en = on.compress_int_array(values)
        assert ec == en, f"{name}: C and numpy restatements disagree"
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
    for name, values, pub, origin in STAGES:
        for o in (oc, on):
            mid, deltas = o.frame_of_reference_encode(values)
            assert int(mid) == pub["median"] and [int(d) for d in deltas] == pub["deltas"], name
            zz = o.zig_zag_encode(deltas)
            assert [int(z) for z in zz] == pub["zigzag"], name
            assert o.bp64_pack(zz)[0] == pub["w"], name
        out["stages"].append({"name": name, "origin": origin, "values": [str(v) for v in values],
                              "published": {"median": str(pub["median"]), "deltas": [str(d) for d in pub["deltas"]],
                                            "zigzag": [str(z) for z in pub["zigzag"]], "w": pub["w"]},
                              "hex": oc.compress_int_array(values).hex()})
    path = os.path.join(ROOT, "tests", "golden", "known_answers.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(out["pipeline"]), "pipeline +", len(out["pack"]), "pack +", len(out["stages"]), "stage vectors")


if __name__ == "__main__":
    main()
