#!/usr/bin/env python3
"""Decode kernel shapes timed COLD, interleaved in one process: a 2 GiB fill empties the caches
before every single launch, so a variant cannot win by finding its packed bytes still in the
256 MB Infinity Cache from the launch before (a column is decoded once).  The back-to-back time
is printed beside it.  Variants as in ab_decode.py: tile:threads:nt:direct:order[:warm_MiB[:warm_bits]]."""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+", help="tile:threads:nt:direct:order")
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--rows-log2", type=int, default=27)
    ap.add_argument("--arrays", type=int, default=1, help="cut the column into this many equal arrays, decoded as one batch")
    ap.add_argument("--tags", action="store_true", help="u32 tag column instead of an int64 column")
    ap.add_argument("--amp-bits", type=int, default=0, help="int64 column of this amplitude (the width sweep's generator); 0 = the headline column")
    ap.add_argument("--ones-percent", type=int, default=-1, help="boolean int64 column with this share of ones (100: a constant column): width 1 above 50, width 2 below")
    args = ap.parse_args()
    import torch

    from zbra_amd import _lib, device, workloads
    dev = torch.device("cuda", 0)
    n = 1 << args.rows_log2
    if args.tags:
        values, _ = workloads.enum_block(0, n, dev)
    elif args.ones_percent >= 0:
        values = (workloads.umod(workloads.sm(0xABCDEF, torch.arange(n, dtype=torch.int64, device=dev)), 100) < args.ones_percent).to(torch.int64)
    elif args.amp_bits:
        rnd = workloads.sm(0xABCDEF, torch.arange(n, dtype=torch.int64, device=dev))
        values = rnd if args.amp_bits == 64 else (rnd & ((1 << args.amp_bits) - 1)) - (1 << (args.amp_bits - 1)) + 1_000_000_007
        del rnd
    else:
        values = workloads.narrow_column(0x5EED0002, 0, n, dev)
    enc = torch.empty(device.encode_bound(n), dtype=torch.uint8, device=dev)
    res = device.new_results(1, dev)
    device.encode_dev(values, enc, res)
    r = device.read_results(res)[0]
    encoded = enc[:r.out_count].clone()
    del enc
    out = torch.empty_like(values)
    flush = torch.empty(1 << 31, dtype=torch.uint8, device=dev)
    algo = r.out_count + values.numel() * values.element_size()

    descs = None
    if args.arrays > 1:  # the same column as `arrays` equal arrays decoded in one batched submission
        m = n // args.arrays
        parts, r_all = [], device.new_results(args.arrays, dev)
        pe = torch.empty(args.arrays * device.encode_bound(m), dtype=torch.uint8, device=dev)
        device.encode_batch_dev(device.make_encode_descs([(values.data_ptr() + k * m * values.element_size(), m, pe.data_ptr() + k * device.encode_bound(m),
                                                           device.encode_bound(m), _lib.ELEM_I64 if values.element_size() == 8 else _lib.ELEM_U32)
                                                          for k in range(args.arrays)]), r_all)
        rs = device.read_results(r_all)
        descs = device.make_descs([(pe.data_ptr() + k * device.encode_bound(m), int(rs[k].out_count), m, out.data_ptr() + k * m * values.element_size(),
                                    _lib.ELEM_I64 if values.element_size() == 8 else _lib.ELEM_U32) for k in range(args.arrays)])
        algo = sum(int(x.out_count) for x in rs) + values.numel() * values.element_size()

    def once(cold):
        if cold:
            flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if descs is not None:
            device.decode_batch_dev(descs, None)
        else:
            device.decode_dev(encoded, n, out, None)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    cold = {v: [] for v in args.variants}
    warm = {v: [] for v in args.variants}
    for rnd in range(args.rounds):
        for v in args.variants:
            t, th, nt, d, x, *rest = (int(k) for k in v.split(":"))
            assert _lib.lib.zbra_tuning_set_decode_variant(t, th, nt, d, x) == 0, v
            # optional sixth and seventh field: packed MiB per warmed chunk (0 or absent = warm-up path off), widest array that takes it
            _lib.lib.zbra_tuning_set_decode_warm(C.c_uint64((rest[0] if rest else 0) << 20), C.c_uint32(rest[1] if len(rest) > 1 else 6), C.c_uint64(8 << 20))
            if rnd == 0:
                once(False)
                assert torch.equal(out, values), v
            cold[v].append(once(True))
            once(False)
            warm[v].append(once(False))
    print(f"width {r.width}, 2^{args.rows_log2} rows, {int(r.out_count)} encoded bytes")
    for v in args.variants:
        c, w = statistics.median(cold[v]), statistics.median(warm[v])
        print(f"{v:22s} cold {c:.4f} ms ({algo / c / 8e9:.3f} of peak)   back to back {w:.4f} ms ({algo / w / 8e9:.3f})")


if __name__ == "__main__":
    main()
