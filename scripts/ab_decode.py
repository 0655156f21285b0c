#!/usr/bin/env python3
"""Interleaved A/B timing of decode kernel shapes in ONE process (variants x rounds), on the
headline column and optionally a u32 tag column.  Prints median and min per variant."""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+", help="tile:threads:nt:direct:xcd")
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rows-log2", type=int, default=28)
    ap.add_argument("--tags", action="store_true", help="u32 tag column instead of the int64 column")
    ap.add_argument("--amp-bits", type=int, default=0, help="int64 column of this amplitude (the width sweep's generator) instead")
    args = ap.parse_args()
    import torch

    from zbra_amd import _lib, device, workloads
    dev = torch.device("cuda", 0)
    n = 1 << args.rows_log2
    if args.tags:
        tags, _ = workloads.enum_block(0, n, dev)
        values = tags
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
    out = torch.empty_like(values)
    algo = r.out_count + values.numel() * values.element_size()
    times = {v: [] for v in args.variants}
    for rnd in range(args.rounds):
        for v in args.variants:
            t, th, nt, d, x = (int(k) for k in v.split(":"))
            assert _lib.lib.zbra_tuning_set_decode_variant(t, th, nt, d, x) == 0, v
            for _ in range(2):
                device.decode_dev(encoded, n, out, None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                device.decode_dev(encoded, n, out, None)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / args.reps)
            if rnd == 0:
                assert torch.equal(out, values), v
    for v in args.variants:
        med, mn = statistics.median(times[v]), min(times[v])
        print(f"{v:22s} median {med:.4f} ms  min {mn:.4f} ms  {algo / med / 1e6:7.0f} GB/s  ({algo / med / 1e6 / 8000:.3f} of peak)")


if __name__ == "__main__":
    main()
