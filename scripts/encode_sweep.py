#!/usr/bin/env python3
"""Encode time per value amplitude on one 2^27-row column (plus the headline column at 2^28 rows and
a sorted timestamp column), encode only: the A/B of encoder builds (ZBRA_HIP_LIB picks the build).
Usage: encode_sweep.py [LABEL]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from zbra_amd import device, workloads

label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("ZBRA_HIP_LIB", "default"))
dev = torch.device("cuda", 0)


def timed(fn, reps=10, warm=10):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[1]


def one(name, v):
    n = v.numel()
    enc = torch.empty(device.encode_bound(n), dtype=torch.uint8, device=dev)
    res = device.new_results(1, dev)
    device.encode_dev(v, enc, res)
    r = device.read_results(res)[0]
    assert r.status == 0
    ms = timed(lambda: device.encode_dev(v, enc, res))
    print(f"{label:>12s} {name:>10s} w {r.width:2d}: encode {ms:.4f} ms ({(16 * n + r.out_count) / ms / 8e9:.3f} of the roof)", flush=True)


n = 1 << 27
rnd = workloads.sm(0xABCDEF, torch.arange(n, dtype=torch.int64, device=dev))
for bits in (6, 12, 17, 20, 25, 29, 32, 40, 64):
    one(f"amp {bits}", rnd.clone() if bits == 64 else (rnd & ((1 << bits) - 1)) - (1 << (bits - 1)) + 1_000_000_007)
one("sorted ms", 1_640_995_200_000 + torch.arange(n, dtype=torch.int64, device=dev) * 1000)
del rnd
one("headline", 1_700_000_000_000 + workloads.sm(0x5EED0002, torch.arange(1 << 28, dtype=torch.int64, device=dev)) % 4000)
