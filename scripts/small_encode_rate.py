#!/usr/bin/env python3
"""Device-resident encode of ONE array per call, back to back on one stream: time per call by size
(width 12).  Under scripts/prof_kernels.sh the per-kernel times show what of a call is kernels and
what is the table copy, the workspace clearing and the kernel boundaries.  Usage: small_encode_rate.py [LOG2 ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from zbra_amd import device, workloads

dev = torch.device("cuda", 0)
for lg in [int(a) for a in sys.argv[1:]] or [13, 16, 20, 24]:
    n = (1 << lg) + 1
    v = 1_700_000_000_000 + workloads.sm(0x5EED0002, torch.arange(n, dtype=torch.int64, device=dev)) % 4000
    enc = torch.empty(device.encode_bound(n), dtype=torch.uint8, device=dev)
    res = device.new_results(1, dev)
    for _ in range(20):
        device.encode_dev(v, enc, res)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            device.encode_dev(v, enc, res)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 40)
    print(f"one array of 2^{lg} + 1 rows: encode {sorted(ts)[2] * 1e3:8.1f} us per call", flush=True)
