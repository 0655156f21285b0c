#!/usr/bin/env python3
"""Rate of the Enum row split (zbra_hip_enum_rows_dev) on a device-resident tag column with the
sum-typed table's tag distribution.  Usage: enum_rate.py [rows_log2]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from zbra_amd import device

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(4)
u = torch.randint(0, 100, (n,), dtype=torch.int32, device=dev, generator=g)
tags = torch.where(u < 50, 0, torch.where(u < 75, 1, torch.where(u < 90, 2, 7))).to(torch.int32)
del u
variant = torch.empty(n, dtype=torch.int32, device=dev)
rank = torch.empty(n, dtype=torch.int64, device=dev)
counts = torch.empty(4, dtype=torch.int64, device=dev)
res = device.new_results(1, dev)
for _ in range(3):
    device.enum_rows_dev(tags, [0, 1, 2, 7], variant, rank, counts, res)
torch.cuda.synchronize()
assert device.read_results(res)[0].status == 0
assert counts.tolist() == [int((tags == t).sum()) for t in (0, 1, 2, 7)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    device.enum_rows_dev(tags, [0, 1, 2, 7], variant, rank, counts, res)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(json.dumps({"what": "2^%d rows, 4 variants" % (n.bit_length() - 1), "ms": round(ms, 4), "rows_per_s": round(n / ms * 1e3),
                  "algorithmic_GBps": round(16.0 * n / ms / 1e6, 1), "hbm_frac": round(16.0 * n / ms / 1e6 / 8000, 4)}))
