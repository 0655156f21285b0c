#!/usr/bin/env python3
"""Encode time of the headline column with the library given by ZBRA_HIP_LIB (one process per
library: run it once per build, alternating, and compare)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from zbra_amd import device, workloads

dev = torch.device("cuda", 0)
n = 1 << 28
v = workloads.narrow_column(0x5EED0002, 0, n, dev)
out = torch.empty(device.encode_bound(n), dtype=torch.uint8, device=dev)
res = device.new_results(1, dev)
for _ in range(3):
    device.encode_dev(v, out, res)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        device.encode_dev(v, out, res)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
print(os.environ.get("ZBRA_HIP_LIB", "default"), "encode ms: median %.4f min %.4f" % (sorted(ts)[2], min(ts)))
