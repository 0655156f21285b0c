#!/usr/bin/env python3
"""One encode of a 2^27-row column with `bits` of amplitude; run under rocprofv3 --kernel-trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zbra_amd import device, workloads
bits = int(sys.argv[1])
dev = torch.device("cuda", 0)
n = 1 << 27
rnd = workloads.sm(0xABCDEF, torch.arange(n, dtype=torch.int64, device=dev))
v = rnd if bits == 64 else (rnd & ((1 << bits) - 1)) - (1 << (bits - 1)) + 1_000_000_007
enc = torch.empty(device.encode_bound(n), dtype=torch.uint8, device=dev)
res = device.new_results(1, dev)
for _ in range(2):
    device.encode_dev(v, enc, res)
torch.cuda.synchronize()
print(device.read_results(res)[0].width)
