#!/usr/bin/env python3
"""Encodes (and decodes) one device-resident column of scripts/width_sweep.py's generator at ONE
amplitude, a few times: run it under rocprofv3 (--kernel-trace --stats, or --pmc FETCH_SIZE /
WRITE_SIZE in runs of their own) to see where an encode of that width spends its time and bytes.
Usage: enc_width_prof.py <amp_bits> [rows_log2] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from zbra_amd import device, workloads

bits = int(sys.argv[1])
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 27)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
idx = torch.arange(n, dtype=torch.int64, device=dev)
rnd = workloads.sm(0xABCDEF, idx)
del idx
v = rnd if bits == 64 else (rnd & ((1 << bits) - 1)) - (1 << (bits - 1)) + 1_000_000_007
enc = torch.empty(device.encode_bound(n), dtype=torch.uint8, device=dev)
res = device.new_results(1, dev)
for _ in range(reps):
    device.encode_dev(v, enc, res)
torch.cuda.synchronize()
r = device.read_results(res)[0]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    device.encode_dev(v, enc, res)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
algo = 16 * n + r.out_count
print(f"amp bits {bits} n 2^{n.bit_length() - 1} -> w {r.width}, E {r.out_count} B; encode {ms:.4f} ms; algorithmic {algo} B "
      f"({algo / ms / 8e9:.3f} of 8 TB/s)")
