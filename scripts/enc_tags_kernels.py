#!/usr/bin/env python3
"""Encodes one device-resident u32 tag column (tags {0,1,2,7}, 50/25/15/10 %) a few times; run under
scripts/prof_kernels.sh for per-kernel times.  Usage: enc_tags_kernels.py [rows_log2]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from zbra_amd import device, workloads

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
dev = torch.device("cuda", 0)
tags, _ = workloads.enum_block(0, n, dev)
out = torch.empty(device.encode_bound(n), dtype=torch.uint8, device=dev)
res = device.new_results(1, dev)
for _ in range(5):
    device.encode_dev(tags, out, res)
torch.cuda.synchronize()
r = device.read_results(res)[0]
print("tags n", n, "dtype", tags.dtype, "width", r.width, "bytes", r.out_count)
