#!/usr/bin/env python3
"""Encode time of a batch of equal device-resident columns, framed into one buffer.
Usage: enc_batch_rate.py ARRAYS ROWS_LOG2 [w40|wide|narrow|time|full]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from zbra_amd import device

na = int(sys.argv[1]) if len(sys.argv) > 1 else 96
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 22)
kind = sys.argv[3] if len(sys.argv) > 3 else "w40"
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(5)
cols = []
for a in range(na):
    if kind == "narrow":
        v = 1_700_000_000_000 + torch.randint(0, 4000, (n,), dtype=torch.int64, device=dev, generator=g)
    elif kind == "wide":
        v = torch.randint(-(1 << 24), 1 << 24, (n,), dtype=torch.int64, device=dev, generator=g)
    elif kind == "time":
        v = 1_640_995_200_000 + torch.arange(n, dtype=torch.int64, device=dev) * 1000 + a
    elif kind == "full":
        v = torch.randint(-(1 << 62), 1 << 62, (n,), dtype=torch.int64, device=dev, generator=g)
    else:
        v = torch.randint(-(1 << 38), 1 << 38, (n,), dtype=torch.int64, device=dev, generator=g)
    cols.append(v)
out = torch.empty(na * (device.encode_bound(n) + 8), dtype=torch.uint8, device=dev)
offs = torch.empty(na + 1, dtype=torch.int64, device=dev)
res = device.new_results(na, dev)
descs = device.make_encode_descs((v.data_ptr(), n, 0, 0, 0) for v in cols)
for _ in range(3):
    device.encode_frames_dev(descs, out, offs, res)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        device.encode_frames_dev(descs, out, offs, res)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5)
r = device.read_results(res)
assert all(x.status == 0 for x in r)
ms = sorted(ts)[2]
print("%d x 2^%d %s (w %d): encode %.3f ms, %.0f GB/s of input" % (na, n.bit_length() - 1, kind, r[0].width, ms, na * n * 8 / ms / 1e6))
