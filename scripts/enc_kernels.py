#!/usr/bin/env python3
"""Encodes one device-resident column a few times (run it under rocprofv3 --kernel-trace --stats to
see the per-kernel times).  Usage: enc_kernels.py narrow|wide|time|w40|heavy|bool|ends|full [rows_log2]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from zbra_amd import device

kind = sys.argv[1] if len(sys.argv) > 1 else "w40"
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 27)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(3)
if kind == "narrow":
    v = 1_700_000_000_000 + torch.randint(0, 4000, (n,), dtype=torch.int64, device=dev, generator=g)
elif kind == "wide":
    v = torch.randint(-(1 << 24), 1 << 24, (n,), dtype=torch.int64, device=dev, generator=g)
elif kind == "time":  # sorted: one timestamp a second
    v = 1_640_995_200_000 + torch.arange(n, dtype=torch.int64, device=dev) * 1000
elif kind == "heavy":  # 45/3/3/49 %: a dense bracket of three keys
    u = torch.randint(0, 100, (n,), dtype=torch.int64, device=dev, generator=g)
    v = torch.where(u < 45, 0, torch.where(u < 48, 1, torch.where(u < 51, 2, 3)))
elif kind == "ends":  # i64::MIN and i64::MAX among small values: the span wraps, the width comes from the OR pass
    v = torch.randint(0, 1000, (n,), dtype=torch.int64, device=dev, generator=g)
    v[n // 3] = -(1 << 63)
    v[2 * n // 3] = (1 << 63) - 1
elif kind == "bool":
    v = torch.randint(0, 2, (n,), dtype=torch.int64, device=dev, generator=g)
elif kind == "w40":
    v = torch.randint(-(1 << 38), 1 << 38, (n,), dtype=torch.int64, device=dev, generator=g)
else:
    v = torch.randint(-(1 << 62), 1 << 62, (n,), dtype=torch.int64, device=dev, generator=g)
out = torch.empty(device.encode_bound(n), dtype=torch.uint8, device=dev)
res = device.new_results(1, dev)
for _ in range(5):
    device.encode_dev(v, out, res)
torch.cuda.synchronize()
r = device.read_results(res)[0]
print(kind, "n", n, "width", r.width, "bytes", r.out_count)
