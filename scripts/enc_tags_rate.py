#!/usr/bin/env python3
"""Encode time of enum tag columns (u32, tags {0,1,2,7} at 50/25/15/10 %): 128 arrays of 2^22 rows
in one submission, and one array of 2^29 rows.  With the argument `wide`: 30 random bits per value."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from zbra_amd import _lib as L
from zbra_amd import device, workloads

dev = torch.device("cuda", 0)
n_all = 1 << 29
u = workloads.sm(0x5EED0004, torch.arange(n_all, dtype=torch.int64, device=dev)) % 100
tags = torch.where(u < 50, 0, torch.where(u < 75, 1, torch.where(u < 90, 2, 7))).to(torch.int32)
del u
if len(sys.argv) > 1 and sys.argv[1] == "wide":  # a u32 column of 30 random bits instead: the wide-bracket route
    tags = (workloads.sm(0x5EED0104, torch.arange(n_all, dtype=torch.int64, device=dev)) & 0x3FFFFFFF).to(torch.int32)


def timed(fn):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    return sorted(ts)[2]


na, n = 128, 1 << 22
outs = torch.empty(na * (device.encode_bound(n) + 8), dtype=torch.uint8, device=dev)
res = device.new_results(na, dev)
per = device.encode_bound(n) + 8
descs = device.make_encode_descs((tags[a * n:(a + 1) * n].data_ptr(), n, outs.data_ptr() + a * per, per, L.ELEM_U32) for a in range(na))
ms = timed(lambda: device.encode_batch_dev(descs, res))
r = device.read_results(res)
assert all(x.status == 0 for x in r)
algo = n_all * 8 + sum(x.out_count for x in r)
print(f"  128 x 2^22 tags (w {r[0].width}): encode {ms:.4f} ms, {algo / ms / 8e9:.3f} of the two-reads-one-write roof")
del outs
out = torch.empty(device.encode_bound(n_all), dtype=torch.uint8, device=dev)
res1 = device.new_results(1, dev)
ms = timed(lambda: device.encode_dev(tags, out, res1))
r = device.read_results(res1)[0]
assert r.status == 0
print(f"  one 2^29-row tag column (w {r.width}): encode {ms:.4f} ms, {(n_all * 8 + r.out_count) / ms / 8e9:.3f} of the roof")
