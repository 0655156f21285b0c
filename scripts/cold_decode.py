#!/usr/bin/env python3
"""Decode time with the caches emptied before every launch (a 2 GiB fill in between), next to the
usual back-to-back timing whose input may still sit in the 256 MB Infinity Cache from the launch
before.  Usage: cold_decode.py [tags|narrow] [rows_log2]   (ZBRA_HIP_LIB selects the build)"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from zbra_amd import device, workloads

what = sys.argv[1] if len(sys.argv) > 1 else "tags"
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else (29 if what == "tags" else 28))
dev = torch.device("cuda", 0)
if what == "tags":
    u = workloads.umod(workloads.sm(0x5EED0004, torch.arange(n, dtype=torch.int64, device=dev)), 100)
    values = torch.where(u < 50, 0, torch.where(u < 75, 1, torch.where(u < 90, 2, 7))).to(torch.int32)
    del u
else:
    values = workloads.narrow_column(0x5EED0002, 0, n, dev)
enc = torch.empty(device.encode_bound(n), dtype=torch.uint8, device=dev)
res = device.new_results(1, dev)
device.encode_dev(values, enc, res)
r = device.read_results(res)[0]
enc = enc[:int(r.out_count)].clone()
out = torch.empty_like(values)
del values
dres = device.new_results(1, dev)
flush = torch.empty(1 << 31, dtype=torch.uint8, device=dev)


def once(cold):
    if cold:
        flush.fill_(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    device.decode_dev(enc, n, out, dres)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for _ in range(3):
    once(False)
warm = statistics.median(once(False) for _ in range(15))
cold = statistics.median(once(True) for _ in range(15))
print("%s 2^%d rows, width %d, %d encoded bytes (%s): back to back %.4f ms, caches emptied first %.4f ms"
      % (what, n.bit_length() - 1, r.width, enc.numel(), os.environ.get("ZBRA_HIP_LIB", "default build"), warm, cold))
