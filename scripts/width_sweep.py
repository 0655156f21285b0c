#!/usr/bin/env python3
"""Decode (and encode) rate per bit width on a single 2^27-row column: finds weak paths.  The decode
is timed with the caches emptied before every launch (a 2 GiB fill; a column is decoded once, and
packed bytes that still sit in the 256 MB Infinity Cache from the launch before flatter any kernel
that reads them with plain loads); the back-to-back time is printed beside it."""
import statistics
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from zbra_amd import device, workloads

dev = torch.device("cuda", 0)
n = 1 << 27
idx = torch.arange(n, dtype=torch.int64, device=dev)
rnd = workloads.sm(0xABCDEF, idx)
del idx
flush = torch.empty(1 << 31, dtype=torch.uint8, device=dev)
for bits in [1, 2, 3, 4, 5, 6, 8, 12, 16, 20, 24, 28, 31, 32, 40, 64]:
    if bits == 64:
        v = rnd.clone()
    else:
        v = (rnd & ((1 << bits) - 1)) - (1 << (bits - 1)) + 1_000_000_007
    enc = torch.empty(device.encode_bound(n), dtype=torch.uint8, device=dev)
    res = device.new_results(1, dev)
    device.encode_dev(v, enc, res)
    r = device.read_results(res)[0]
    out = torch.empty_like(v)

    def t(fn, reps=20, warm=20):
        for _ in range(warm):  # the clocks drop while the host checks results: bring them back before timing
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    dwarm = t(lambda: device.decode_dev(enc, n, out, None, length=r.out_count))
    assert torch.equal(out, v)

    def cold_once():
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        device.decode_dev(enc, n, out, None, length=r.out_count)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    dms = statistics.median(cold_once() for _ in range(11))
    ems = t(lambda: device.encode_dev(v, enc, res), 10, 10)
    algo = r.out_count + 8 * n
    print(f"amp bits {bits:2d} -> w {r.width:2d}: decode {dms:.4f} ms {algo / dms / 1e6:6.0f} GB/s ({algo / dms / 8e9:.3f}; back to back {dwarm:.4f} ms {algo / dwarm / 8e9:.3f}) | "
          f"encode {ems:.4f} ms in {8 * n / ems / 1e6:6.0f} GB/s algo ({(16 * n + r.out_count) / ems / 8e9:.3f})")
