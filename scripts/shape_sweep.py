#!/usr/bin/env python3
"""Encode and decode time of one 2^26-row column per value distribution and element kind: every
amplitude from 1 to 33 bits plus a set of shapes (sorted, constant, skewed tags, heavy ends, ...).
An anomaly finder: a route of the encoder's selection or of the decoder that is slow for some
distribution shows as an outlier among its neighbours.  Decode is timed cold (2 GiB fill first)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from zbra_amd import device, workloads

dev = torch.device("cuda", 0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 26)
idx = torch.arange(n, dtype=torch.int64, device=dev)
rnd = workloads.sm(0xC0FFEE, idx)
flush = torch.empty(1 << 31, dtype=torch.uint8, device=dev)


def shapes():
    for bits in list(range(1, 21)) + [24, 28, 31, 32, 33]:
        yield f"uniform {bits:2d} bits", (rnd & ((1 << bits) - 1)) - (1 << (bits - 1)) + 1_000_000_007
    u = workloads.umod(rnd, 100)
    yield "tags 50/25/15/10 %", torch.where(u < 50, 0, torch.where(u < 75, 1, torch.where(u < 90, 2, 7)))
    yield "heavy ends 45/3/3/49 %", torch.where(u < 45, 0, torch.where(u < 48, 1, torch.where(u < 51, 2, 3)))
    yield "boolean 50/50", u & 1
    yield "boolean 99/1", (u == 0).to(torch.int64)
    yield "constant", torch.full_like(idx, 42)
    yield "sorted, step 1000", idx * 1000 + 1_700_000_000_000
    yield "sorted, step 1", idx.clone()
    yield "runs of 64 equal", (idx >> 6) % 50000 + 1000
    yield "20 equally likely values", workloads.umod(rnd, 20)
    yield "outlier in narrow data", torch.where(idx == n // 3, 1 << 40, workloads.umod(rnd, 1000))
    yield "geometric (small mostly)", (rnd & 0xFFFF).to(torch.float64).add(1).log2().neg().add(16).mul(4).to(torch.int64)


def time_one(v):
    enc = torch.empty(device.encode_bound(v.numel()), dtype=torch.uint8, device=dev)
    res = device.new_results(1, dev)
    for _ in range(5):
        device.encode_dev(v, enc, res)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        device.encode_dev(v, enc, res)
    e1.record(); torch.cuda.synchronize()
    ems = e0.elapsed_time(e1) / 10
    r = device.read_results(res)[0]
    out = torch.empty_like(v)
    ts = []
    for _ in range(7):
        flush.fill_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); device.decode_dev(enc, v.numel(), out, None, length=int(r.out_count)); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    assert torch.equal(out, v)
    dms = statistics.median(ts)
    es = v.element_size()
    return r.width, ems, (2 * es * v.numel() + r.out_count) / ems / 8e9, dms, (es * v.numel() + r.out_count) / dms / 8e9


for kind, dt in (("int64", torch.int64), ("u32", torch.int32)):
    print(f"== {kind}, 2^{n.bit_length() - 1} rows: width | encode ms (fraction of two reads + one write at 8 TB/s) | decode ms cold (fraction)")
    for name, v in shapes():
        if dt == torch.int32:
            if int(v.max()) >= 1 << 31 or int(v.min()) < -(1 << 31):
                continue
            v = v.to(torch.int32)
        w, ems, ef, dms, df = time_one(v.contiguous())
        print(f"{name:28s} w {w:2d} | encode {ems:.4f} ms ({ef:.3f}) | decode {dms:.4f} ms ({df:.3f})")
