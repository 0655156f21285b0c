#!/usr/bin/env python3
"""Device-resident batch encode and decode of 2^26 rows cut into equal arrays: from one array of
2^26 rows to 65536 arrays of 2^10.  Shows what a submission of many short arrays pays over one long
one (tiles that are part empty, per-array set-up, the array table).  Decode is timed cold."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from zbra_amd import _lib, device, workloads

dev = torch.device("cuda", 0)
total_log2 = 26
n_all = 1 << total_log2
kinds = sys.argv[1:] or ["i64", "u32"]
flush = torch.empty(1 << 31, dtype=torch.uint8, device=dev)
for kind in kinds:
    es, ek = (8, _lib.ELEM_I64) if kind == "i64" else (4, _lib.ELEM_U32)
    col = workloads.narrow_column(0x5EED0002, 0, n_all, dev)
    if kind == "u32":
        col = (col - 1_700_000_000_000).to(torch.int32)
    out = torch.empty_like(col)
    print(f"== {kind}: 2^{total_log2} rows of width 12 as equal arrays: encode ms (fraction of two reads + one write at 8 TB/s) | decode ms cold (fraction)")
    for log_m in (26, 24, 22, 20, 18, 16, 14, 12, 10):
        m = 1 << log_m
        k = n_all >> log_m
        bound = device.encode_bound(m)
        stride = (bound + 255) // 256 * 256
        pe = torch.empty(k * stride, dtype=torch.uint8, device=dev)
        res = device.new_results(k, dev)
        ed = device.make_encode_descs([(col.data_ptr() + i * m * es, m, pe.data_ptr() + i * stride, bound, ek) for i in range(k)])
        for _ in range(12):  # the clocks drop while the host checks results: bring them back before timing
            device.encode_batch_dev(ed, res)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            device.encode_batch_dev(ed, res)
        e1.record(); torch.cuda.synchronize()
        ems = e0.elapsed_time(e1) / 10
        rs = device.read_results(res)
        enc_bytes = sum(int(r.out_count) for r in rs)
        dd = device.make_descs([(pe.data_ptr() + i * stride, int(rs[i].out_count), m, out.data_ptr() + i * m * es, ek) for i in range(k)])
        for _ in range(2):
            device.decode_batch_dev(dd, None)
        ts = []
        for _ in range(7):
            flush.fill_(1)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); device.decode_batch_dev(dd, None); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        assert torch.equal(out, col)
        dms = statistics.median(ts)
        print(f"{k:6d} arrays of 2^{log_m:2d}: encode {ems:8.4f} ms ({(2 * es * n_all + enc_bytes) / ems / 8e9:.3f}) | decode {dms:8.4f} ms ({(es * n_all + enc_bytes) / dms / 8e9:.3f})")
