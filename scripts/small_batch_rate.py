#!/usr/bin/env python3
"""Device-resident batched decode of a few arrays per call (a Block's worth: what a reader that goes
Block by Block submits): time per call, back to back on one stream, for 2..64 arrays of 2^18 and
2^22 rows (width 12).  Usage: small_batch_rate.py [LABEL]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from zbra_amd import _lib as L
from zbra_amd import device, workloads

label = sys.argv[1] if len(sys.argv) > 1 else ""
dev = torch.device("cuda", 0)
for lg in (18, 22):
    n = 1 << lg
    v = 1_700_000_000_000 + workloads.sm(0x5EED0002, torch.arange(n, dtype=torch.int64, device=dev)) % 4000
    enc = torch.empty(device.encode_bound(n), dtype=torch.uint8, device=dev)
    res1 = device.new_results(1, dev)
    device.encode_dev(v, enc, res1)
    elen = int(device.read_results(res1)[0].out_count)
    for na in (2, 4, 8, 16, 64):
        outs = [torch.empty(n, dtype=torch.int64, device=dev) for _ in range(na)]
        encs = [enc.clone() for _ in range(na)]
        descs = device.make_descs([(e.data_ptr(), elen, n, o.data_ptr(), L.ELEM_I64) for e, o in zip(encs, outs)])
        res = device.new_results(na, dev)
        for _ in range(20):
            device.decode_batch_dev(descs, res)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                device.decode_batch_dev(descs, res)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 50)
        assert torch.equal(outs[-1], v)
        ms = sorted(ts)[2]
        print(f"{label:>10s} {na:3d} arrays of 2^{lg} rows: {ms * 1e3:8.1f} us per call, {na * (elen + 8 * n) / ms / 8e9:.3f} of the roof", flush=True)
