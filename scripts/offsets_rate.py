#!/usr/bin/env python3
"""Rate of the fused lengths -> offsets decode against decoding the lengths and scanning them
afterwards (torch.cumsum), on 2^28 short-string lengths (0..49) resident in HBM."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from zbra_amd import device

dev = torch.device("cuda", 0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
g = torch.Generator(device=dev)
g.manual_seed(1)
lengths = torch.randint(0, 50, (n,), dtype=torch.int64, device=dev, generator=g)
enc = torch.empty(device.encode_bound(n), dtype=torch.uint8, device=dev)
res = device.new_results(1, dev)
device.encode_dev(lengths, enc, res)
r = device.read_results(res)[0]
e = enc[: r.out_count].clone()
offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
out = torch.empty(n, dtype=torch.int64, device=dev)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


flush = torch.empty(1 << 31, dtype=torch.uint8, device=dev)


def cold(fn, reps=9):
    """One call at a time behind a 2 GiB fill: nothing of the last call is left in the caches."""
    import statistics
    ts = []
    for _ in range(reps):
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


t_fused = timed(lambda: device.decode_offsets_dev(e, n, offs, None))
t_fused_cold = cold(lambda: device.decode_offsets_dev(e, n, offs, None))
want = torch.cumsum(lengths, 0)
assert offs[0] == 0 and torch.equal(offs[1:], want)
t_dec = timed(lambda: device.decode_dev(e, n, out, None))
t_scan = timed(lambda: torch.cumsum(out, 0, out=offs[1:]))
algo = 2 * int(r.out_count) + 8 * (n + 1)
print(json.dumps({"what": f"2^{n.bit_length() - 1} lengths, width {r.width}", "fused_ms": round(t_fused, 4), "fused_cold_ms": round(t_fused_cold, 4), "hbm_frac_cold": round(algo / t_fused_cold / 1e6 / 8000, 3),
                  "fused_algorithmic_GBps": round(algo / t_fused / 1e6, 1), "hbm_frac": round(algo / t_fused / 1e6 / 8000, 3),
                  "decode_ms": round(t_dec, 4), "torch_cumsum_ms": round(t_scan, 4), "decode_plus_cumsum_ms": round(t_dec + t_scan, 4)}))
