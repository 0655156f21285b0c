#!/usr/bin/env python3
"""The batched forms of the Enum row split and of the lengths -> offsets decode against their
single-array forms on the same rows: B arrays of 2^r rows in ONE submission, B single calls one after
another, and one array of B * 2^r rows.  Defaults are the sum-typed configuration's share of one
GPU: 128 Blocks of 2^22 rows.  Every timing is taken COLD (a 2 GiB fill between calls) and back to
back.  Usage: batch_forms_rate.py [blocks] [rows_log2]"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from zbra_amd import _lib, device

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
R = int(sys.argv[2]) if len(sys.argv) > 2 else 22
n, N = 1 << R, B << R
dev = torch.device("cuda", 0)
flush = torch.empty(1 << 31, dtype=torch.uint8, device=dev)


def back_to_back(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def cold(fn, reps=7):
    ts = []
    for _ in range(reps):
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


# ---- Enum row split ---------------------------------------------------------------------------
g = torch.Generator(device=dev)
g.manual_seed(4)
u = torch.randint(0, 100, (N,), dtype=torch.int32, device=dev, generator=g)
tags = torch.where(u < 50, 0, torch.where(u < 75, 1, torch.where(u < 90, 2, 7))).to(torch.int32)
del u
variant = torch.empty(N, dtype=torch.int32, device=dev)
rank = torch.empty(N, dtype=torch.int64, device=dev)
counts = torch.empty((B, 4), dtype=torch.int64, device=dev)
res = device.new_results(B, dev)
VT = [0, 1, 2, 7]
parts = [(tags[b * n:(b + 1) * n], variant[b * n:(b + 1) * n], rank[b * n:(b + 1) * n], counts[b]) for b in range(B)]


def enum_batch():
    device.enum_rows_batch_dev(parts, VT, res)


def enum_singles():
    for t, v, r, c in parts:
        device.enum_rows_dev(t, VT, v, r, c, None)


def enum_whole():
    device.enum_rows_dev(tags, VT, variant, rank, counts[0], None)


enum_batch()
torch.cuda.synchronize()
assert all(r.status == 0 for r in device.read_results(res))
want_rank, want_counts = rank.clone(), counts.clone()
assert want_counts.sum().item() == N and torch.equal(want_counts[3], torch.stack([(parts[3][0] == t).sum() for t in VT]))
enum_singles()
torch.cuda.synchronize()
assert torch.equal(rank, want_rank) and torch.equal(counts, want_counts)   # the batch ranks every Block on its own, as the single calls do
line = {"what": f"{B} arrays of 2^{R} rows ({N} rows in all)", "enum_rows": {}}
for name, fn in (("batch_one_submission", enum_batch), ("single_calls", enum_singles), ("one_array_of_all_rows", enum_whole)):
    c, b2b = cold(fn), back_to_back(fn)
    line["enum_rows"][name] = {"cold_ms": round(c, 4), "back_to_back_ms": round(b2b, 4), "hbm_frac_cold": round(16.0 * N / c / 1e6 / 8000, 3)}
del variant, rank, want_rank, tags, parts

# ---- lengths -> offsets ------------------------------------------------------------------------
g.manual_seed(1)
lengths = torch.randint(0, 50, (N,), dtype=torch.int64, device=dev, generator=g)
cap = device.encode_bound(n)
enc = torch.empty(B * (cap + 3), dtype=torch.uint8, device=dev)
eres = device.new_results(B, dev)
device.encode_batch_dev(device.make_encode_descs([(lengths.data_ptr() + 8 * b * n, n, enc.data_ptr() + b * (cap + 3), cap, _lib.ELEM_I64)
                                                  for b in range(B)]), eres)
ers = device.read_results(eres)
assert all(r.status == 0 for r in ers)
whole = torch.empty(device.encode_bound(N), dtype=torch.uint8, device=dev)
wres = device.new_results(1, dev)
device.encode_dev(lengths, whole, wres)
wr = device.read_results(wres)[0]
offs = torch.empty(B * (n + 1), dtype=torch.int64, device=dev)
odescs = device.make_descs([(enc.data_ptr() + b * (cap + 3), ers[b].out_count, n, offs.data_ptr() + 8 * b * (n + 1), _lib.ELEM_I64) for b in range(B)])
ores = device.new_results(B, dev)
offs_whole = torch.empty(N + 1, dtype=torch.int64, device=dev)


def off_batch():
    device.decode_offsets_batch_dev(odescs, ores)


def off_singles():
    for b in range(B):
        device.decode_offsets_dev(enc[b * (cap + 3):], n, offs[b * (n + 1):], None, length=ers[b].out_count)


def off_whole():
    device.decode_offsets_dev(whole, N, offs_whole, None, length=wr.out_count)


off_batch()
torch.cuda.synchronize()
assert all(r.status == 0 and r.out_count == n for r in device.read_results(ores))
got = offs.view(B, n + 1)
assert (got[:, 0] == 0).all() and torch.equal(got[:, 1:], torch.cumsum(lengths.view(B, n), 1))
enc_bytes = sum(r.out_count for r in ers)
algo = 2 * enc_bytes + 8 * (N + B)
line["decode_offsets"] = {"width": ers[0].width}
for name, fn in (("batch_one_submission", off_batch), ("single_calls", off_singles), ("one_array_of_all_rows", off_whole)):
    c, b2b = cold(fn), back_to_back(fn)
    line["decode_offsets"][name] = {"cold_ms": round(c, 4), "back_to_back_ms": round(b2b, 4), "hbm_frac_cold": round(algo / c / 1e6 / 8000, 3)}
print(json.dumps(line))
