#!/usr/bin/env python3
"""One array at the largest count a frame can carry, 2^32 - 1 values (`values.len() as u32`,
zbra-core/src/binary.rs:640), through every device-resident entry point: int64 and u32 encode and
decode, the fused lengths -> offsets decode and the Enum row split, each checked on the device
(round trip, exact median by counting, running sum against torch.cumsum in pieces, ranks against
a per-variant cumsum).  Also the `packed.len() as u32` truncation of compression.rs:288: an array
whose packed bytes pass 4 GiB gets the low 32 bits in its header, and the decoder then finds the
payload short.  Needs about 110 GB of HBM; prints one line per check."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from zbra_amd import _lib, device

dev = torch.device("cuda", 0)
N = (1 << 32) - 1
STEP = 1 << 28


def fill(v, fn):
    for i0 in range(0, v.numel(), STEP):
        i1 = min(v.numel(), i0 + STEP)
        idx = torch.arange(i0, i1, dtype=torch.int64, device=dev)
        v[i0:i1] = fn(idx).to(v.dtype)
        del idx


def exact_median_ok(v, mid, n):
    below = above = 0
    for i0 in range(0, n, STEP):
        part = v[i0:i0 + STEP]
        below += int((part < mid).sum()); above += int((part > mid).sum())
    return below <= (n - 1) // 2 and above <= n // 2


def roundtrip(name, dt, fn, width_in):
    v = torch.empty(N, dtype=dt, device=dev)
    fill(v, fn)
    enc = torch.empty(13 + 2 * N, dtype=torch.uint8, device=dev)
    res = device.new_results(1, dev)
    device.encode_dev(v, enc, res)
    r = device.read_results(res)[0]
    assert r.status == 0 and r.width in width_in and r.out_count == 13 + (N * r.width + 7) // 8, (r.status, r.width, r.out_count)
    assert exact_median_ok(v, r.midpoint, N), r.midpoint
    out = torch.empty_like(v)
    dres = device.new_results(1, dev)
    device.decode_dev(enc, N, out, dres, length=int(r.out_count))
    d = device.read_results(dres)[0]
    assert d.status == 0 and d.out_count == N
    assert all(torch.equal(out[i0:i0 + STEP], v[i0:i0 + STEP]) for i0 in range(0, N, STEP))
    print(json.dumps({"check": name, "n": N, "width": r.width, "encoded_bytes": int(r.out_count), "ok": True}), flush=True)
    return v, enc, int(r.out_count)


# 1. int64 values below 100 + a slow drift: width 7 or 8
v, enc, elen = roundtrip("int64 encode + decode", torch.int64, lambda i: (i * 2654435761) % 97 + (i >> 31), (7, 8))

# 1b. the host-pointer calls on the same array: pageable memory in, pageable memory out
import ctypes as C

import numpy as np

v_h = np.empty(N, dtype=np.int64)
for i0 in range(0, N, STEP):
    v_h[i0:i0 + STEP] = v[i0:i0 + STEP].cpu().numpy()
e_h = np.empty(elen + 4096, dtype=np.uint8)
got = C.c_size_t(0)
st = _lib.lib.zbra_hip_encode_i64(v_h.ctypes.data_as(C.c_void_p), N, e_h.ctypes.data_as(C.c_void_p), e_h.size, C.byref(got))
assert st == 0 and got.value == elen, (st, got.value)
assert all(np.array_equal(e_h[i0:min(elen, i0 + STEP)], enc[i0:min(elen, i0 + STEP)].cpu().numpy()) for i0 in range(0, elen, STEP))
d_h = np.empty(N, dtype=np.int64)
cnt = C.c_size_t(0)
st = _lib.lib.zbra_hip_decode_i64(e_h.ctypes.data_as(C.c_void_p), elen, N, d_h.ctypes.data_as(C.c_void_p), C.byref(cnt))
assert st == 0 and cnt.value == N, (st, cnt.value)
assert all(np.array_equal(d_h[i0:i0 + STEP], v_h[i0:i0 + STEP]) for i0 in range(0, N, STEP))
print(json.dumps({"check": "host-pointer int64 encode + decode", "n": N, "ok": True}), flush=True)
del v_h, e_h, d_h

# 1c. a narrow column (width 3 or 4): its decode takes the warm-up route, here in two dozen chunks
roundtrip("int64 encode + decode, narrow (warm-up route)", torch.int64, lambda i: (i * 2654435761) % 5 + (i >> 31) - 2, (3, 4))
torch.cuda.empty_cache()

# 2. the same bytes as lengths: offsets[i + 1] = offsets[i] + v[i]
offs = torch.empty(N + 1, dtype=torch.int64, device=dev)
device.decode_offsets_dev(enc, N, offs, None, length=elen)
carry, ok = 0, int(offs[0]) == 0
for i0 in range(0, N, STEP):
    c = torch.cumsum(v[i0:i0 + STEP], 0) + carry
    ok = ok and torch.equal(offs[i0 + 1:i0 + 1 + c.numel()], c)
    carry = int(c[-1])
    del c
assert ok
print(json.dumps({"check": "lengths -> offsets", "n": N, "total": carry, "ok": True}), flush=True)
del offs, v, enc
torch.cuda.empty_cache()

# 3. u32 tags 0, 1, 2, 7
tag_of = lambda i: torch.tensor([0, 0, 1, 2, 0, 7, 1, 0], dtype=torch.int64, device=dev)[((i * 2654435761) >> 7) & 7]
tags, enc, elen = roundtrip("u32 tag encode + decode", torch.int32, tag_of, (4,))
del enc

# 4. Enum row split of the same tags
vt = [0, 1, 2, 7]
variant = torch.empty(N, dtype=torch.int32, device=dev)
rank = torch.empty(N, dtype=torch.int64, device=dev)
counts = torch.zeros(len(vt), dtype=torch.int64, device=dev)
res = device.new_results(1, dev)
device.enum_rows_dev(tags, vt, variant, rank, counts, res)
assert device.read_results(res)[0].status == 0
seen = [0] * len(vt)
ok = True
for i0 in range(0, N, STEP):
    t = tags[i0:i0 + STEP]
    for k, tag in enumerate(vt):
        m = t == tag
        ok = ok and bool((variant[i0:i0 + STEP][m] == k).all())
        r_want = torch.cumsum(m.to(torch.int64), 0)[m] - 1 + seen[k]
        ok = ok and torch.equal(rank[i0:i0 + STEP][m], r_want)
        seen[k] += int(m.sum())
        del m, r_want
assert ok and counts.tolist() == seen, (counts.tolist(), seen)
print(json.dumps({"check": "enum row split", "n": N, "counts": seen, "ok": True}), flush=True)
del tags, variant, rank
torch.cuda.empty_cache()

# 5. packed bytes past 4 GiB: 2^31 + 5 values of width 20
n = (1 << 31) + 5
v = torch.empty(n, dtype=torch.int64, device=dev)
fill(v, lambda i: (i * 2654435761) % 1000003)
enc = torch.empty(13 + 3 * n, dtype=torch.uint8, device=dev)
res = device.new_results(1, dev)
device.encode_dev(v, enc, res)
r = device.read_results(res)[0]
packed = 1 + (n * r.width + 7) // 8
assert r.status == 0 and r.width == 20 and r.out_count == 12 + packed and packed > 1 << 32
hdr = enc[:13].cpu().numpy().tobytes()
assert int.from_bytes(hdr[8:12], "little") == packed & 0xFFFFFFFF, "packed.len() as u32 (compression.rs:288)"
out = torch.full((16,), -1, dtype=torch.int64, device=dev)
dres = device.new_results(1, dev)
device.decode_dev(enc, n, out.new_empty(n), dres, length=int(r.out_count))
d = device.read_results(dres)[0]
assert d.status == _lib.ERR_TRUNCATED and d.out_count == 0, (d.status, d.out_count)
print(json.dumps({"check": "packed bytes past 4 GiB", "n": n, "width": r.width, "packed_bytes": packed,
                  "header_field": packed & 0xFFFFFFFF, "decode_status": "ZBRA_ERR_TRUNCATED", "ok": True}), flush=True)
