#!/usr/bin/env python3
"""Host batch calls on many tiny arrays (what a file of small Blocks looks like): time of the C calls alone."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from zbra_amd import _lib

na = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(0)
arrs = [1_700_000_000_000 + rng.integers(0, 4000, n, dtype=np.int64) for _ in range(na)]
descs = (_lib.EncodeDesc * na)()
for i, a in enumerate(arrs):
    descs[i].values = a.ctypes.data; descs[i].n = n; descs[i].kind = _lib.ELEM_I64
bound = na * (8 + _lib.lib.zbra_hip_encode_bound(n))
out = np.empty(bound, dtype=np.uint8)
offsets = (C.c_uint64 * (na + 1))()
statuses = (C.c_int * na)()
for rep in range(2):
    t0 = time.perf_counter()
    st = _lib.lib.zbra_hip_encode_frames(descs, na, out.ctypes.data_as(C.c_void_p), bound, offsets, statuses)
    te = time.perf_counter() - t0
    assert st == 0
ddescs = (_lib.ArrayDesc * na)()
outs = np.empty((na, n), dtype=np.int64)
for i in range(na):
    ddescs[i].data = out.ctypes.data + offsets[i] + 8
    ddescs[i].len = offsets[i + 1] - offsets[i] - 8
    ddescs[i].count = n
    ddescs[i].out = outs.ctypes.data + i * n * 8
    ddescs[i].kind = _lib.ELEM_I64
for rep in range(2):
    t0 = time.perf_counter()
    st = _lib.lib.zbra_hip_decode_batch(ddescs, na, statuses)
    td = time.perf_counter() - t0
    assert st == 0
assert all(np.array_equal(outs[i], arrs[i]) for i in range(0, na, 997))
print(f"{na} arrays of {n} values: encode_frames {te * 1e3:.1f} ms, decode_batch {td * 1e3:.1f} ms "
      f"({na * n * 8 / te / 1e9:.2f} / {na * n * 8 / td / 1e9:.2f} GB/s of values)")
