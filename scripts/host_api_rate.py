#!/usr/bin/env python3
"""Rate of the host-pointer (drop-in) entry points: numpy arrays in pageable host memory in and out,
as zbra-core's write/read_int_array_compressed would call them."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from zbra_amd import compression as zc

for lg in (16, 20, 24, 26):
    n = 1 << lg
    rng = np.random.default_rng(lg)
    v = 1_700_000_000_000 + rng.integers(0, 4000, n, dtype=np.int64)
    e = zc.compress_int_array(v)
    assert np.array_equal(zc.decompress_int_array(e, n), v)
    reps = max(3, (1 << 27) // n // 4)
    t0 = time.perf_counter()
    for _ in range(reps):
        e = zc.compress_int_array(v)
    te = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        out = zc.decompress_int_array(e, n)
    td = (time.perf_counter() - t0) / reps
    print(f"n=2^{lg}: compress_int_array {te * 1e3:9.3f} ms ({8 * n / te / 1e9:6.2f} GB/s of values)   "
          f"decompress_int_array {td * 1e3:9.3f} ms ({8 * n / td / 1e9:6.2f} GB/s of values)")
