#!/usr/bin/env python3
"""Host-pointer calls from several threads at once on one device: what the host lanes buy.
Three workloads, each timed with 1, 2 and 4 caller threads:
  small   - compress + decompress of 2^16-value arrays (latency-bound calls)
  mid     - compress + decompress of 2^23-value arrays (64 MiB of values: not piecewise)
  mixed   - half the threads only compress, the other half only decompress 2^24-value arrays
            (the two directions of the link)
Run with ZBRA_HIP_HOST_LANES=1 for the serialised behaviour.  Usage: host_threads_rate.py [LABEL]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from zbra_amd import compression as zc

label = sys.argv[1] if len(sys.argv) > 1 else "lanes=" + os.environ.get("ZBRA_HIP_HOST_LANES", "default")


def col(n, seed):
    return 1_700_000_000_000 + np.random.default_rng(seed).integers(0, 4000, n, dtype=np.int64)


def run(threads, fn, reps):
    fn(0)  # warm: buffers of lane 0
    barrier = threading.Barrier(threads + 1)

    def work(t):
        fn(t)  # warm this thread's lane
        barrier.wait()
        for _ in range(reps):
            fn(t)
        barrier.wait()

    ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for t in ts:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    barrier.wait()
    dt = time.perf_counter() - t0
    for t in ts:
        t.join()
    return dt


for name, n, reps in (("small", 1 << 16, 200), ("mid", 1 << 23, 10)):
    cols = [col(n, 10 + t) for t in range(4)]
    encs = [zc.compress_int_array(c) for c in cols]

    def both(t, cols=cols, n=n):
        e = zc.compress_int_array(cols[t])
        zc.decompress_int_array(e, n)

    for threads in (1, 2, 4):
        dt = run(threads, both, reps)
        calls = threads * reps
        print(f"{label:>14s} {name:>6s} {threads} threads: {dt / calls * 1e3:8.3f} ms per compress+decompress, {calls * n * 16 / dt / 1e9:6.2f} GB/s of values both ways", flush=True)

n = 1 << 24
cols = [col(n, 20 + t) for t in range(4)]
encs = [zc.compress_int_array(c) for c in cols]


def mixed(t):
    if t % 2 == 0:
        zc.compress_int_array(cols[t])
    else:
        zc.decompress_int_array(encs[t], n)


for threads in (2, 4):
    dt = run(threads, mixed, 8)
    print(f"{label:>14s}  mixed {threads} threads: {threads * 8 * n * 8 / dt / 1e9:6.2f} GB/s of values (half compressed, half decompressed)", flush=True)
