import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


I64_MIN, I64_MAX = -(1 << 63), (1 << 63) - 1

SIZES = [1, 2, 3, 7, 8, 9, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4095, 4097, 10000, 100003]


def column(kind: str, n: int, seed: int) -> np.ndarray:
    """Deterministic int64 test columns covering every width class and the wrap-around corners."""
    rng = np.random.default_rng(seed)
    if kind == "narrow":       # w <= 12, the headline shape
        return 1_700_000_000_000 + rng.integers(0, 4000, n, dtype=np.int64)
    if kind == "constant":     # all-zero deltas -> w = 1
        return np.full(n, -123456789, dtype=np.int64)
    if kind == "binary":       # two values -> w = 1 or 2
        return rng.integers(0, 2, n, dtype=np.int64) * 7 - 3
    if kind == "small_signed":
        return rng.integers(-1000, 1000, n, dtype=np.int64)
    if kind == "w31":          # widest packed width
        return rng.integers(-(1 << 29), 1 << 29, n, dtype=np.int64)
    if kind == "w32":          # first raw width
        return rng.integers(-(1 << 30), 1 << 30, n, dtype=np.int64)
    if kind == "w40":
        return rng.integers(-(1 << 38), 1 << 38, n, dtype=np.int64) + (1 << 50)
    if kind == "full":         # full 64-bit range: wrap-around in every stage
        return rng.integers(I64_MIN, I64_MAX, n, dtype=np.int64, endpoint=True)
    if kind == "extremes":     # only MIN / MAX / 0 / -1
        return rng.choice(np.array([I64_MIN, I64_MAX, 0, -1], dtype=np.int64), n)
    if kind == "sequential":   # benches/compression.rs:8-10
        return np.arange(n, dtype=np.int64)
    if kind == "clustered":    # benches/compression.rs:25-34
        i = np.arange(n, dtype=np.int64)
        return np.array([100, 1000, 10000, 100000], dtype=np.int64)[i % 4] + (i % 20) - 10
    if kind == "time_series":  # benches/compression.rs:36-40
        return 1640995200000 + np.arange(n, dtype=np.int64) * 1000
    if kind == "width_sweep":  # amplitude chosen by seed: every width 1..40 gets hit across seeds
        bits = 1 + seed % 40
        return rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), n, dtype=np.int64) + rng.integers(-(1 << 47), 1 << 47)
    raise ValueError(kind)


KINDS = ["narrow", "constant", "binary", "small_signed", "w31", "w32", "w40", "full", "extremes", "sequential",
         "clustered", "time_series"]


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "known_answers.json")) as f:
        return json.load(f)
