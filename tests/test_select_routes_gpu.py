"""The encoder's exact median on column shapes that take each route of its selection, on arrays long
enough for the many-workgroup levels (more than 2^16 values): bytes against the oracle's, whose
median comes from a full sort (compression.rs:58-68).  Shapes: a wide bracket whose keys are gathered
after one counted digit; a candidate list that overflows (two far-apart values, half each); a run of
one value at the median of otherwise wide data; a dense cluster of different values at the median
of wide data (which no level resolves: the finishing workgroup reads on, digit by digit); sorted
wide data; u32 values over their whole range; brackets of a few keys, dense (most of the values
inside: per-lane counting columns) and sparse (shared histogram); a median value that occurs several
times in wide data, and middle ranks that lie far apart (what the finishing workgroup makes of the
keys it holds in LDS)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

N0 = (1 << 18) + 4096


def _shape(name, n, seed):
    rng = np.random.default_rng(seed)
    if name == "wide_uniform":
        return rng.integers(-(1 << 63), (1 << 63) - 1, n, dtype=np.int64, endpoint=True)
    if name == "wide_40":
        return rng.integers(-(1 << 39), 1 << 39, n, dtype=np.int64) + 1_000_000_007
    if name == "two_far":
        v = np.where(np.arange(n) % 2 == 0, -7_000_000_000_000_000, 9_000_000_000_000_000).astype(np.int64)
        return v
    if name == "three_far":
        return rng.choice(np.array([-(1 << 62), 12345, (1 << 62) + 99], dtype=np.int64), n, p=[0.3, 0.4, 0.3])
    if name == "popular_in_wide":
        v = np.empty(n, dtype=np.int64)
        m = int(n * 0.03)
        lo = (n - m) // 2
        v[:lo] = -rng.integers(1 << 40, 1 << 62, lo, dtype=np.int64)
        v[lo:lo + m] = 12345
        v[lo + m:] = rng.integers(1 << 40, 1 << 62, n - lo - m, dtype=np.int64)
        rng.shuffle(v)
        return v
    if name == "cluster_in_wide":
        v = np.empty(n, dtype=np.int64)
        m = int(n * 0.04)
        lo = (n - m) // 2
        v[:lo] = -rng.integers(1 << 40, 1 << 62, lo, dtype=np.int64)
        v[lo:lo + m] = 500_000_000 + rng.integers(0, 5000, m, dtype=np.int64)
        v[lo + m:] = rng.integers(1 << 40, 1 << 62, n - lo - m, dtype=np.int64)
        rng.shuffle(v)
        return v
    if name == "sorted_wide":
        return (np.arange(n, dtype=np.int64) - n // 3) * 3_000_000_000_017
    if name == "extremes_mix":      # span 2^64 - 1: the width needs the OR pass unless an extreme takes 64 bits
        return rng.choice(np.array([-(1 << 63), (1 << 63) - 1, 0, -1, 1], dtype=np.int64), n)
    if name == "min_max_only":      # v - mid wraps for every value: width 1 (compression.rs:60-68 with b - a = -1)
        return np.where(rng.integers(0, 2, n) == 0, -(1 << 63), (1 << 63) - 1).astype(np.int64)
    if name == "few_keys_dense":    # a bracket of three or four keys that holds most of the values: counted in per-lane columns
        return 1000 + rng.choice(np.array([0, 1, 2, 3], dtype=np.int64), n, p=[0.45, 0.03, 0.03, 0.49])
    if name == "few_keys_sparse":   # 48 equally likely values: a bracket of a few keys with a few per cent of the values, shared histogram
        return -5000 + rng.integers(0, 48, n, dtype=np.int64)
    if name == "dup_at_median":     # wide data whose median value occurs seven times: for an even count b is a again
        v = np.sort(rng.integers(-(1 << 62), 1 << 62, n, dtype=np.int64))
        k = (n - 1) // 2
        v[k - 3:k + 4] = v[k]
        rng.shuffle(v)
        return v
    if name == "split_halves":      # the two middle ranks lie far apart: b is the smallest key above everything that shares a's digits
        h = (n + 1) // 2
        v = np.concatenate([-rng.integers(1 << 40, 1 << 62, h, dtype=np.int64), rng.integers(1 << 40, 1 << 62, n - h, dtype=np.int64)])
        rng.shuffle(v)
        return v
    raise KeyError(name)


SHAPES = ["wide_uniform", "wide_40", "two_far", "three_far", "popular_in_wide", "cluster_in_wide", "sorted_wide", "extremes_mix",
          "min_max_only", "few_keys_dense", "few_keys_sparse", "dup_at_median", "split_halves"]


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    from oracle import zbra_oracle_c as oc
    from zbra_amd import compression as zc
    return oc, zc


@pytest.mark.parametrize("name", SHAPES)
def test_long_arrays_match_oracle(env, name):
    oc, zc = env
    for n in (N0, N0 + 1, (1 << 21) + 7):
        v = _shape(name, n, seed=n % 1000 + len(name))
        want = oc.compress_int_array(v)
        got = zc.compress_int_array(v)
        assert got[:13] == want[:13], (name, n, got[:13].hex(), want[:13].hex())
        assert got == want, (name, n)


def test_u32_over_the_whole_range(env):
    oc, zc = env
    for n in (N0, N0 + 1):
        tags = np.random.default_rng(n).integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        assert zc.compress_u32_array(tags) == oc.compress_int_array(tags.astype(np.int64))


def test_batch_of_long_arrays_of_every_shape(env):
    """All shapes in ONE submission: the levels deal their workgroups over several pending arrays
    that are at different stages (counting, gathering, finished)."""
    oc, zc = env
    arrs = [_shape(name, N0 + 3 * i, seed=70 + i) for i, name in enumerate(SHAPES)] + [np.arange(10, dtype=np.int64), np.zeros(0, dtype=np.int64)]
    frames, offs = zc.write_int_arrays_compressed(arrs)
    for i, v in enumerate(arrs):
        e = oc.compress_int_array(v)
        want = len(v).to_bytes(4, "little") + len(e).to_bytes(4, "little") + e
        assert frames[offs[i]:offs[i + 1]] == want, i


@pytest.mark.parametrize("extra", [{"ZBRA_HIP_ENCODE_BRACKET": "0"}, {"ZBRA_HIP_SMALL_SELECT_KEYS": "1"},
                                   {"ZBRA_HIP_SMALL_SELECT_KEYS": "1000000"}])
def test_shapes_on_the_other_routes(extra):
    """The same shapes with the sampled bracket off (selection over all values: three counted digits,
    then a gather), with every list left to the levels, and with every list left to one workgroup.
    A process of its own each: the library reads these variables once."""
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "import test_select_routes_gpu as t\n"
        "from oracle import zbra_oracle_c as oc\n"
        "from zbra_amd import compression as zc\n"
        "for name in t.SHAPES:\n"
        "    for n in (t.N0, t.N0 + 1):\n"
        "        v = t._shape(name, n, 5)\n"
        "        assert zc.compress_int_array(v) == oc.compress_int_array(v), (name, n)\n"
        "import test_encode_u32_gpu as tu\n"   # u32 arrays have a bracket pass and a pack kernel of their own
        "for n in (70001, (1 << 20) + 3):\n"
        "    for k, v in tu._shapes(np.random.default_rng(n), n).items():\n"
        "        assert zc.compress_u32_array(v) == oc.compress_int_array(v.astype(np.int64)), (k, n)\n"
        "print('routes ok')\n")
    p = subprocess.run([sys.executable, "-c", code], env={**os.environ, **extra}, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0 and "routes ok" in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]
