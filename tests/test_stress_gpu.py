"""Randomised batches at sizes beyond the CPU oracle's reach: every array's midpoint against a
device sort, its width against the zig-zag maximum, then the batch's round trip (the generator
and the checks live in scripts/stress_encode.py, which runs longer campaigns)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


@pytest.mark.parametrize("seed,max_log2,max_arrays", [(1, 23, 300), (2, 23, 300), (1003, 23, 300), (5004, 26, 12), (20005, 14, 3000)])
def test_random_batches(seed, max_log2, max_arrays):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    import stress_encode
    from zbra_amd import _lib, device
    assert stress_encode.run_trial(torch, device, _lib, seed, max_log2, max_arrays) > 0


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_random_decode_batches(seed):
    """Random payload bytes under crafted headers (any width byte, short tails, trailing bytes):
    statuses and values against the oracle, guard words around every output."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    import stress_decode
    from oracle import zbra_oracle_c as oc
    from zbra_amd import _lib, device
    assert stress_decode.run_trial(torch, device, _lib, oc, seed) > 0


@pytest.mark.parametrize("env", [{"ZBRA_HIP_ENCODE_BRACKET": "0"},            # every array: selection over all its values
                                 {"ZBRA_HIP_SMALL_SELECT_KEYS": "1000000"},   # long candidate lists finished by one workgroup
                                 {"ZBRA_HIP_SMALL_SELECT_KEYS": "1"}])        # every list left to the pass-per-launch kernels
def test_random_batches_on_the_other_selection_routes(env):
    """The encoder's routes that the default thresholds rarely take, each in a process of its own
    (the library reads these variables once): same generator, same checks against a device sort."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "stress_encode.py"), "--trials", "4", "--seed", "4242",
                        "--max-log2", "22", "--max-arrays", "120"],
                       env={**os.environ, **env}, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "all midpoints, widths and round trips exact" in p.stdout
