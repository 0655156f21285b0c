"""Runs tests/c/test_wrapper_pattern.c: the C ABI driven from plain C in the allocate / call /
truncate pattern of the Rust wrappers (rust/zbra-hip-sys), every result against the CPU oracle."""
import os
import subprocess

import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "tests", "c", "test_wrapper_pattern")


def _ensure_built():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c")])


@pytest.mark.gpu
def test_wrapper_pattern_on_gpu():
    _ensure_built()
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert " 0 failures" in p.stdout


def test_wrapper_pattern_refuses_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    _ensure_built()
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=60)
    assert p.returncode == 2 and "need a GPU" in p.stdout


MULTI = os.path.join(ROOT, "tests", "c", "test_multi_device")


@pytest.mark.gpu
def test_every_device_from_one_process():
    """tests/c/test_multi_device.c: two threads per device of zbra_hip_device_count (device-resident
    calls incl. the no-allocation decode, and host-pointer calls), all at once, against the oracle."""
    if not os.path.exists(MULTI):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c")])
    p = subprocess.run([MULTI], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert " 0 failures" in p.stdout and "device 0 at " in p.stdout


HOST_PATHS = os.path.join(ROOT, "tests", "c", "test_host_paths")


@pytest.mark.gpu
def test_threaded_host_paths_from_c():
    """tests/c/test_host_paths.c: the two-lane write path over several rounds at three capacities and a
    failing one, and zbra_hip_file_decode_multi with two jobs per device, against the oracle."""
    if not os.path.exists(HOST_PATHS):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c")])
    p = subprocess.run([HOST_PATHS], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert " 0 failures" in p.stdout
