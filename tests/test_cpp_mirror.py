"""Runs the C++ parity tests of the host mirror (include/zbra/*.hpp): the reference's own unit,
property and pipeline tests restated in C++, every result also compared with the CPU oracle."""
import os
import subprocess

import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "tests", "cpp", "test_compression")


def _ensure_built():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])


@pytest.mark.gpu
def test_cpp_mirror_on_gpu():
    _ensure_built()
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "0 failures" in p.stdout


def test_cpp_mirror_refuses_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    _ensure_built()
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=60)
    assert p.returncode == 2 and "need a GPU" in p.stdout
