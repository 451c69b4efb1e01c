"""Runs the C++ tests of the host interface (tests/cpp/test_host.cpp): the reference's own test
scenarios expressed against pangenie_amd/host (same class and method names as the reference)."""
import subprocess

import pytest

from pangenie_amd import build


@pytest.fixture(scope="module")
def binary():
    build.build_host()
    return str(build.HOST_TEST)


def test_host_classes_cpu(binary):
    r = subprocess.run([binary, "cpu", str(build.ROOT / "tests" / "golden")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_hmm_through_cpp_adapter_gpu(binary):
    r = subprocess.run([binary, "gpu", str(build.ROOT / "tests" / "golden")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
