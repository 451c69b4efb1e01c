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


def test_multi_gpu_job_loop_with_the_device_mocked_at_the_c_abi(binary):
    """pangenie::run_contigs_multi_gpu on 1 / 2 / 3 / 8 / 12 'devices' without a GPU: pg_job_* / pg_comm_* /
    pg_hmm_gather_to_host are replaced at the C-ABI seam (tests/cpp/test_multi_gpu_mock.cpp), so the plan (longest
    processing time first), one job per device, the ONE exchange with its block sizes, and the reassembly of every
    task's GenotypingResults run for real; what is left untested without hardware is RCCL itself."""
    r = subprocess.run([str(build.HOST_MOCK_TEST)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and " 0 failed" in r.stdout, r.stdout + r.stderr


def test_host_classes_cpu_under_sanitizers(binary, tmp_path):
    """The same CPU scenarios with the host sources compiled under AddressSanitizer + UndefinedBehaviorSanitizer (archive
    readers on malformed input, the k-mer counters' threads and tables): no report, same checks."""
    import os
    import shutil
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    host = build.ROOT / "pangenie_amd" / "host"
    exe = tmp_path / "test_host_san"
    cmd = [cxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined",
           str(build.ROOT / "tests" / "cpp" / "test_host.cpp"), str(host / "pangenie_host.cpp"), str(host / "cereal_io.cpp"),
           str(host / "kmer_counts.cpp"), str(host / "graph_io.cpp"), str(host / "index_builder.cpp"), "-o", str(exe),
           f"-L{build.ROOT / 'pangenie_amd' / 'csrc'}", "-lpangenie_hmm", "-lz", "-lpthread", f"-Wl,-rpath,{build.ROOT / 'pangenie_amd' / 'csrc'}"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0")   # (the HIP runtime the library links keeps its own allocations)
    r = subprocess.run([str(exe), "cpu", str(build.ROOT / "tests" / "golden")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_hmm_through_cpp_adapter_gpu(binary):
    r = subprocess.run([binary, "gpu", str(build.ROOT / "tests" / "golden")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
