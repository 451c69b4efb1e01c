"""Builds the in-tree native code: the HIP product library (hipcc, gfx950) and —
separately, as test infrastructure — the CPU oracle (gcc)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "pangenie_amd" / "csrc"
HIP_LIB = CSRC / "libpangenie_hmm.so"
HIP_SOURCES = [CSRC / "pg_kernels.hip", CSRC / "pg_shim.cpp", CSRC / "pg_gather.cpp", CSRC / "pg_sampler.hip", CSRC / "pg_viterbi.hip"]
HIP_DEPS = HIP_SOURCES + [CSRC / "pg_device.h", CSRC / "pg_devmath.h", CSRC / "pg_small16x.h", CSRC / "pg_experiments.h", ROOT / "include" / "pangenie_hmm.h", ROOT / "include" / "pangenie_sampler.h"]


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def build_hip(force: bool = False, verbose: bool = False, out: Path | None = None, defines=(), extra=()) -> Path:
    """hipcc --offload-arch=gfx950 -> pangenie_amd/csrc/libpangenie_hmm.so (in-tree).
    `out`/`defines` build a variant elsewhere (tools/exp_pipe.py build ...: -DPG_CHAIN_PROF and the masks of pg_experiments.h)."""
    target = Path(out) if out else HIP_LIB
    if force or _stale(target, HIP_DEPS):
        target.parent.mkdir(parents=True, exist_ok=True)
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-Wno-unused-value", "-Wno-unused-result",
               # fp64 MFMA results straight into VGPRs (the wave totals of the lean sweep): no v_accvgpr_read round trip
               "-mllvm", "-amdgpu-mfma-vgpr-form",
               *[f"-D{d}" for d in defines], *extra,
               *map(str, HIP_SOURCES), "-ldl", "-o", str(target)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or r.returncode:
            print(" ".join(cmd))
            print(r.stdout, r.stderr)
        if r.returncode:
            raise RuntimeError("hipcc failed:\n" + r.stderr)
    return target


def build_oracle(force: bool = False) -> Path:
    """TEST INFRASTRUCTURE: oracle/_build/libpg_oracle.so (gcc, long double)."""
    if force:
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "clean"], check=True, capture_output=True)
    subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    if Path("/root/reference/src/samplingtransitions.cpp").exists():
        # oracle/_ref: the one reference translation unit of the path that compiles here (see oracle/Makefile)
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "ref"], check=True, capture_output=True)
    return ROOT / "oracle" / "_build" / "libpg_oracle.so"


HOST_DIR = ROOT / "pangenie_amd" / "host"
HOST_LIB = HOST_DIR / "libpangenie_host.so"
HOST_TEST = ROOT / "tests" / "cpp" / "test_host.bin"
HOST_MOCK_TEST = ROOT / "tests" / "cpp" / "test_multi_gpu_mock.bin"


def build_host(force: bool = False) -> Path:
    """g++ -> pangenie_amd/host/libpangenie_host.so (C++ mirror of the reference interface over
    the C ABI) and tests/cpp/test_host.bin."""
    build_hip()
    cxx = shutil.which("g++") or "g++"
    deps = [HOST_DIR / "pangenie_host.cpp", HOST_DIR / "pangenie_host.hpp", HOST_DIR / "cereal_io.cpp", HOST_DIR / "cereal_io.hpp",
            HOST_DIR / "kmer_counts.cpp", HOST_DIR / "kmer_counts.hpp", HOST_DIR / "graph_io.cpp", HOST_DIR / "graph_io.hpp",
            HOST_DIR / "index_builder.cpp", HOST_DIR / "index_builder.hpp", HOST_DIR / "archive_bytes.hpp", ROOT / "include" / "pangenie_hmm.h"]
    if force or _stale(HOST_LIB, deps):
        cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", str(HOST_DIR / "pangenie_host.cpp"), str(HOST_DIR / "cereal_io.cpp"),
               str(HOST_DIR / "kmer_counts.cpp"), str(HOST_DIR / "graph_io.cpp"), str(HOST_DIR / "index_builder.cpp"), "-o", str(HOST_LIB), f"-L{CSRC}", "-lpangenie_hmm", "-lpthread", "-lz", "-Wl,-rpath,$ORIGIN/../csrc"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("g++ (host lib) failed:\n" + r.stderr)
    tsrc = ROOT / "tests" / "cpp" / "test_host.cpp"
    if force or _stale(HOST_TEST, [tsrc, HOST_LIB]):
        cmd = [cxx, "-O1", "-std=c++17", "-Wall", str(tsrc), "-o", str(HOST_TEST), f"-L{HOST_DIR}", "-lpangenie_host",
               f"-L{CSRC}", "-lpangenie_hmm", "-lz", "-lpthread", "-Wl,-rpath,$ORIGIN/../../pangenie_amd/host:$ORIGIN/../../pangenie_amd/csrc"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("g++ (host tests) failed:\n" + r.stderr)
    msrc = ROOT / "tests" / "cpp" / "test_multi_gpu_mock.cpp"
    if msrc.exists() and (force or _stale(HOST_MOCK_TEST, [msrc, HOST_LIB])):
        # the multi-GPU job loop with the device calls replaced at the C-ABI seam (definitions in the executable, -rdynamic)
        cmd = [cxx, "-O1", "-std=c++17", "-Wall", "-rdynamic", str(msrc), "-o", str(HOST_MOCK_TEST), f"-L{HOST_DIR}", "-lpangenie_host",
               f"-L{CSRC}", "-lpangenie_hmm", "-lz", "-lpthread", "-Wl,-rpath,$ORIGIN/../../pangenie_amd/host:$ORIGIN/../../pangenie_amd/csrc"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("g++ (multi-GPU mock test) failed:\n" + r.stderr)
    return HOST_LIB
