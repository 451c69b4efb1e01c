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
HIP_SOURCES = [CSRC / "pg_kernels.hip", CSRC / "pg_shim.cpp"]
HIP_DEPS = HIP_SOURCES + [CSRC / "pg_device.h", ROOT / "include" / "pangenie_hmm.h"]


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


def build_hip(force: bool = False, verbose: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 -> pangenie_amd/csrc/libpangenie_hmm.so (in-tree)."""
    if force or _stale(HIP_LIB, HIP_DEPS):
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-Wno-unused-value", "-Wno-unused-result",
               *map(str, HIP_SOURCES), "-o", str(HIP_LIB)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or r.returncode:
            print(" ".join(cmd))
            print(r.stdout, r.stderr)
        if r.returncode:
            raise RuntimeError("hipcc failed:\n" + r.stderr)
    return HIP_LIB


def build_oracle(force: bool = False) -> Path:
    """TEST INFRASTRUCTURE: oracle/_build/libpg_oracle.so (gcc, long double)."""
    if force:
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "clean"], check=True, capture_output=True)
    subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    return ROOT / "oracle" / "_build" / "libpg_oracle.so"
