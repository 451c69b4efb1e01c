"""Compile-time guard (CPU: hipcc cross-compiles gfx950 without a GPU): the hot loops of the sweep kernels stay free of
scratch traffic.  Round 4 found the fused phase 2 at 128 paths spilling ~22 doubles INSIDE its state loops (43 KB to and
from scratch per 128 KB column: 0.42 of the HBM peak instead of 0.59) and 32 emission selects compiled into
32 loads from a two-entry table in scratch — neither visible in any result, both visible in the ISA.  The test compiles the
kernel file to assembly and looks at every LARGE basic block (the unrolled state loops) of the kernels named below."""
import re
import subprocess
from pathlib import Path

import pytest

from pangenie_amd import build

SRC = Path(build.__file__).resolve().parent / "csrc" / "pg_kernels.hip"

# mangled kernel name -> (blocks of at least this many instructions must have no scratch operation, whole-kernel scratch allowed)
KERNELS = {
    "_Z7k_sweepILi128ELi32ELi1ELb0ELi2EEvPK9DevContigj": (200, True),    # general kernel, 128 paths, fused phase 2 (rare paths may spill)
    "_Z7k_sweepILi64ELi16ELi1ELb1ELi2EEvPK9DevContigj": (200, False),    # general kernel, 64 paths, phase 2
    "_Z13k_sweep_lean2ILi16EEvPK9DevContig": (200, True),                 # triangle chains, phase 2
    "_Z14k_sweep_leanx2PK9DevContig": (300, True),                        # ... with multiallelic objects (round 6): the state loops (both allele-count variants) carry no scratch
    "_Z12k_sweep_leanILi1ELi16ELb0EEvPK9DevContigj": (100, False),        # the lone-chain lean step
    "_Z17k_sweep_leanx_triPK9DevContig": (100, True),                     # phase 1 of 64-path triangle chains with multiallelic objects (the lean-x step)
    "_Z18k_sweep_leanx_triwPK9DevContig": (100, True),                    # ... with wide columns (round 6: wide_fix — the rare branch may spill, the state loop must not)
    "_Z15k_sweep_small16ILi1EEvPK9DevContigPKjjjPd": (100, False),        # four half-chains per wave, phase 1
    "_Z15k_sweep_small16ILi2EEvPK9DevContigPKjjjPd": (100, False),        # ... phase 2
    "_Z16k_sweep_small16xILi1EEvPK9DevContigPKjjjPd": (100, False),       # the same step with table emissions (16 paths, multiallelic objects), phase 1
    "_Z16k_sweep_small16xILi2EEvPK9DevContigPKjjjPd": (100, False),       # ... phase 2 (256 + AGPRs: register moves, no scratch)
}


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    try:
        hipcc = build.hipcc_path()
    except RuntimeError:
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "pg_kernels.s"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
           "-Wno-unused-value", "-Wno-unused-result", "-mllvm", "-amdgpu-mfma-vgpr-form", str(SRC), "-o", str(out)]
    r = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text(), r.stderr


def blocks_of(text, name):
    i = text.index(name + ":")
    body = text[i:text.index(".Lfunc_end", i)].split("\n")
    blocks, cur = [], None
    for line in body:
        if re.match(r"^\.LBB\d+_\d+:", line):
            cur = [line.split(":")[0], 0, 0]
            blocks.append(cur)
            continue
        s = line.strip()
        if cur is None or not s or s.startswith(";") or s.startswith("."):
            continue
        cur[1] += 1
        if s.startswith("scratch_"):
            cur[2] += 1
    return blocks


@pytest.mark.parametrize("kernel", sorted(KERNELS))
def test_hot_loops_have_no_scratch_traffic(asm, kernel):
    min_instr, spills_elsewhere_ok = KERNELS[kernel]
    blocks = blocks_of(asm[0], kernel)
    assert blocks, kernel
    hot = [b for b in blocks if b[1] >= min_instr]
    assert hot, "no large basic block found: the kernel's shape changed, adjust the threshold"
    dirty = [(b[0], b[1], b[2]) for b in hot if b[2]]
    # (the 128-path kernel's resume prologue is a large block too: it runs once per launch and may spill a few values; the
    # spilling version of round 3 had THREE large blocks — the three variants of the state loop — with 21 scratch stores each)
    if spills_elsewhere_ok:
        assert len(dirty) <= 1 and all(d[2] <= 20 for d in dirty), dirty
        assert sum(1 for b in hot if not b[2]) >= 3, "the unrolled state loops must be scratch-free"
    else:
        assert not dirty, dirty


# waves per SIMD the measured configurations rely on: a register more and the occupancy — and with it the
# number of half-chains a CU holds — drops a step
OCCUPANCY = {
    "_Z7k_sweepILi32ELi8ELi1ELb1ELi1EEvPK9DevContigj": 4,     # 17 ... 32 paths, store-only phases: two waves per half-chain, eight half-chains per CU
    "_Z7k_sweepILi32ELi8ELi1ELb1ELi2EEvPK9DevContigj": 3,     # ... fused phase 2
    "_Z7k_sweepILi64ELi16ELi1ELb1ELi2EEvPK9DevContigj": 2,
    "_Z7k_sweepILi128ELi32ELi1ELb0ELi2EEvPK9DevContigj": 2,   # eight waves per workgroup: two per SIMD
    "_Z16k_sweep_lean_triILi1ELi16EEvPK9DevContigj": 2,       # two workgroups per CU
    "_Z13k_sweep_lean2ILi16EEvPK9DevContig": 2,
    "_Z14k_sweep_leanx2PK9DevContig": 2,                      # two workgroups per CU
    "_Z15k_sweep_small16ILi1EEvPK9DevContigPKjjjPd": 2,
    "_Z15k_sweep_small16ILi2EEvPK9DevContigPKjjjPd": 1,       # three partner-column buffers: one wave per SIMD, 2048 waves per launch
    "_Z12k_bins_lean2PK9DevContig": 8,
    "_Z16k_sweep_small16xILi1EEvPK9DevContigPKjjjPd": 2,      # 18 KB of LDS per wave: eight workgroups per CU
    "_Z16k_sweep_small16xILi3EEvPK9DevContigPKjjjPd": 2,
    "_Z8k_bins_xPK9DevContig": 5,                             # the bins of a column in registers, no LDS row per thread
}


def test_occupancy_of_the_measured_configurations(asm):
    remarks = asm[1]
    got = {}
    for blk in remarks.split("Function Name: ")[1:]:
        name = blk.split()[0]
        m = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", blk)
        if m:
            got[name] = int(m.group(1))
    for name, want in OCCUPANCY.items():
        assert name in got, name
        assert got[name] >= want, (name, got[name], want)
