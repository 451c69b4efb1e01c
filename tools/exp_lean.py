"""Timing experiments on the lean sweep kernel (k_sweep_lean): builds variants of the product library
with one ingredient of the step compiled out (-DPG_LEANX=<mask>, see pg_kernels.hip) and prints the
phase-1 sweep time per column.  Results of the variants are wrong by construction; tooling only.
usage: python tools/exp_lean.py build [masks...]  (here, CPU)   |   python tools/exp_lean.py run [masks...]  (GPU box)"""
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
NAMES = {0: "baseline", 1: "no column stores", 2: "no MFMA total", 4: "no u_i round trip", 8: "no column-sum reads", 16: "no per-state arithmetic",
         32: "no barrier", 64: "no record reads", 128: "no scale stores", 129: "no stores at all", 255: "none of them",
         17: "no arithmetic, no stores", 256: "one MFMA total instead of two", 258: "no MFMA at all (2|256)", 6: "no MFMA, no u", 14: "no sums at all", 46: "no sums, no barrier"}
# a mask may carry a codegen-variant suffix: "0v3" = PG_LEANX=0, PG_LEANV=3; "0f" adds -mllvm -amdgpu-mfma-vgpr-form
args_ = sys.argv[2:] or [str(k) for k in NAMES]
def parse(a):
    f = a.endswith("f"); a = a.rstrip("f")
    m, _, v = a.partition("v")
    return int(m), int(v or 0), f
masks = [parse(a) for a in args_]
lib = lambda m: os.path.join(os.getcwd(), "tools", "_build", "libpangenie_hmm_leanx%dv%d%s.so" % (m[0], m[1], "f" if m[2] else ""))
if sys.argv[1] == "build":
    from pangenie_amd import build
    for m in masks:
        build.build_hip(out=lib(m), defines=("PG_LEANX=%d" % m[0], "PG_LEANV=%d" % m[1], "PG_CHAIN_PROF=1"), force=True,
                        extra=(("-mllvm", "-amdgpu-mfma-vgpr-form") if m[2] else ()))
        print("built", lib(m))
else:
    for m in masks:
        env = dict(os.environ, PANGENIE_HMM_LIB=lib(m))
        code = ("import sys; sys.path.insert(0,'.'); from pangenie_amd import hmm; from pangenie_amd.panel import synthetic_panel, default_table_args;"
                "b=synthetic_panel(50000,64,20,seed=12345); job=hmm.Job([b],hmm.ProbabilityTable(*default_table_args()),hmm.make_params(1.26,False,1e-5));"
                "job.run(); job.run(); ms=job.kernel_ms(); C=job.fetch(0).n_columns; q=job.profile_counters(0).astype(float);"
                "print('%%-28s phase1 %%7.2f ms = %%5.0f ns/column | cycles/column forward %%5.0f backward %%5.0f | phase2 %%7.2f ms' %% (%r, ms['k_sweep_phase1'], ms['k_sweep_phase1']*1e6/(C/2), q[0]/(C/2), q[16]/(C/2), ms['k_sweep_phase2']))" % (NAMES.get(m[0], str(m[0])) + (" v%d" % m[1] if m[1] else "") + (" vgpr-mfma" if m[2] else "")))
        subprocess.run([sys.executable, "-c", code], env=env)
