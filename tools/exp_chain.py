"""Timing experiments on the chain kernels: builds variants of the product library with one
ingredient of the recursion step compiled out (-DPG_EXP=<mask>, see pg_kernels.hip) and prints the
sweep kernel times.  Results of the variants are wrong by construction; tooling only.
usage: python tools/exp_chain.py build   (here, CPU)   |   python tools/exp_chain.py run [H]  (GPU box)
A single chain runs the chunked mode by default (the phase-2 columns show the last chunk's store-only
sweep); set PG_SWEEP_MODE=fused in the environment to time the fused phase 2."""
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
MASKS = {0: "baseline", 1: "no column stores", 2: "no wave_sum", 4: "no u_i round trip", 8: "no posterior",
         16: "no recursion arithmetic", 32: "no ring read", 64: "no sum read", 128: "no column DMA", 255: "none of them"}
lib = lambda m: os.path.join(os.getcwd(), "tools", "_build", "libpangenie_hmm_exp%d.so" % m)
if sys.argv[1] == "build":
    from pangenie_amd import build
    for m in MASKS:
        build.build_hip(out=lib(m), defines=("PG_EXP=%d" % m, "PG_CHAIN_PROF=1"))
        print("built", lib(m))
elif sys.argv[1] == "run":
    H = sys.argv[2] if len(sys.argv) > 2 else "64"
    for m, name in MASKS.items():
        env = dict(os.environ, PANGENIE_HMM_LIB=lib(m), PG_DEBUG="8")
        code = ("import sys; sys.path.insert(0,'.'); from pangenie_amd import hmm; from pangenie_amd.panel import synthetic_panel, default_table_args;"
                "b=synthetic_panel(50000,%s,20,seed=12345); job=hmm.Job([b],hmm.ProbabilityTable(*default_table_args()),hmm.make_params(1.26,False,1e-5));"
                "job.run(); job.run(); ms=job.kernel_ms(); C=job.fetch(0).n_columns; q=job.profile_counters(0).astype(float);"
                "print('%%-26s phase1 %%7.2f ms  phase2 %%7.2f ms | cycles/col fwd1 %%5.0f bwd1 %%5.0f fwd2 %%5.0f bwd2 %%5.0f' %% (%r, ms['k_sweep_phase1'], ms['k_sweep_phase2'], q[0]/max(q[2],1), q[16]/max(q[18],1), q[8]/max(q[10],1), q[24]/max(q[26],1)))" % (H, name))
        subprocess.run([sys.executable, "-c", code], env=env)
