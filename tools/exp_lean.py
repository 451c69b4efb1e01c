"""Timing experiments on the lean sweep kernel (k_sweep_lean): builds variants of the product library with compile-time
knobs of the lean step (-DNAME=value, e.g. PG_LEAN_EXP=1: no column stores) and prints the phase-1 / phase-2 sweep time per column of a
50 000-variant, 64-path chain; every variant is also checked against the default library's results (bit for bit unless
the knob changes the arithmetic).  Tooling only.
usage: python tools/exp_lean.py build NAME=V[,NAME=V...] ...   (here, CPU)
       python tools/exp_lean.py run   NAME=V[,NAME=V...] ...   (GPU box; "default" = the product library)"""
import os
import subprocess
import sys

sys.path.insert(0, os.getcwd())
variants = sys.argv[2:] or ["default"]


def lib(v):
    if v == "default":
        return os.path.join(os.getcwd(), "pangenie_amd", "csrc", "libpangenie_hmm.so")
    return os.path.join(os.getcwd(), "tools", "_build", "libpangenie_hmm_%s.so" % v.replace("=", "").replace(",", "_"))


if sys.argv[1] == "build":
    from pangenie_amd import build
    for v in variants:
        if v == "default":
            continue
        build.build_hip(out=lib(v), defines=tuple(v.split(",")) + ("PG_CHAIN_PROF=1",), force=True)
        print("built", lib(v))
else:
    code = ("import sys, numpy as np; sys.path.insert(0,'.'); from pangenie_amd import hmm; from pangenie_amd.panel import synthetic_panel, default_table_args;"
            "b=synthetic_panel(50000,64,20,seed=12345); job=hmm.Job([b],hmm.ProbabilityTable(*default_table_args()),hmm.make_params(1.26,False,1e-5));"
            "job.run(); job.run(); ms=job.kernel_ms(); r=job.fetch(0); C=r.n_columns; q=job.profile_counters(0).astype(float);"
            "import hashlib; h=hashlib.sha1(r.lik.tobytes()+r.lik_exp.tobytes()).hexdigest()[:12];"
            "print('%%-28s phase1 %%7.2f ms = %%5.0f ns/column | cycles/column forward %%5.0f backward %%5.0f | phase2 %%7.2f ms = %%5.0f ns/column | results %%s' %% (%r, ms['k_sweep_phase1'], ms['k_sweep_phase1']*1e6/(C/2), q[0]/(C/2), q[16]/(C/2), ms['k_sweep_phase2'], ms['k_sweep_phase2']*1e6/(C/2), h))")
    for v in variants:
        env = dict(os.environ, PANGENIE_HMM_LIB=lib(v))
        subprocess.run([sys.executable, "-c", code % v], env=env)
