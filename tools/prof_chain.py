import os, sys
sys.path.insert(0, os.getcwd())
os.environ["PG_DEBUG"] = sys.argv[1] if len(sys.argv) > 1 else "8"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
# instrumented variant of the product library (the product build has no profiling code)
from pangenie_amd import build as _build
_prof_lib = os.path.join(os.getcwd(), "tools", "_build", "libpangenie_hmm_prof.so")
if not os.environ.get("PG_PROF_PREBUILT"):
    _build.build_hip(out=_prof_lib, defines=("PG_CHAIN_PROF=1",))
os.environ["PANGENIE_HMM_LIB"] = _prof_lib
from pangenie_amd import hmm
from pangenie_amd.panel import synthetic_panel, default_table_args
V = 50000
b = synthetic_panel(V, H, 20, seed=12345)
job = hmm.Job([b], hmm.ProbabilityTable(*default_table_args()), hmm.make_params(1.26, False, 1e-5))
job.run(); job.run()
ms = job.kernel_ms(); r = job.fetch(0); C = r.n_columns
p = job.profile_counters(0).astype(float)
print("H", H, "dbg", os.environ["PG_DEBUG"], "phase1 ms", ms["k_sweep_phase1"], "phase2 ms", ms["k_sweep_phase2"], "cols", C)
for name, o in (("forward  phase1", 0), ("forward  phase2", 8), ("backward phase1", 16), ("backward phase2", 24)):
    n = max(p[o + 2], 1)
    print(" %s: %6.0f cycles/col (%d steps)" % (name, p[o] / n, n))
