"""Per-segment cycle timeline of one column step of the plain lean kernel (k_sweep_lean, wave 0): runs the -DPG_LEAN_TIMELINE
build (tools/_build/libpangenie_hmm_timeline.so; build: python tools/exp_timeline.py build) on a lone 50 000-variant,
64-path chain and prints the mean cycles per segment and step for both roles.  Tooling only."""
import os
import sys

sys.path.insert(0, os.getcwd())
LIB = os.path.join(os.getcwd(), "tools", "_build", "libpangenie_hmm_timeline.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from pangenie_amd import build
    build.build_hip(out=LIB, defines=("PG_LEAN_TIMELINE=1", "PG_CHAIN_PROF=1"), force=True)
    print("built", LIB)
    sys.exit(0)
os.environ["PANGENIE_HMM_LIB"] = LIB
PIPE = len(sys.argv) > 1 and sys.argv[1] == "pipe"
if PIPE:
    os.environ["PG_KERNELS"] = "leanpipe"
from pangenie_amd import hmm  # noqa: E402
from pangenie_amd.panel import default_table_args, synthetic_panel  # noqa: E402

b = synthetic_panel(50000, 64, 20, seed=12345)
job = hmm.Job([b], hmm.ProbabilityTable(*default_table_args()), hmm.make_params(1.26, False, 1e-5))
job.run(); job.run()
ms = job.kernel_ms(); C = job.fetch(0).n_columns
p = job.profile_counters(0).astype(float)
print(("pipelined" if PIPE else "plain") + " lean step, timeline build (stamps perturb the step: compare the sum with the unstamped cycles per column)")
print("phase 1 %.2f ms = %.0f ns per column (%d columns)" % (ms["k_sweep_phase1"], ms["k_sweep_phase1"] * 1e6 / (C / 2), C))
FWD = ["barrier release -> column sums back from LDS (4 + 4 reads, 3 adds)", "first MFMA + 3 adds", "second MFMA (total S)",
       "zero test, exponent, scaled constants", "u_j, first row pair's states", "other seven row pairs (states, stores, emission reads)",
       "partial sum to LDS, per-column scalar", "barrier"]
BWD = ["exponent, scaled constants, scalar parked", "barrier", "column sums back from LDS + next records", "first MFMA + 3 adds",
       "second MFMA (total)", "u_j, first row pair's states", "other seven row pairs", "partial sum to LDS, scalars, zero test"]
PSEG = ["top of the step: LDS reads issued, descriptor of the column after next", "states 0, 1", "states 2, 3 + Y partials back from LDS, closed form of the column sums",
        "states 4 .. 10 + six DPP levels of the two class totals", "state 11 + readlanes, zero test, next step's constants", "states 12 .. 15 + next descriptor's reads",
        "Y partial to LDS, last store, per-column scalars", "barrier"]
if PIPE:
    FWD = BWD = PSEG
for name, o, segs in (("forward role (last chunk of phase 2)", 32, FWD), ("backward role", 48, BWD)):
    n = max(p[o + 15], 1.0)
    print("%s: %d steps" % (name, n))
    tot = 0.0
    for i, sname in enumerate(segs):
        print("   %7.1f cycles  %s" % (p[o + i] / n, sname))
        tot += p[o + i] / n
    print("   %7.1f cycles  sum of the stamped segments" % tot)
