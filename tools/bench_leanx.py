"""One H = 128 chain: k_sweep_leanx against the general kernel, at several multiallelic fractions.
usage (GPU box): python tools/bench_leanx.py [variants]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402

from pangenie_amd import hmm  # noqa: E402
from pangenie_amd.panel import default_table_args, synthetic_panel  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
table = hmm.ProbabilityTable(*default_table_args())
params = hmm.make_params(1.26, False, 1e-5)
for H in (128, 64, 50):
    for multi in ((0.2, 1.0) if H != 128 else (0.0, 0.2)):
        b = synthetic_panel(V, H, 20, seed=77, multiallelic_frac=multi)
        out = {}
        for lx in ("1", "0"):
            os.environ["PG_KERNELS"] = "leanx" if lx == "1" else "noleanx"
            job = hmm.Job([b], table, params)
            job.run()
            job.run()
            ms = job.kernel_ms()
            r = job.fetch(0)
            C = r.n_columns
            out[lx] = r
            print("H %3d multi %.1f leanx=%s %-8s phase1 %8.2f ms = %5.0f ns/column  phase2 %8.2f ms = %5.0f ns/column  %.3f M variants/s" % (
                H, multi, lx, job.sweep_mode()[0], ms["k_sweep_phase1"], ms["k_sweep_phase1"] * 1e6 / (C / 2), ms["k_sweep_phase2"],
                ms["k_sweep_phase2"] * 1e6 / (C / 2), V / (sum(ms.values()) * 1e-3) / 1e6), flush=True)
            job.close()
        a, c = out["1"].likelihoods_ld(), out["0"].likelihoods_ld()
        den = np.maximum(np.abs(a), np.abs(c))
        print("      max relative difference of the two kernels' likelihoods: %.2e" % float(np.where(den > 0, np.abs(a - c) / np.where(den > 0, den, 1), 0).max()), flush=True)
