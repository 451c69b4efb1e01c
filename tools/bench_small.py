"""Many H = 16 chains on one GPU: k_sweep_small16 (four half-chains per wave) against the general kernel.
usage (GPU box): python tools/bench_small.py [chains ...]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402

from pangenie_amd import hmm  # noqa: E402
from pangenie_amd.panel import default_table_args, synthetic_panel, synthetic_sample_counts  # noqa: E402

V = 8000
index = [synthetic_panel(V, 16, 20, seed=900 + i) for i in range(8)]
table = hmm.ProbabilityTable(*default_table_args())
params = hmm.make_params(1.26, False, 1e-5)
for n_chains in [int(a) for a in sys.argv[1:]] or [24, 512, 4096]:
    S = max(1, n_chains // 8)
    samples = []
    for s in range(S):
        kcs, covs = zip(*[synthetic_sample_counts(ix, seed=1000 * s + i) for i, ix in enumerate(index)])
        samples.append((list(kcs), list(covs)))
    for small in ("0", "1"):
        os.environ["PG_KERNELS"] = "small" if small == "1" else "nosmall"
        job = hmm.Job.cohort(index, samples, table, params)
        job.run()
        t0 = time.perf_counter()
        job.run()
        dt = time.perf_counter() - t0
        ms = job.kernel_ms()
        mode = job.sweep_mode()[0]
        r = job.fetch(S * 8 - 1)
        print("chains %5d small=%s %-8s run %8.2f ms  phase1 %8.2f phase2 %8.2f  %.1f M variants/s  (check %.6e)" % (
            S * 8, small, mode, dt * 1e3, ms["k_sweep_phase1"], ms["k_sweep_phase2"], S * 8 * V / dt / 1e6, float(np.abs(r.lik).sum())), flush=True)
        job.close()
