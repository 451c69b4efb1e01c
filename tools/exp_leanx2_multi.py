"""k_sweep_leanx2: what a column with 3-5 local alleles costs against one with two — phase 2 of a 256-chain 64-path cohort at
different shares of multiallelic objects.  usage: python tools/exp_leanx2_multi.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangenie_amd import hmm
from pangenie_amd.panel import default_table_args, synthetic_panel, synthetic_sample_counts

table, params = hmm.ProbabilityTable(*default_table_args()), hmm.make_params(1.26, False, 1e-5)
V, NC, S = 6000, 8, 32
for multi in (0.02, 0.2, 0.5, 1.0):
    index = [synthetic_panel(V, 64, 20, seed=777 + i, multiallelic_frac=multi) for i in range(NC)]
    pool = []
    for s in range(4):
        kcs, covs = zip(*[synthetic_sample_counts(ix, seed=100 * s + i) for i, ix in enumerate(index)])
        pool.append((list(kcs), list(covs)))
    job = hmm.Job.cohort(index, [pool[s % 4] for s in range(S)], table, params)
    for _ in range(2):
        job.run()
    km = {}
    for _ in range(3):
        job.run()
        for k, v in job.kernel_ms().items():
            km[k] = km.get(k, 0.0) + v / 3
    cols = sum(r.n_columns for r in job.fetch_all())
    plan = job.plan().splitlines()[1][:110]
    job.close()
    print(f"multi {multi:4.2f}: phase 1 {km['k_sweep_phase1']:6.2f}  phase 2 {km['k_sweep_phase2']:6.2f} ms  bins {km['k_bins']:5.2f}  ({cols / 1e6:.2f} M columns: {km['k_sweep_phase2'] * 1e3 / cols * 1e3:6.1f} ps per column)   {plan}", flush=True)
