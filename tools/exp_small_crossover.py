"""Where does k_sweep_small16[x] (four 16-path half-chains per wave, a throughput kernel) overtake the general kernel (one half-chain per
wave, four states per lane)?  pg_shim.cpp switches at 256 chains (320 with multiallelic objects; 512 until round 6): this measures the step of a 16-path cohort at 64 ... 1024
chains with either kernel forced (PG_KERNELS=small / nosmall).  usage: python tools/exp_small_crossover.py [--multi 0.2]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pangenie_amd import hmm
from pangenie_amd.panel import default_table_args, synthetic_panel, synthetic_sample_counts

ap = argparse.ArgumentParser()
ap.add_argument("--multi", type=float, default=0.0)
ap.add_argument("--V", type=int, default=8000)
args = ap.parse_args()
NC = 8
index = [synthetic_panel(args.V, 16, 20, seed=777 + i, multiallelic_frac=args.multi) for i in range(NC)]
pool = []
for s in range(8):
    kcs, covs = zip(*[synthetic_sample_counts(ix, seed=100 * s + i) for i, ix in enumerate(index)])
    pool.append((list(kcs), list(covs)))
table, params = hmm.ProbabilityTable(*default_table_args()), hmm.make_params(1.26, False, 1e-5)
print(f"16 paths, multiallelic_frac {args.multi}, {NC} contigs x {args.V} variants per sample; step ms (M variants/s)")
for S in (8, 16, 32, 64, 128):
    row = []
    for tok in ("small", "nosmall"):
        os.environ["PG_KERNELS"] = tok
        os.environ["PG_SWEEP_MODE"] = "fused"
        job = hmm.Job.cohort(index, [pool[s % 8] for s in range(S)], table, params)
        for _ in range(2):
            job.run()
        t0 = time.perf_counter()
        for _ in range(4):
            job.run()
        ms = (time.perf_counter() - t0) / 4 * 1e3
        km = job.kernel_ms()
        job.close()
        row.append(f"{tok}: {ms:7.2f} ms ({S * NC * args.V / ms / 1e3:6.1f} M/s; sweeps {km.get('k_sweep_phase1', 0) + km.get('k_sweep_phase2', 0):6.2f})")
    print(f"  {S * NC:5d} chains   " + "   ".join(row), flush=True)
