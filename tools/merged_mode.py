"""merged one-shot jobs (the reference's thread-pool pattern: one pg_hmm_genotype_contig call per contig,
merged by the library into one device job) always run CHUNKED, so that a caller's bits do not depend on who else was in
flight.  What that costs a wide pool: 64 / 128 / 256 concurrent callers, chunked (the product) vs fused (PG_SWEEP_MODE).
usage (GPU box; no torch needed): python tools/merged_mode.py [V] [H]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
from pangenie_amd import hmm
from pangenie_amd.panel import default_table_args, synthetic_panel

V = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
table = hmm.ProbabilityTable(*default_table_args())
params = hmm.make_params(1.26, False, 1e-5)
pool = [synthetic_panel(V, H, 20, seed=4242 + i, multiallelic_frac=0.2) for i in range(8)]
print(f"contigs of {V} variants x {H} haplotypes x 20 k-mers/variant, a fifth multiallelic; one host thread per contig")
for n in (64, 128, 256):
    batches = [pool[i % len(pool)] for i in range(n)]
    row = {}
    for mode in ("chunked", "fused"):
        os.environ["PG_SWEEP_MODE"] = mode
        into = None
        for _ in range(2):   # arena pool, pinned buffers
            got = hmm.genotype_contigs_threaded(batches, table, params)
        bad = [g for g in got if isinstance(g, Exception)]
        if bad:
            raise bad[0]
        st0, ms = hmm.coalesce_stats(), []
        for _ in range(3):
            t0 = time.perf_counter()
            hmm.genotype_contigs_threaded(batches, table, params)
            ms.append((time.perf_counter() - t0) * 1e3)
        st1 = hmm.coalesce_stats()
        row[mode] = (min(ms), (st1["merged_jobs"] - st0["merged_jobs"]) / 3.0)
        hmm._lib.load_hip().pg_hmm_release_cache()
    c, f = row["chunked"], row["fused"]
    print(f"{n:4d} callers: chunked {c[0]:8.1f} ms/round ({n * V / c[0] / 1e3:6.2f} M variants/s, {c[1]:.1f} device jobs/round)   "
          f"fused {f[0]:8.1f} ms/round ({n * V / f[0] / 1e3:6.2f} M variants/s, {f[1]:.1f} jobs/round)   fused/chunked {c[0] / f[0]:.2f}x", flush=True)
