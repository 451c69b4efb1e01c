"""Randomised parity soak (GPU box): many seeded panels of mixed shape, both sweep modes, small chunk
sizes, narrow and wide columns, regularised and unregularised tables — HIP path vs the oracle.
usage: python tools/soak_parity.py [n_panels] [seed0]   (80 panels: about two minutes)
SOAK_TRI=1: only all-biallelic H = 64 panels in fused mode (triangle storage, k_sweep_lean2), from 1 variant up.
SOAK_X=1: only 16-path panels on k_sweep_small16[x] (PG_KERNELS=small[,nosmall2]): multiallelic and wide objects (6-12 alleles of
which the sixteen paths carry up to nine), both sweep modes.
SOAK_PERSIST=1: only all-biallelic H = 64 panels in chunked mode on the persistent phase-2 pair (PG_KERNELS=persist), even chunk sizes.
SOAK_LX2=1: only H = 64 panels with 3-5-allele objects in fused mode (triangle storage, phase 2 on k_sweep_leanx2 + k_bins_q; phase 1
on k_sweep_leanx_tri or, PG_KERNELS=noleanx, the general kernel with triangle stores), from 1 variant up; the job's plan is checked.
SOAK_WIDEF=1: only 41 ... 64-path panels with objects of 6 ... 69 alleles (wide columns) in fused mode (DevContig::widef: the job must stay
fused; phase 1 on k_sweep_leanx_triw / k_sweep_tri1 / the general kernel, phase 2 through aux slots + k_bins_wide), from 1 variant up."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import pyoracle as orc
from pangenie_amd import hmm
from pangenie_amd.panel import default_table_args, synthetic_panel
from tests.parity_util import assert_parity, rel_errors

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed0)
worst, t0 = 0.0, time.time()
for it in range(n):
    H = int(rng.choice([1, 2, 5, 13, 16, 17, 17, 20, 24, 27, 30, 32, 33, 50, 64, 64, 64, 65, 100, 128, 129, 200, 300]))
    V = int(rng.integers(1, 900 if H <= 64 else (250 if H <= 128 else 60)))
    K = int(rng.choice([8, 20, 40, 128]))
    multi = float(rng.choice([0.0, 0.2, 0.6]))
    wide = bool(rng.random() < 0.3) and K >= 40
    kw = dict(multiallelic_frac=multi, undefined_frac=float(rng.choice([0.0, 0.05, 0.3])), zero_kmer_frac=float(rng.choice([0.0, 0.05])))
    if wide:
        kw.update(max_alleles=int(rng.integers(6, 70)), local_alts=int(rng.integers(5, 60)), multiallelic_frac=max(multi, 0.2))
    if os.environ.get("SOAK_TRI") == "1":
        H, wide = 64, False
        V = int(rng.choice([1, 2, 3, 4, 5, 7, 64, 65, 129, int(rng.integers(6, 900))]))
        kw.update(multiallelic_frac=0.0)
    if os.environ.get("SOAK_PERSIST") == "1":
        H, wide = 64, False
        V = int(rng.choice([1, 2, 3, 5, 64, 65, 129, 257, int(rng.integers(6, 900)), int(rng.integers(6, 900))]))
        kw.update(multiallelic_frac=0.0)
    if os.environ.get("SOAK_LX2") == "1":
        H, wide = 64, False
        V = int(rng.choice([1, 2, 3, 4, 5, 7, 15, 16, 17, 33, 48, 64, 65, 129, int(rng.integers(6, 900)), int(rng.integers(6, 900))]))
        kw.update(multiallelic_frac=float(rng.choice([0.05, 0.2, 0.6, 1.0])))
        kw.pop("max_alleles", None); kw.pop("local_alts", None)
    if os.environ.get("SOAK_WIDEF") == "1":
        H, wide, K = int(rng.choice([64, 64, 64, 41, 50, 63])), True, 40
        V = int(rng.choice([1, 2, 3, 4, 5, 7, 15, 16, 17, 33, 64, 65, 129, int(rng.integers(6, 900)), int(rng.integers(6, 900))]))
        kw.update(max_alleles=int(rng.integers(6, 70)), local_alts=int(rng.integers(5, 60)), multiallelic_frac=float(rng.choice([0.2, 0.6, 1.0])))
    if os.environ.get("SOAK_X") == "1":
        H, wide = 16, False
        V = int(rng.choice([2, 3, 4, 5, 7, 9, 15, 16, 17, 64, 65, 129, int(rng.integers(6, 900)), int(rng.integers(6, 900))]))
        kw.update(multiallelic_frac=float(rng.choice([0.0, 0.2, 0.45, 1.0])), wide_frac=float(rng.choice([0.0, 0.0, 0.03, 0.3])) if V > 1 else 0.0)
        kw.pop("max_alleles", None); kw.pop("local_alts", None)
    b = synthetic_panel(V, H, K, seed=int(rng.integers(1 << 30)), **kw)
    reg = float(rng.choice([0.01, 0.01, 0.0, 0.001]))
    if reg == 0.0:
        b.kmer_count[::3] = 0
        # out-of-table counts: the oracle evaluates them like the reference, with a loop over the count
        # per lookup — keep them small here (tests/ covers 60000 on a small panel); this tool runs on
        # the GPU box and must not burn GPU-minutes on CPU work
        b.kmer_count[1::17] = 300
    args = (6, 108, 54, reg)
    recomb, uniform, N = [(1.26, False, 1e-5), (1.26, True, 1e-5), (0.001, False, 1e-5), (446.287102628, False, 0.25), (1.26, False, 25000.0)][int(rng.integers(5))]
    mode = str(rng.choice(["fused", "chunked", "chunked"]))
    if os.environ.get("SOAK_TRI") == "1":
        mode = "fused"
    os.environ["PG_SWEEP_MODE"] = mode
    kern = str(rng.choice(["", "", "", "general", "generic", "prepwave", "small", "small,nosmall2", "fullcols", "nocls4"]))
    if os.environ.get("SOAK_TRI") == "1":
        kern = ""
    if os.environ.get("SOAK_LX2") == "1":
        os.environ["PG_SWEEP_MODE"] = mode = "fused"
        kern = str(rng.choice(["", "", "noleanx"]))
    if os.environ.get("SOAK_WIDEF") == "1":
        os.environ["PG_SWEEP_MODE"] = mode = "fused"
        kern = str(rng.choice(["", "", "", "noleanx", "general", "prepwave", "notri"]))
    if os.environ.get("SOAK_X") == "1":
        kern = str(rng.choice(["small", "small", "small", "small,nosmall2", "small,prepwave"]))
    if kern:
        os.environ["PG_KERNELS"] = kern
    else:
        os.environ.pop("PG_KERNELS", None)
    os.environ["PG_CHUNK_COLS"] = str(int(rng.choice([1, 3, 16, 64, 4096])))
    if os.environ.get("SOAK_PERSIST") == "1":
        os.environ["PG_SWEEP_MODE"] = mode = "chunked"
        os.environ["PG_KERNELS"] = kern = "persist"
        os.environ["PG_CHUNK_COLS"] = str(int(rng.choice([2, 16, 64, 4096])))
    if os.environ.get("SOAK_LX2") == "1" or os.environ.get("SOAK_WIDEF") == "1":
        job = hmm.Job([b], hmm.ProbabilityTable(*args), hmm.make_params(recomb, uniform, N))
        if os.environ.get("SOAK_WIDEF") == "1":
            assert job.sweep_mode()[0] == "fused", job.plan()
            n_wide_f = sum(len(set(r)) > 5 for r in b.path_allele.reshape(b.n_variants, b.n_paths))
            globals()["wide_cols_seen"] = globals().get("wide_cols_seen", 0) + n_wide_f
        elif int(np.diff(b.allele_off.astype(np.int64)).max()) > 2:
            assert "k_sweep_leanx2" in job.plan() and "k_bins_q" in job.plan(), job.plan()
        job.run()
        res = job.fetch(0)
        job.close()
    else:
        res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params(recomb, uniform, N))
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(recomb, uniform, N))
    try:
        assert_parity(b, res, ref)
    except AssertionError as e:
        print("FAIL", it, dict(H=H, V=V, K=K, kw=kw, reg=reg, recomb=recomb, uniform=uniform, N=N, mode=mode, kern=kern, chunk=os.environ["PG_CHUNK_COLS"]), str(e)[:300])
        sys.exit(1)
    r = rel_errors(b, res.likelihoods_ld(), ref.lik)
    rm = float(r.max()) if r.size else 0.0
    if rm > 1e-8:
        print("note", it, f"{rm:.2e}", dict(H=H, V=V, K=K, reg=reg, recomb=recomb, uniform=uniform, N=N, mode=mode, wide=wide))
    worst = max(worst, rm)
print(f"soak OK: {n} panels, worst relative error {worst:.3e}, {time.time() - t0:.0f} s" + (f", {globals().get('wide_cols_seen', 0)} wide columns" if os.environ.get("SOAK_WIDEF") == "1" else ""))
