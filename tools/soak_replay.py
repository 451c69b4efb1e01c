"""Replay one case of tools/soak_parity.py (same random stream) and list the bins that miss the bar,
in long double.  usage: python tools/soak_replay.py <case index> <seed0>"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import pyoracle as orc
from pangenie_amd import hmm
from pangenie_amd.panel import synthetic_panel
from tests.parity_util import rel_errors
target, seed0 = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0)
for it in range(target + 1):
    H = int(rng.choice([1, 2, 5, 13, 16, 17, 27, 32, 33, 50, 64, 65, 100, 128]))
    V = int(rng.integers(1, 900 if H <= 64 else 250))
    K = int(rng.choice([8, 20, 40, 128]))
    multi = float(rng.choice([0.0, 0.2, 0.6]))
    wide = bool(rng.random() < 0.3) and K >= 40
    kw = dict(multiallelic_frac=multi, undefined_frac=float(rng.choice([0.0, 0.05, 0.3])), zero_kmer_frac=float(rng.choice([0.0, 0.05])))
    if wide:
        kw.update(max_alleles=int(rng.integers(6, 33)), local_alts=int(rng.integers(5, 32)), multiallelic_frac=max(multi, 0.2))
    pseed = int(rng.integers(1 << 30))
    reg = float(rng.choice([0.01, 0.01, 0.0, 0.001]))
    par = [(1.26, False, 1e-5), (1.26, True, 1e-5), (0.001, False, 1e-5), (446.287102628, False, 0.25), (1.26, False, 25000.0)][int(rng.integers(5))]
    mode = str(rng.choice(["fused", "chunked", "chunked"]))
    chunk = str(int(rng.choice([1, 3, 16, 64, 4096])))
b = synthetic_panel(V, H, K, seed=pseed, **kw)
if reg == 0.0:
    b.kmer_count[::3] = 0
    b.kmer_count[1::17] = 300
args = (6, 108, 54, reg)
print(dict(H=H, V=V, K=K, kw=kw, reg=reg, par=par, mode=mode, chunk=chunk))
pa = b.path_allele.reshape(V, H)
nloc = np.array([len(set(r)) for r in pa])
A = np.diff(b.allele_off)
go = b.geno_off.astype(np.int64)
ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(*par))
kept_idx = np.nonzero(ref.kept)[0]
col_of = {int(v): c for c, v in enumerate(kept_idx)}
for ck in ("4096",):
    os.environ["PG_SWEEP_MODE"] = "chunked"
    os.environ["PG_CHUNK_COLS"] = ck
    res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params(*par))
    got = res.likelihoods_ld()
    r = rel_errors(b, got, ref.lik)
    bad = np.nonzero(r > 1e-6)[0]
    bv = sorted(set(int(np.searchsorted(go, x, side='right') - 1) for x in bad))
    print("chunk", ck, "ncols", res.n_columns, ref.n_columns, "C/2", res.n_columns // 2, "bad bins", bad.size, "bad variants", len(bv))
    print("  bad variants (v, column, nlocal, A):", [(v, col_of.get(v), int(nloc[v]), int(A[v])) for v in bv[:24]])
    for v in bv[:5]:
        Av = int(A[v]); a0 = int(b.allele_off[v])
        present = sorted(set(int(x) for x in pa[v]))
        ids = [int(x) for x in b.allele_id[a0:a0 + Av]]
        fl = [int(x) for x in b.allele_flags[a0:a0 + Av]]
        seg_g, seg_r = got[go[v]:go[v+1]], ref.lik[go[v]:go[v+1]]
        mx = float(np.max(np.abs(seg_r)))
        print("   v", v, "A", Av, "npresent", len(present), "lik_exp", int(res.lik_exp[v]), "raw lik of bad bins follow; device raw max", float(np.max(res.lik[go[v]:go[v+1]])))
        k = 0
        for sa in range(Av):
            for sb in range(sa, Av):
                idx = sa * Av - sa * (sa - 1) // 2 + (sb - sa)
                if r[go[v] + idx] > 1e-6:
                    print("      bin slots (%d,%d) got %s ref %s ratio got/ref %s ref/maxbin %s" % (sa, sb, str(seg_g[idx]), str(seg_r[idx]), str(seg_g[idx] / seg_r[idx]) if seg_r[idx] != 0 else "inf", str(seg_r[idx] / np.max(np.abs(seg_r)))))
                    k += 1
                    if k > 8: break
            if k > 8: break
print("wide variants:", [(int(v), col_of.get(int(v)), int(nloc[v])) for v in np.nonzero(nloc > 5)[0]][:30])
