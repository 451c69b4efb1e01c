import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import pyoracle as orc
from pangenie_amd import hmm
from pangenie_amd.panel import synthetic_panel
from tests.parity_util import rel_errors
# replay soak iteration `target` of seed0
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
    b.kmer_count[1::17] = 60000
args = (6, 108, 54, reg)
print(dict(H=H, V=V, K=K, kw=kw, reg=reg, par=par, mode=mode, chunk=chunk))
ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(*par))
for m in ("fused", "chunked"):
    os.environ["PG_SWEEP_MODE"] = m
    os.environ["PG_CHUNK_COLS"] = chunk
    res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params(*par))
    r = rel_errors(b, res.likelihoods_ld(), ref.lik)
    bad = np.nonzero(r > 1e-6)[0]
    go = b.geno_off.astype(np.int64)
    print(m, "n_cols", res.n_columns, ref.n_columns, "bad bins", bad.size, "first bad variants", sorted(set(np.searchsorted(go, bad, side='right') - 1))[:10])
    for bi in bad[:4]:
        v = int(np.searchsorted(go, bi, side='right') - 1)
        print("  v", v, "A", b.allele_off[v+1]-b.allele_off[v], "got", res.likelihoods_ld()[go[v]:go[v+1]], "ref", ref.lik[go[v]:go[v+1]], "paths", b.path_allele.reshape(V, H)[v])
