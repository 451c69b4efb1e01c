import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from pangenie_amd import hmm
from pangenie_amd.panel import synthetic_panel, default_table_args
from oracle import pyoracle as orc
from tests.parity_util import rel_errors
for (V,H) in [(800,32),(600,64),(300,16)]:
    b = synthetic_panel(V, H, 20, seed=1000+V+H)
    args = default_table_args()
    res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5))
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    rel = rel_errors(b, res.likelihoods_ld(), ref.lik).astype(float)
    bad_bins = np.nonzero(rel > 1e-6)[0]
    goff = b.geno_off.astype(np.int64)
    bad_vars = np.unique(np.searchsorted(goff, bad_bins, side="right") - 1)
    cols = np.cumsum(res.kept) - 1
    print(V, H, "C", res.n_columns, "mid", res.n_columns // 2, "bad variants:", len(bad_vars), "their column idx:", cols[bad_vars][:20], "...", cols[bad_vars][-5:] if len(bad_vars) else "")
    if len(bad_vars):
        v = bad_vars[0]; print("  first bad: got", res.likelihoods_ld()[goff[v]:goff[v+1]], "exp", res.lik_exp[v], "ref", ref.lik[goff[v]:goff[v+1]])
    if len(bad_vars):
        L = res.likelihoods_ld(); kv = np.nonzero(res.kept)[0]
        m = res.n_columns // 2
        for c in range(m - 2, m + 4):
            v = kv[c]
            print("   col", c, "got", np.array(L[goff[v]:goff[v+1]], dtype=float), "ref", np.array(ref.lik[goff[v]:goff[v+1]], dtype=float), "exp", res.lik_exp[v])
