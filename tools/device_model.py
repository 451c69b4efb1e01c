"""fp64 model of the DEVICE arithmetic (numpy, sequential) — design check, not product code.

The HIP kernels compute in fp64 where the reference computes in 80-bit long double.  This file
restates, state by state in plain numpy fp64, the scaled recursion the kernels implement
 so that its range/precision claims can be checked on a CPU against the oracle:

  * stored forward column  P'_t = (A x_{t-1} A^T) * 2^-es,  es = exponent(sum x_{t-1}) - BF
    — the column BEFORE the emission multiply.  Its entries are >= q^2 * (column sum): a bounded
    dynamic range for any positive recombination rate.  The recursion continues with
    x_t = P'_t . E'_t  (E' = emission table scaled by 2^-X_t, flushed fp64 is fine there);
  * stored backward column  beta'_t = (A (y . E'_{t+1}) A^T) * 2^-esb, esb = exponent(sum y) - BB;
  * posterior bins  L_t({a,b}) = e_t(a,b) * sum_{states in the bin} P'_t * beta'_t / (m_f m_b)
    with the emission of the bin applied ONCE, as (mantissa, exponent), after the sum — so a bin
    keeps full relative precision however small its emission is (down to the reference's own
    long double underflow);
  * output per bin: lik in [0.5,1) (or 0) and an int32 exponent.

usage (CPU, needs the oracle):  python tools/device_model.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LD = np.longdouble
BF = 400   # forward column bias  (stored P' columns sum to ~2^BF)
BB = 400   # backward column bias
LD_MIN_EXP = -16444


def pair_tables(batch, table, orc, v):
    """(mantissa fp64, exponent) of the emission product of every allele-slot pair of variant v,
    after the all_zeros rule (reference src/emissionprobabilitycomputer.cpp:9-34)."""
    E, all_zeros = orc.emission_table(batch, table, v)
    m, e = np.frexp(E)  # long double frexp
    m = m.astype(np.float64)
    e = e.astype(np.int64)
    m[E == 0] = 0.0
    e[E == 0] = 0
    return m, e, all_zeros


def one_minus_exp_neg_like_reference(x):
    t = -np.expm1(-x)
    if x < 0.6931471805599453:
        s = t * 2.0 ** 64
        if s < 2.0 ** 52:
            t = np.rint(s) * 2.0 ** -64
    return t


def transition_consts(d, H, uniform):
    if uniform:
        c0, c1, c2 = 0.0, 0.0, 1.0
    else:
        x = d / H
        r = np.exp(-x)
        q = one_minus_exp_neg_like_reference(x) / H
        c0, c1, c2 = r * r, q * r, q * q
    kappa = c0 + 2.0 * H * c1 + H * H * c2
    return c0, c1, c2, kappa


def exponent_of(x):
    return int(np.frexp(x)[1])


def genotype_contig_model(batch, table_args, recomb, uniform, eff_N, orc):
    """Returns (lik mantissa fp64 [sumG], lik_exp int64 [sumG], kept)."""
    V, H = batch.n_variants, batch.n_paths
    table = orc.OracleTable(*table_args)
    geno_off = batch.geno_off.astype(np.int64)
    lik = np.zeros(int(geno_off[-1]))
    lexp = np.zeros(int(geno_off[-1]), np.int64)
    pa = batch.path_allele.reshape(V, H)
    kept = np.zeros(V, np.uint8)
    cols = []
    for v in range(V):
        a0, a1 = int(batch.allele_off[v]), int(batch.allele_off[v + 1])
        ids = list(batch.allele_id[a0:a1])
        slots = np.array([ids.index(a) for a in pa[v]])
        nonref = any(pa[v][p] != 0 and not (batch.allele_flags[a0 + slots[p]] & 1) for p in range(H))
        if nonref:
            kept[v] = 1
            cols.append((v, slots, a1 - a0))
    C = len(cols)
    if C == 0:
        return lik, lexp, kept
    dist_scale = float(LD(0.000004) * LD(recomb) * LD(eff_N))
    # per column: pair tables, X, E' on the states
    Em, Ee, Xs, Estate, recs = [], [], [], [], []
    for (v, slots, A) in cols:
        m, e, az = pair_tables(batch, table, orc, v)
        present = sorted(set(slots))
        sub = [(a, b) for a in present for b in present]
        if az:
            X = 0
        else:
            ex = [e[a, b] for (a, b) in sub if m[a, b] > 0]
            X = max(ex) if ex else 0
        Ep = np.zeros((A, A))
        for (a, b) in sub:
            Ep[a, b] = np.ldexp(m[a, b], int(e[a, b] - X)) if m[a, b] > 0 else m[a, b]
        Em.append(m); Ee.append(e); Xs.append(int(X))
        Estate.append(Ep[np.ix_(slots, slots)])
    unif = 1.0 / (H * H)
    P = [None] * C        # stored forward columns (pre-emission)
    fscale = np.ones(C)
    fbias = np.full(C, BF)
    fb = np.zeros(C, bool)
    x = None
    # ---- forward ----
    for t in range(C):
        v = cols[t][0]
        if t == 0:
            Pt = np.full((H, H), 2.0 ** BF)
            fscale[0] = 1.0
        else:
            pv = cols[t - 1][0]
            d = float(int(batch.variant_pos[v]) - int(batch.variant_pos[pv])) * dist_scale
            c0, c1, c2, kappa = transition_consts(d, H, uniform)
            S = x.sum()
            if not S > 0.0:
                # the previous column summed to zero: the reference replaces it by the uniform column, fsum := 1
                fb[t - 1] = True
                P[t - 1] = np.full((H, H), unif)
                x = np.full((H, H), unif)
                S = 1.0
            Cs = x.sum(axis=0)
            es = exponent_of(S) - BF
            sc = 2.0 ** (-es)
            fscale[t] = np.ldexp(S, -exponent_of(S))
            Pt = (c0 * sc) * x + ((c1 * Cs)[:, None] * sc + ((c1 * Cs)[None, :] + c2 * S) * sc)
        P[t] = Pt
        x = Pt * Estate[t]
    if not x.sum() > 0.0:
        fb[C - 1] = True
        P[C - 1] = np.full((H, H), unif)
    # ---- backward + bins ----
    y = np.ones((H, H))
    Sy = float(H * H)
    bscale_t = 1.0
    bbias_t = 0
    for t in range(C - 1, -1, -1):
        v, slots, A = cols[t]
        if t < C - 1:
            nv = cols[t + 1][0]
            d = float(int(batch.variant_pos[nv]) - int(batch.variant_pos[v])) * dist_scale
            c0, c1, c2, kappa = transition_consts(d, H, uniform)
            if not Sy > 0.0:
                y = np.full((H, H), unif)
                Sy = 1.0
            es = exponent_of(Sy) - BB
            bscale_t = np.ldexp(Sy, -exponent_of(Sy))
            bbias_t = BB
            w = y * Estate[t + 1]
            Ws = w.sum(axis=0)
            Sw = w.sum()
            k0, k1, k2 = c0 * 2.0 ** -es, c1 * 2.0 ** -es, c2 * 2.0 ** -es
            y = k0 * w + ((k1 * Ws)[:, None] + ((k1 * Ws)[None, :] + k2 * Sw))
            Sy = (kappa * 2.0 ** -es) * Sw
            if not Sy > 0.0:
                y = np.zeros((H, H))  # own posteriors are 0; the next step starts from the uniform column
        prod = P[t] * y
        Xn = Xs[t + 1] if t + 1 < C else 0
        fs = 1.0 if fb[t] else fscale[t]
        bias = (0 if fb[t] else BF) + bbias_t
        present = sorted(set(slots))
        for ia, a in enumerate(present):
            for b in present[ia:]:
                mask = (slots[:, None] == a) & (slots[None, :] == b)
                if a != b:
                    mask |= (slots[:, None] == b) & (slots[None, :] == a)
                s = prod[mask].sum() / (fs * bscale_t)
                pm, pe = (0.5, 1) if fb[t] else (Em[t][a, b], int(Ee[t][a, b]))
                val = s * pm
                mm, ee = np.frexp(val)
                idx = geno_off[v] + a * A - a * (a - 1) // 2 + (b - a)
                if val == 0.0:
                    lik[idx], lexp[idx] = 0.0, 0
                else:
                    E = int(ee) + pe + Xn - bias
                    if E < LD_MIN_EXP:
                        lik[idx], lexp[idx] = 0.0, 0
                    else:
                        lik[idx], lexp[idx] = mm, E
    return lik, lexp, kept


def compare(batch, table_args, params, orc):
    recomb, uniform, N = params
    lik, lexp, kept = genotype_contig_model(batch, table_args, recomb, uniform, N, orc)
    ref = orc.genotype_contig(batch, orc.OracleTable(*table_args), orc.make_params(recomb, uniform, N))
    got = np.ldexp(lik.astype(LD), lexp)
    den = np.maximum(np.abs(got), np.abs(ref.lik))
    rel = np.where(den > 0, np.abs(got - ref.lik) / np.where(den > 0, den, LD(1)), LD(0))
    assert (kept == ref.kept).all()
    return float(rel.max()) if rel.size else 0.0, got, ref.lik


if __name__ == "__main__":
    from oracle import pyoracle as orc
    from pangenie_amd.panel import synthetic_panel
    cases = [
        ("biallelic H=16", synthetic_panel(300, 16, 20, seed=7), (6, 108, 54, 0.01), (1.26, False, 1e-5)),
        ("unregularised zeros H=16", None, (6, 108, 54, 0.0), (1.26, False, 1e-5)),
        ("16-allele columns H=65 K=128 reg=0", synthetic_panel(120, 65, 128, seed=99, multiallelic_frac=0.6, max_alleles=17, local_alts=15, undefined_frac=0.05), (6, 108, 54, 0.0), (1.26, False, 1e-5)),
        ("same, recomb 1e-3", synthetic_panel(80, 33, 128, seed=98, multiallelic_frac=0.6, max_alleles=17, local_alts=15), (6, 108, 54, 0.0), (0.001, False, 1e-5)),
        ("uniform", synthetic_panel(100, 13, 40, seed=5, multiallelic_frac=0.3), (6, 108, 54, 0.01), (1.26, True, 1e-5)),
        ("N=25000", synthetic_panel(100, 13, 40, seed=6, multiallelic_frac=0.3), (6, 108, 54, 0.01), (1.26, False, 25000.0)),
    ]
    for name, b, targs, par in cases:
        if b is None:
            b = synthetic_panel(300, 16, 20, seed=5)
            b.kmer_count[::3] = 0
            b.kmer_count[1::17] = 300
        elif targs[3] == 0.0:
            b.kmer_count[::3] = 0
            b.kmer_count[1::17] = 300
        worst, got, ref = compare(b, targs, par, orc)
        nz = ref[ref > 0]
        span = float(np.log10(nz.max() / nz.min())) if nz.size else 0.0
        print(f"{name:40s} max rel err {worst:.3e}   (bins span {span:.0f} decades overall)")
