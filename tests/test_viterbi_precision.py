"""Why the device Viterbi keeps its columns in double-double — on the CPU, no GPU needed.

With few paths and the reference's default effective_N, exp(-d/H) drops below 1e-16 of q: p and q agree to all 53 bits
of a double while the reference's 80-bit products still tell "stay" (p^2) from "switch" (pq).  Three restatements of
the recursion on the same inputs (the oracle's long double emission tables and transition probabilities):
  exact  — rational arithmetic: the decisions exact arithmetic makes (what double-double reproduces),
  oracle — the long double restatement of the reference's loop (oracle/pg_oracle.c),
  fp64   — the same four-candidate step in plain doubles,
  dd     — the device's arithmetic (pg_viterbi.hip): double-double values, products by FMA (emulated exactly with
           rationals), transition probabilities as exact (hi, lo) pairs of the long doubles, emissions as doubles,
           columns rescaled by powers of two.
exact == oracle wherever the margin clears the reference's own rounding (the panels below keep p - q at 35+ ulps of its
64-bit mantissa); plain fp64 does not.  Closer than a few of those ulps the reference decides on its own
rounding noise, and no arithmetic but a bit-exact x87 emulation in the reference's order would follow it."""
from fractions import Fraction

import numpy as np

from oracle import pyoracle as orc
from pangenie_amd.panel import default_table_args, synthetic_panel


def frac(x):
    n, d = np.longdouble(x).as_integer_ratio()
    return Fraction(int(n), int(d))


def viterbi_four_candidates(batch, table, recomb, eff_n, number, emission=None, rescale=False):
    """The O(H^2) step with the reference's tie rule over `number`-typed values (`emission`: type of the emission
    probabilities if different; `rescale`: multiply every column by the power of two that brings its maximum into
    [1/2, 1), as the device does); returns the haplotype alleles per variant."""
    emission = emission or number
    V, H = batch.n_variants, batch.n_paths
    pa = batch.path_allele.reshape(V, H)
    cols = [v for v in range(V) if any(pa[v, p] != 0 and not (batch.allele_flags[batch.allele_off[v] + list(batch.allele_id[batch.allele_off[v]:batch.allele_off[v + 1]]).index(pa[v, p])] & 1) for p in range(H))]
    n = H * H
    prev, backs = None, []
    for c, v in enumerate(cols):
        E, _ = orc.emission_table(batch, table, v)
        ids = list(batch.allele_id[batch.allele_off[v]:batch.allele_off[v + 1]])
        slot = [ids.index(a) for a in pa[v]]
        e = [[emission(E[slot[i], slot[j]]) for j in range(H)] for i in range(H)]
        cur = [None] * n
        back = [0] * n
        if c == 0:
            for i in range(H):
                for j in range(H):
                    cur[i * H + j] = number(e[i][j]) if emission is not number else e[i][j]
        else:
            t = [number(x) for x in orc.transition_probs(int(batch.variant_pos[cols[c - 1]]), int(batch.variant_pos[v]), recomb, H, False, eff_n)]
            zero = number(0.0)
            rowmax, rowidx = [zero] * H, [0] * H
            colmax, colidx = [zero] * H, [0] * H
            gmax, gidx = zero, 0
            for s in range(n):
                a, b = divmod(s, H)
                if prev[s] >= rowmax[a]: rowmax[a], rowidx[a] = prev[s], s
                if prev[s] >= colmax[b]: colmax[b], colidx[b] = prev[s], s
                if prev[s] >= gmax: gmax, gidx = prev[s], s
            for i in range(H):
                for j in range(H):
                    s = i * H + j
                    best, bi = zero, 0
                    for val, idx in ((prev[s] * t[0], s), (rowmax[i] * t[1], rowidx[i]), (colmax[j] * t[1], colidx[j]), (gmax * t[2], gidx)):
                        if val > best or (val == best and idx >= bi):
                            best, bi = val, idx
                    if best == zero:
                        bi = n - 1
                    back[s] = bi
                    cur[s] = best * e[i][j]
        top = max(cur, key=lambda x: x.key()) if rescale else max(cur)
        if top == 0:
            cur = [number(1.0)] * n
        elif rescale:  # (hi, lo) both by the exact power of two that brings the maximum's hi into [1/2, 1)
            sc = 2.0 ** (-int(np.floor(np.log2(top.hi))) - 1)
            cur = [DD(x.hi * sc, x.lo * sc) for x in cur]
        elif number is float:  # an exact power of two keeps the doubles in range and changes no comparison
            sc = 2.0 ** (-int(np.floor(np.log2(top))) - 1)
            cur = [x * sc for x in cur]
        prev = cur
        backs.append(back)
    best, bv = 0, prev[0] - prev[0]
    for s in range(n):
        if prev[s] >= bv: bv, best = prev[s], s
    states = [0] * len(cols)
    for c in range(len(cols) - 1, -1, -1):
        states[c] = best
        if c > 0:
            best = backs[c][best]
    hap1 = np.zeros(V, np.uint16); hap2 = np.zeros(V, np.uint16)
    for c, v in enumerate(cols):
        hap1[v], hap2[v] = pa[v, states[c] // H], pa[v, states[c] % H]
    return hap1, hap2


def windowed_panel(V, H, seed):
    """gaps chosen so that distance / H lies in [39.5, 41.5], exp(-d/H) in [1e-18, 7e-18]: p and q agree to all 53 bits
    of a double, yet differ by 35+ ulps of the reference's 64-bit mantissa — well above its rounding"""
    from pangenie_amd.panel import ContigBatch
    b = synthetic_panel(V, H, 20, seed=seed, zero_kmer_frac=0.08)
    rng = np.random.default_rng(seed)
    gaps = 314 * H + rng.integers(0, 15 * H, size=V)  # distance / H = 0.126 * gap / H
    pos = (10000 + np.cumsum(gaps)).astype(np.uint64)
    return ContigBatch(H, pos, b.coverage, b.kmer_off, b.kmer_count, b.allele_off, b.allele_id, b.allele_flags,
                       b.allele_kmer_off, b.allele_kmer_mask, b.path_allele)


def test_exact_decisions_are_the_references_and_plain_fp64_is_not():
    table = orc.OracleTable(*default_table_args())
    recomb, eff_n = 1.26, 25000.0  # the reference's default effective_N
    total_fp64 = 0
    for (V, H, seed) in ((400, 2, 7402), (300, 3, 11), (200, 4, 12)):
        batch = windowed_panel(V, H, seed)
        t = orc.transition_probs(int(batch.variant_pos[0]), int(batch.variant_pos[1]), recomb, H, False, eff_n)
        # "stay" beats "one switch" by less than half an ulp of a double, but by 35+ ulps of a long double
        assert t[0] > t[1] > t[2] and (t[0] - t[1]) / t[1] < np.longdouble(2.0) ** -53 and (t[0] - t[1]) / t[1] > 30 * np.longdouble(2.0) ** -64
        ref = orc.viterbi_contig(batch, table, orc.make_params(recomb, False, eff_n, run_genotyping=False, run_phasing=True), form=0)
        x1, x2 = viterbi_four_candidates(batch, table, recomb, eff_n, frac)
        assert np.array_equal(x1, ref.hap1) and np.array_equal(x2, ref.hap2), (V, H, "exact arithmetic vs the long double oracle")
        f1, f2 = viterbi_four_candidates(batch, table, recomb, eff_n, float)
        total_fp64 += int(np.sum((f1 != ref.hap1) | (f2 != ref.hap2)))
    assert total_fp64 > 0, "plain fp64 was expected to miss decisions that hang on 1e-17 relative differences"


# ---------------------------------------------------------------------------------------------------------------------
#  the device's double-double arithmetic, emulated exactly (an FMA is one rounding of an exact rational)
# ---------------------------------------------------------------------------------------------------------------------
def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))


class DD:
    """value = hi + lo, |lo| <= ulp(hi) / 2; ordered by (hi, lo) — pg_viterbi.hip: struct dd, dd_mul, dd_mul_d"""
    __slots__ = ("hi", "lo")

    def __init__(self, hi, lo=0.0):
        self.hi, self.lo = hi, lo

    @staticmethod
    def of_longdouble(x):
        hi = float(x)
        return DD(hi, float(np.longdouble(x) - np.longdouble(hi)))

    def __mul__(self, o):
        if isinstance(o, DD):
            p = self.hi * o.hi
            e = fma(self.hi, o.hi, -p)
            e = fma(self.hi, o.lo, e)
            e = fma(self.lo, o.hi, e)
        else:  # a double (an emission probability)
            p = self.hi * o
            e = fma(self.hi, o, -p)
            e = fma(self.lo, o, e)
        s = p + e
        return DD(s, e - (s - p))

    def key(self):
        return (self.hi, self.lo)

    def __gt__(self, o): return self.key() > o.key()
    def __ge__(self, o): return self.key() >= o.key()
    def __eq__(self, o): return self.key() == (o.key() if isinstance(o, DD) else (o, 0.0))
    def __sub__(self, o): return DD(0.0)  # (only used to form a typed zero)


def dd_number(x):
    """the `number` hook of viterbi_four_candidates: long double transition probabilities -> exact (hi, lo) pairs,
    emission probabilities -> plain doubles (k_prep's products are doubles), 0.0 / 1.0 literals -> DD"""
    if isinstance(x, float):
        return DD(x)
    return DD.of_longdouble(x)


def test_the_device_arithmetic_follows_the_oracle():
    """double-double as pg_viterbi.hip specifies it, on the windowed panels above and on ordinary ones"""
    table = orc.OracleTable(*default_table_args())
    cases = [(windowed_panel(300, 2, 7402), 1.26, 25000.0), (windowed_panel(200, 3, 11), 1.26, 25000.0),
             (synthetic_panel(200, 3, 20, seed=21, zero_kmer_frac=0.05), 1.26, 1e-5),
             (synthetic_panel(150, 4, 20, seed=22), 446.287102628, 0.25)]
    for batch, recomb, eff_n in cases:
        ref = orc.viterbi_contig(batch, table, orc.make_params(recomb, False, eff_n, run_genotyping=False, run_phasing=True), form=0)
        d1, d2 = viterbi_four_candidates(batch, table, recomb, eff_n, dd_number, emission=float, rescale=True)
        assert np.array_equal(d1, ref.hap1) and np.array_equal(d2, ref.hap2), (batch.n_variants, batch.n_paths, recomb, eff_n)
