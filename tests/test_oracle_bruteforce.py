"""Second opinion on the CPU oracle: a dense-matrix restatement of the recursion in numpy long double,
written from the maths (reference src/hmm.cpp:175-405,
src/transitionprobabilitycomputer.cpp:14-18, src/columnindexer.cpp:24-31) and sharing no code with
oracle/pg_oracle.c: columns are H x H matrices, the transition is the explicit matrix
A = (p-q) I + q 11^T applied from both sides, genotype bins are filled by looping over states.
Emission tables are taken from the oracle's own entry point (pinned separately on the reference's
EmissionProbabilityComputer tests); everything downstream of them is independent.  Small panels
only (pure-Python loops), including columns with more than five alleles on the paths and more than
ten alleles per variant, which no golden vector of the reference covers."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from pangenie_amd.panel import default_table_args, synthetic_panel

LD = np.longdouble


def brute_force(b, table, recombrate, uniform, effective_N):
    V, H = b.n_variants, b.n_paths
    pa = b.path_allele.reshape(V, H).astype(np.int64)
    aoff = b.allele_off.astype(np.int64)
    geno_off = b.geno_off.astype(np.int64)
    lik = np.zeros(int(geno_off[-1]), dtype=LD)
    # --- column selection: kept iff a selected path carries a defined non-reference allele
    slot_of, kept = [], []
    for v in range(V):
        ids = [int(x) for x in b.allele_id[aoff[v]:aoff[v + 1]]]
        flags = b.allele_flags[aoff[v]:aoff[v + 1]]
        slots = np.array([ids.index(int(a)) for a in pa[v]])
        slot_of.append(slots)
        kept.append(any(ids[s] != 0 and not (flags[s] & 1) for s in slots))
    cols = [v for v in range(V) if kept[v]]
    if not cols:
        return lik, np.array(kept, np.uint8)
    # --- emissions per column as H x H matrices
    em = []
    for v in cols:
        E, all_zeros = orc.emission_table(b, table, v)
        s = slot_of[v]
        em.append(np.ones((H, H), LD) if all_zeros else E[np.ix_(s, s)].astype(LD))
    # --- transition matrices of the gaps
    ones = np.ones((H, H), LD)
    trans = [None]
    for c in range(1, len(cols)):
        if uniform:
            trans.append(None)
            continue
        d = LD(int(b.variant_pos[cols[c]]) - int(b.variant_pos[cols[c - 1]])) * LD("0.000004") * LD(recombrate) * LD(effective_N)
        q = (LD(1) - np.exp(-d / LD(H))) / LD(H)
        p = np.exp(-d / LD(H)) + q
        trans.append((p - q) * np.eye(H, dtype=LD) + q * ones)

    def step(M, c):  # A M A^T for the gap c-1 -> c ; uniform: every entry = sum(M)
        return ones * M.sum() if trans[c] is None else trans[c] @ M @ trans[c].T

    C = len(cols)
    unif = ones / LD(H * H)
    # --- forward: alpha_hat (normalised) and fsum per column
    alpha, fsum = [], []
    for c in range(C):
        v = em[c] if c == 0 else em[c] * step(alpha[c - 1], c)
        s = v.sum()
        if s > 0:
            alpha.append(v / s); fsum.append(s)
        else:
            alpha.append(unif.copy()); fsum.append(LD(1))
    # --- backward + posteriors
    beta_hat_next = None
    for c in range(C - 1, -1, -1):
        bt = ones.copy() if c == C - 1 else step(beta_hat_next * em[c + 1], c + 1)
        v = cols[c]
        A = int(aoff[v + 1] - aoff[v])
        post = alpha[c] * bt * fsum[c]
        s = slot_of[v]
        for i in range(H):
            for j in range(H):
                a, bb = (s[i], s[j]) if s[i] <= s[j] else (s[j], s[i])
                lik[geno_off[v] + a * A - a * (a - 1) // 2 + (bb - a)] += post[i, j]
        sb = bt.sum()
        beta_hat_next = bt / sb if sb > 0 else unif.copy()
    return lik, np.array(kept, np.uint8)


CASES = [
    # V, H, K, synthetic_panel kwargs, (recombrate, uniform, effective_N)
    (25, 4, 20, dict(multiallelic_frac=0.3), (1.26, False, 1e-5)),
    (30, 6, 24, dict(multiallelic_frac=0.5, undefined_frac=0.2, zero_kmer_frac=0.1), (1.26, False, 25000.0)),
    (20, 5, 20, dict(multiallelic_frac=0.0), (1.26, True, 1e-5)),
    (18, 9, 64, dict(multiallelic_frac=0.8, max_alleles=12, local_alts=8), (1.26, False, 1e-5)),        # wide columns
    (14, 12, 96, dict(multiallelic_frac=1.0, max_alleles=24, local_alts=11, undefined_frac=0.1), (446.287102628, False, 0.25)),
    (16, 3, 96, dict(multiallelic_frac=0.7, max_alleles=32), (0.001, False, 1e-5)),                       # many alleles, few on paths
    # the default production shape (15 sampled paths + the reference path) as the round-5 parity tests and bench lines build it:
    # a fifth .. half of the objects with 3-5 alleles, some bubbles of 6-12 alleles of which the paths carry up to nine
    (12, 16, 40, dict(multiallelic_frac=0.45, wide_frac=0.25), (1.26, False, 1e-5)),
]


@pytest.mark.parametrize("V,H,K,kw,par", CASES)
def test_oracle_matches_dense_matrix_restatement(V, H, K, kw, par):
    b = synthetic_panel(V, H, K, seed=900 + V + H, **kw)
    if kw.get("local_alts", 0) > 5 or kw.get("wide_frac", 0) > 0:  # these cases are meant to contain wide columns
        assert max(len(set(r)) for r in b.path_allele.reshape(V, H)) > 5
    for reg in (0.01, 0.0):
        if reg == 0.0:
            b.kmer_count[::3] = 0  # exact zeros: uniform fallbacks, all_zeros
        args = default_table_args()[:3] + (reg,)
        table = orc.OracleTable(*args)
        ref = orc.genotype_contig(b, table, orc.make_params(*par))
        lik, kept = brute_force(b, table, *par)
        assert (kept == ref.kept).all()
        den = np.maximum(np.abs(lik), np.abs(ref.lik))
        rel = np.where(den > 0, np.abs(lik - ref.lik) / np.where(den > 0, den, 1), 0)
        # different summation order, same 64-bit-mantissa arithmetic
        assert float(rel.max()) < 1e-13, (reg, float(rel.max()))
        assert ((lik == 0) == (ref.lik == 0)).all()


def test_oracle_viterbi_path_is_the_most_probable_one():
    """The oracle's Viterbi (the reference's loop, src/hmm.cpp:408-511, and its four-candidate form) against an
    ENUMERATION of all (H^2)^C sequences of path pairs on tiny panels, in exact rational arithmetic: the probability of
    the reported path (product of transition and emission probabilities along it) is the maximum over all sequences.
    Ties between sequences are common (the two orders of a pair), so probabilities are compared, not the paths."""
    from fractions import Fraction
    from itertools import product

    def frac(x):
        n, d = np.longdouble(x).as_integer_ratio()
        return Fraction(int(n), int(d))
    table = orc.OracleTable(*default_table_args())
    checked = 0
    for (V, H, seed, recomb, effn) in ((6, 2, 1, 1.26, 25000.0), (5, 3, 2, 446.287102628, 0.25), (6, 2, 3, 1.26, 1e-5), (4, 4, 4, 446.287102628, 0.25)):
        b = synthetic_panel(V, H, 6, seed=900 + seed, undefined_frac=0.0)
        pa = b.path_allele.reshape(V, H)
        for v in range(V):
            pa[v, v % H] = 1  # every variant is a column
        b._c = None
        prm = orc.make_params(recomb, False, effn, run_genotyping=False, run_phasing=True)
        for form in (0, 1):
            r = orc.viterbi_contig(b, table, prm, form=form)
            cols = [v for v in range(V) if r.kept[v]]
            assert len(cols) == V
            n = H * H
            em = []
            for v in cols:
                E, _ = orc.emission_table(b, table, v)
                ids = [int(x) for x in b.allele_id[b.allele_off[v]:b.allele_off[v + 1]]]
                s = [ids.index(int(a)) for a in pa[v]]
                em.append([frac(E[s[i], s[j]]) for i in range(H) for j in range(H)])
            tr = [None] + [[frac(x) for x in orc.transition_probs(int(b.variant_pos[cols[c - 1]]), int(b.variant_pos[cols[c]]), recomb, H, False, effn)] for c in range(1, len(cols))]

            def prob(seq):
                p = em[0][seq[0]]
                for c in range(1, len(seq)):
                    a, bb = seq[c - 1], seq[c]
                    p *= tr[c][(a // H != bb // H) + (a % H != bb % H)] * em[c][bb]
                return p
            best = max(prob(seq) for seq in product(range(n), repeat=len(cols)))
            # the reported path: any state sequence that spells the reported alleles AND is a path of maximal probability
            # need not be unique, so take the best sequence among those that spell the oracle's haplotype alleles
            spelled = [[s for s in range(n) if pa[v, s // H] == r.hap1[v] and pa[v, s % H] == r.hap2[v]] for v in cols]
            got = max(prob(seq) for seq in product(*spelled))
            assert got == best and best > 0, (V, H, form, float(got), float(best))
            checked += 1
    assert checked == 8
