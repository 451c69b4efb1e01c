"""Viterbi phasing on the device (pangenie_amd/csrc/pg_viterbi.hip, HMM(..., run_phasing=true)) against the CPU oracle's
long double restatement of the reference's loop (oracle/pg_oracle.c:pgo_viterbi_contig; reference src/hmm.cpp:112-173,
:408-511).  Haplotype alleles are integers: the bar is identity.  The reference's own pin is tests/HMMTest.cpp:393-439.
"""
import numpy as np
import pytest

from pangenie_amd import hmm
from pangenie_amd.panel import default_table_args, synthetic_panel
from tests.fixtures_util import build_batch, fill_table

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import pyoracle
    return pyoracle


def tables(spec, orc):
    t = hmm.ProbabilityTable(default=True) if spec["default"] else hmm.ProbabilityTable(*spec["args"])
    o = orc.OracleTable(default=True) if spec["default"] else orc.OracleTable(*spec["args"])
    return fill_table(t, spec, orc.copynumber_regularized), fill_table(o, spec, orc.copynumber_regularized)


def check(batch, res, ref, what):
    assert res.n_columns == ref.n_columns, what
    assert np.array_equal(res.kept, ref.kept), what
    bad = np.flatnonzero((res.haplotype_1 != ref.hap1) | (res.haplotype_2 != ref.hap2))
    assert bad.size == 0, (what, bad[:10], res.haplotype_1[bad[:10]], ref.hap1[bad[:10]], res.haplotype_2[bad[:10]], ref.hap2[bad[:10]])


def test_viterbi_known_answers(golden, orc):
    """every HMM fixture of the reference's tests (zero emissions, underflow, undefined alleles, only_paths, uniform
    transitions, ...): haplotypes, kept columns and the by-column-index meta data as the oracle gives them; the one
    fixture for which the reference pins haplotypes (no_unique_kmers3) against its expected values"""
    for case in golden["hmm"]:
        batch = build_batch(case["variants"], case["hmm"]["only_paths"])
        t, o = tables(case["table"], orc)
        h = case["hmm"]
        for geno in (False, True):
            res = hmm.genotype_contig(batch, t, hmm.make_params(h["recombrate"], h["uniform"], h["effective_N"],
                                                                run_genotyping=geno, run_phasing=True))
            ref = orc.viterbi_contig(batch, o, orc.make_params(h["recombrate"], h["uniform"], h["effective_N"],
                                                               run_genotyping=False, run_phasing=True))
            check(batch, res, ref, (case["name"], geno))
            if not geno:  # (sic: by column index, reference src/hmm.cpp:164-165)
                assert np.array_equal(res.n_kmers, ref.n_kmers) and np.array_equal(res.coverage, ref.coverage), case["name"]
                assert all(g.contains_no_likelihoods() for g in res.genotyping_results()), case["name"]
        if "expected_haplotype1" in case:
            e1, e2 = case["expected_haplotype1"], case["expected_haplotype2"]
            g = res.genotyping_results()
            got1, got2 = [r.haplotype_1 for r in g], [r.haplotype_2 for r in g]
            assert (got1 == e1 and got2 == e2) or (got1 == e2 and got2 == e1), case["name"]


SHAPES = [  # (V, H, multiallelic fraction, max alleles)
    (400, 2, 0.0, 5), (300, 5, 0.3, 5), (500, 12, 0.2, 5), (600, 16, 0.0, 5), (400, 17, 0.3, 5), (700, 30, 0.2, 5),
    (300, 32, 0.0, 5), (250, 33, 0.3, 5), (300, 45, 0.2, 5), (200, 64, 0.2, 5), (250, 24, 0.5, 12),
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "V%d_H%d_m%g_A%d" % s)
def test_viterbi_vs_oracle(shape, orc):
    """seeded panels over every kernel width (16 / 32 / 64 lanes per row of states), with multiallelic and wide
    columns (more than 5 alleles on the selected paths), in four transition regimes: the reference's default, the
    production effective_N, no recombination at all (q == 0: the all-products-zero rule) and uniform transitions"""
    V, H, multi, maxa = shape
    batch = synthetic_panel(V, H, 20, seed=7000 + V + H, multiallelic_frac=multi, max_alleles=maxa, local_alts=8)
    t = hmm.ProbabilityTable(*default_table_args())
    o = orc.OracleTable(*default_table_args())
    form = 0 if H <= 20 else 1  # (form 1 == form 0: tests/test_oracle_golden.py)
    for recomb, effn, uni in ((1.26, 25000.0, False), (1.26, 1e-5, False), (0.0, 25000.0, False), (1.26, 25000.0, True),
                              (446.287102628, 0.25, False)):
        res = hmm.genotype_contig(batch, t, hmm.make_params(recomb, uni, effn, run_genotyping=False, run_phasing=True))
        ref = orc.viterbi_contig(batch, o, orc.make_params(recomb, uni, effn, run_genotyping=False, run_phasing=True), form=form)
        check(batch, res, ref, (shape, recomb, effn, uni))
        assert res.n_columns > 0


def test_viterbi_duplicated_paths_and_equal_positions(orc):
    """exact ties: duplicated paths (equal rows / columns of the state matrix), neighbouring variants at the SAME
    position (distance 0: q == 0 for that gap only) and variants without k-mers (emission 1 everywhere)"""
    V, H = 400, 28
    batch = synthetic_panel(V, H, 20, seed=99, zero_kmer_frac=0.2)
    pa = batch.path_allele.reshape(V, H).copy()
    pa[:, 20:] = pa[:, :8]
    pos = batch.variant_pos.copy()
    pos[50:60] = pos[50]
    pos[200:203] = pos[200]
    from pangenie_amd.panel import ContigBatch
    b2 = ContigBatch(H, pos, batch.coverage, batch.kmer_off, batch.kmer_count, batch.allele_off, batch.allele_id,
                     batch.allele_flags, batch.allele_kmer_off, batch.allele_kmer_mask, pa.reshape(-1))
    t = hmm.ProbabilityTable(*default_table_args())
    o = orc.OracleTable(*default_table_args())
    for recomb, effn in ((1.26, 25000.0), (1.26, 1e-5)):
        res = hmm.genotype_contig(b2, t, hmm.make_params(recomb, False, effn, run_genotyping=False, run_phasing=True))
        ref = orc.viterbi_contig(b2, o, orc.make_params(recomb, False, effn, run_genotyping=False, run_phasing=True), form=1)
        check(b2, res, ref, (recomb, effn))


def test_viterbi_job_of_mixed_chains(orc):
    """one resident job over chains of different widths and lengths (block staging across several 64-column blocks,
    a contig without kept columns), genotyping and phasing together; the likelihoods are those of a
    genotyping-only run"""
    shapes = [(1000, 30), (70, 9), (64, 16), (65, 40), (129, 64), (33, 8), (300, 20)]
    batches = [synthetic_panel(V, H, 20, seed=500 + i, multiallelic_frac=0.2) for i, (V, H) in enumerate(shapes)]
    none = synthetic_panel(40, 6, 20, seed=77)
    none.path_allele[:] = 0  # every path carries the reference allele: no column is kept
    none._c = None
    batches.append(none)
    t = hmm.ProbabilityTable(*default_table_args())
    o = orc.OracleTable(*default_table_args())
    job = hmm.Job(batches, t, hmm.make_params(1.26, False, 25000.0, run_genotyping=True, run_phasing=True))
    plain = hmm.Job(batches, t, hmm.make_params(1.26, False, 25000.0))
    try:
        for _ in range(2):  # (a second run of the resident job gives the same)
            job.run()
            plain.run()
            for i, b in enumerate(batches):
                res = job.fetch(i)
                ref = orc.viterbi_contig(b, o, orc.make_params(1.26, False, 25000.0, run_genotyping=False, run_phasing=True),
                                         form=0 if b.n_paths <= 20 else 1)
                check(b, res, ref, i)
                p = plain.fetch(i)
                assert np.array_equal(res.lik, p.lik) and np.array_equal(res.lik_exp, p.lik_exp), i
        assert job.viterbi_ms() > 0.0
    finally:
        job.close()
        plain.close()


def test_viterbi_long_chain_properties():
    """a chr22-scale chain (200 k variants, 30 paths: what the reference's callers pass, src/commands.cpp:939), too long
    for the oracle: determinism, every haplotype allele is one a selected path carries at that variant, and the
    unordered allele pair of the Viterbi path is the forward-backward run's likeliest genotype at most columns"""
    V, H = 200_000, 30
    batch = synthetic_panel(V, H, 20, seed=4242)
    t = hmm.ProbabilityTable(*default_table_args())
    prm = hmm.make_params(1.26, False, 1e-5, run_genotyping=False, run_phasing=True)
    a = hmm.genotype_contig(batch, t, prm)
    b = hmm.genotype_contig(batch, t, prm)
    assert np.array_equal(a.haplotype_1, b.haplotype_1) and np.array_equal(a.haplotype_2, b.haplotype_2)
    pa = batch.path_allele.reshape(V, H)
    kept = a.kept.astype(bool)
    assert a.n_columns == int(kept.sum()) > V // 2
    on1 = (pa == a.haplotype_1[:, None]).any(axis=1)
    on2 = (pa == a.haplotype_2[:, None]).any(axis=1)
    assert on1[kept].all() and on2[kept].all()
    assert not a.haplotype_1[~kept].any() and not a.haplotype_2[~kept].any()
    g = hmm.genotype_contig(batch, t, hmm.make_params(1.26, False, 1e-5))
    from tests.parity_util import calls
    gt = calls(batch, g.likelihoods_ld())  # (all biallelic: the bin of the pair (lo, hi) is lo + hi)
    pair_bin = a.haplotype_1.astype(np.int64) + a.haplotype_2.astype(np.int64)
    sel = kept & (gt >= 0)
    agree = float(np.mean(pair_bin[sel] == gt[sel]))
    assert agree > 0.9, agree
