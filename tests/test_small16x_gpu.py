"""k_sweep_small16x (pangenie_amd/csrc/pg_small16x.h): 16-path chains with multiallelic objects — the default production
shape, 15 sampled paths + the reference path (reference src/commands.cpp:799-803, src/haplotypesampler.cpp:43,296-309,
src/multiallelicuniquekmers.cpp:195-232, src/emissionprobabilitycomputer.cpp:9-53) — against the oracle and the general
kernel: narrow multiallelic columns by table, wide columns per column (fused jobs stay fused), both sweep modes."""
import numpy as np
import pytest

from pangenie_amd import hmm
from pangenie_amd.panel import default_table_args, synthetic_panel, synthetic_sample_counts
from tests.parity_util import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import pyoracle
    return pyoracle


def _agree(a, c, tol):
    a, c = a.likelihoods_ld(), c.likelihoods_ld()
    den = np.maximum(np.abs(a), np.abs(c))
    worst = float(np.where(den > 0, np.abs(a - c) / np.where(den > 0, den, 1), 0).max()) if a.size else 0.0
    assert worst < tol, worst


def _unregularise(b):
    b.kmer_count[::3] = 0
    b.kmer_count[1::17] = 60000


@pytest.mark.parametrize("mode,K", [("chunked", 1), ("chunked", 2), ("chunked", 7), ("chunked", 64), ("chunked", 4096), ("fused", 0)])
def test_small16x_kernel_multiallelic_h16_vs_oracle_and_general(mode, K, orc, monkeypatch):
    """A fifth / almost half of the objects with 3-5 alleles, undefined alleles, regularised and unregularised table (forward
    columns that fall back to uniform, all-zero backward columns — on, before and behind chunk boundaries, in narrow
    multiallelic columns too): store-only phases (chunked) and both phases (fused) on k_sweep_small16x; the general kernel
    (PG_KERNELS=nosmall) and the oracle must agree with it."""
    monkeypatch.setenv("PG_SWEEP_MODE", mode)
    if K:
        monkeypatch.setenv("PG_CHUNK_COLS", str(K))
    for seed, reg, V, multi, undef in ((21, 0.0, 330, 0.2, 0.05), (22, 0.01, 330, 0.45, 0.05), (23, 0.0, 131, 0.45, 0.01), (24, 0.01, 3, 0.45, 0.0), (25, 0.01, 2, 1.0, 0.0)):
        args = (6, 108, 54, reg)
        b = synthetic_panel(V, 16, 20, seed=seed, multiallelic_frac=multi, undefined_frac=undef)
        if reg == 0.0:
            _unregularise(b)
        t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
        monkeypatch.setenv("PG_KERNELS", "small")   # (by default only jobs with hundreds of such chains take this kernel)
        small = hmm.genotype_contig(b, t, p)
        monkeypatch.setenv("PG_KERNELS", "nosmall")
        gen = hmm.genotype_contig(b, t, p)
        monkeypatch.delenv("PG_KERNELS", raising=False)
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, small, ref)
        assert_parity(b, gen, ref)
        _agree(small, gen, 1e-10)


def _wide_panels(reg):
    sizes = [640, 3, 260, 1, 2, 511, 64, 65, 33, 400, 129, 7, 300]
    batches = []
    for i, v in enumerate(sizes):
        at = (0, 1, v // 2 - 2, v // 2 - 1, v // 2, v // 2 + 1, v // 2 + 2, v - 2, v - 1) if v > 10 else ()
        batches.append(synthetic_panel(v, 16, 20, seed=900 + i, multiallelic_frac=0.2 if v > 3 else 0.0, undefined_frac=0.05,
                                       wide_frac=0.05 if v > 10 else 0.0, wide_at=at))
    batches.insert(5, synthetic_panel(150, 16, 20, seed=950))                           # an all-biallelic chain: k_sweep_small16
    batches.insert(9, synthetic_panel(120, 64, 20, seed=951, multiallelic_frac=0.2))    # and one of another width
    if reg == 0.0:
        for b in batches:
            if b.n_variants > 3:
                b.kmer_count[::3] = 0
    return batches


@pytest.mark.parametrize("reg", [0.01, 0.0])
@pytest.mark.parametrize("mode", ["fused", "chunked"])
def test_small16x_wide_columns_cost_the_column_not_the_job(mode, reg, orc, monkeypatch):
    """Objects of 6-12 alleles of which the sixteen paths carry up to nine — wide columns first, last, around the meeting
    point of the two directions and scattered (5 %) — next to narrow multiallelic and biallelic columns, thirteen chains of
    1 ... 640 columns sharing waves.  A fused job STAYS fused (one such object used to make every chain of the
    job chunked): the wide column's emissions come from the side table inside the sweep, its phase-2 column goes to the aux
    slot and k_bins_wide forms the bins from the two stored columns.  Against the oracle, and against the general kernel in
    the chunked mode (k_post: the path every wide column took before)."""
    monkeypatch.setenv("PG_SWEEP_MODE", mode)
    monkeypatch.setenv("PG_CHUNK_COLS", "97")
    batches = _wide_panels(reg)
    nl = [len(set(r)) for b in batches if b.n_paths == 16 for r in b.path_allele.reshape(b.n_variants, 16)]
    assert max(nl) > 5   # (there ARE wide columns)
    args = (6, 108, 54, reg)
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    monkeypatch.setenv("PG_KERNELS", "small")
    job = hmm.Job(batches, t, p)
    assert job.sweep_mode()[0] == mode
    job.run()
    got = job.fetch_all()
    job.run()
    again = job.fetch_all()
    job.close()
    monkeypatch.setenv("PG_KERNELS", "nosmall")
    monkeypatch.setenv("PG_SWEEP_MODE", "chunked")
    job = hmm.Job(batches, t, p)
    job.run()
    gen = job.fetch_all()
    job.close()
    monkeypatch.delenv("PG_KERNELS", raising=False)
    for b, r, r2, g in zip(batches, got, again, gen):
        assert np.array_equal(r.lik, r2.lik) and np.array_equal(r.lik_exp, r2.lik_exp)
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, r, ref)
        assert_parity(b, g, ref)
        _agree(r, g, 1e-10)


def test_small16x_single_wide_column_chain(orc, monkeypatch):
    """A chain whose ONLY column is wide.  Round 6: on the split path (the default for such chains, pg_split.h) a single column
    needs no sweep and k_bins_wide_s forms its bins — fused or chunked, the chain matches the oracle.  With the per-sample
    preparation (PG_KERNELS=nosplit) phase 2 of a one-column chain runs on the general kernel, which has no wide path inside a
    fused job: PG_ERR_UNSUPPORTED, never a wrong bin."""
    b = synthetic_panel(1, 16, 20, seed=77, wide_at=(0,), wide_alleles=(12, 12))
    assert len(set(b.path_allele.tolist())) > 5
    args = default_table_args()
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    monkeypatch.setenv("PG_KERNELS", "small")
    monkeypatch.setenv("PG_SWEEP_MODE", "chunked")
    assert_parity(b, hmm.genotype_contig(b, t, p), ref)
    monkeypatch.setenv("PG_SWEEP_MODE", "fused")
    assert_parity(b, hmm.genotype_contig(b, t, p), ref)
    monkeypatch.setenv("PG_KERNELS", "small,nosplit")
    with pytest.raises(hmm.PanGenieError):
        hmm.genotype_contig(b, t, p)


@pytest.mark.parametrize("mode", ["fused", "chunked"])
def test_small16x_cohort_rows_of_a_wave_share_their_contig(mode, orc, monkeypatch):
    """A cohort over a shared index (pg_cohort_new): the chain ids of k_sweep_small16x are ordered by index contig, so the four
    half-chains of a wave are four samples of ONE contig.  Six samples x three contigs (18 chains: a partial last wave, waves
    that straddle two contigs), multiallelic and wide objects: every (sample, contig) matches the oracle on that sample's
    counts."""
    monkeypatch.setenv("PG_SWEEP_MODE", mode)
    monkeypatch.setenv("PG_KERNELS", "small")
    index = [synthetic_panel(v, 16, 20, seed=1200 + i, multiallelic_frac=0.25, wide_frac=0.03, undefined_frac=0.03) for i, v in enumerate((210, 97, 333))]
    samples = []
    for s in range(6):
        kcs, covs = zip(*[synthetic_sample_counts(ix, seed=5000 + 10 * s + i) for i, ix in enumerate(index)])
        samples.append((list(kcs), list(covs)))
    args = default_table_args()
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    job = hmm.Job.cohort(index, samples, t, p)
    assert job.sweep_mode()[0] == mode
    job.run()
    got = job.fetch_all()
    job.close()
    for s in range(6):
        for i, ix in enumerate(index):
            b = ix.with_counts(samples[s][0][i], samples[s][1][i])
            ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
            assert_parity(b, got[s * 3 + i], ref)


def test_default_kernel_choice_by_chain_count(orc, monkeypatch):
    """No PG_KERNELS override: the planner's own choice (pg_shim.cpp: the small kernels — and with them the split path — from 256
    sixteen-path chains on, from 320 when some have multiallelic objects; the general kernel below; measured crossover,
    profiles/r06_small_crossover.txt).  Both sides of each threshold are built, the plans say which kernels they got, and a few
    chains of each job are held to the oracle."""
    monkeypatch.delenv("PG_KERNELS", raising=False)
    monkeypatch.delenv("PG_SWEEP_MODE", raising=False)
    args = default_table_args()
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    for multi, below, above, kernel in ((0.0, 31, 32, "k_sweep_small16<2>"), (0.25, 39, 40, "k_sweep_small16x<2>")):
        index = [synthetic_panel(v, 16, 20, seed=1500 + i, multiallelic_frac=multi) for i, v in enumerate((70, 41, 96, 33, 64, 57, 80, 49))]   # (no wide objects: below the threshold they would make the job chunked)
        for S in (below, above):   # 8 S chains
            samples = []
            for s in range(S):
                kcs, covs = zip(*[synthetic_sample_counts(ix, seed=7000 + 10 * (s % 5) + i) for i, ix in enumerate(index)])
                samples.append((list(kcs), list(covs)))
            job = hmm.Job.cohort(index, samples, t, p)
            plan = job.plan()
            assert job.sweep_mode()[0] == "fused", plan
            assert (kernel in plan) == (S == above), plan
            assert ("index pass once (k_index_scan, k_compact, k_index_cols)" in plan) == (S == above), plan   # the split path comes with them
            job.run()
            for s, i in ((0, 0), (S - 1, 7), (3, 2)):
                b = index[i].with_counts(samples[s][0][i], samples[s][1][i])
                assert_parity(b, job.fetch(s * 8 + i), orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5)))
            job.close()
