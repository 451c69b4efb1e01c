"""Wide columns at 64 paths inside FUSED jobs (round 6, DevContig::widef; VERDICT r5 item 3: "a wide column costs the column, not the
job" carried from k_sweep_small16x to the 64-path chains of the general kernel).  A 64-path chain whose objects include some with
more than PG_AMAX = 5 alleles on the selected paths used to make its whole job chunked (k_post was the only wide path).  Now the
job stays fused: the phase-2 role of a wide column stores its own column in the variant's aux slot instead of forming posterior
partials (pg_kernels.hip: forward_body / backward_body, widef_col / store_aux) and k_bins_wide forms the bins from that column and
the stored partner (post_ab — what k_post does for every column of a chunked job).  PG_KERNELS=nowidef: the chunked job, as before.
Reference: src/hmm.cpp:275-405 (posterior by allele pair), src/emissionprobabilitycomputer.cpp:9-29."""
import numpy as np
import pytest

from pangenie_amd import hmm
from pangenie_amd.panel import synthetic_panel, synthetic_sample_counts
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


def _panels(reg, H):
    sizes = [330, 1, 2, 3, 97, 64, 65, 200]
    batches = []
    for i, v in enumerate(sizes):
        # wide columns first, last, around the meeting point of the two directions, and scattered
        at = (0, 1, v // 2 - 2, v // 2 - 1, v // 2, v // 2 + 1, v // 2 + 2, v - 2, v - 1) if v > 10 else tuple(range(v))
        batches.append(synthetic_panel(v, H, 20, seed=1900 + i, multiallelic_frac=0.2 if v > 3 else 0.0, undefined_frac=0.03,
                                       wide_frac=0.05 if v > 10 else 0.0, wide_at=at))
    batches.insert(3, synthetic_panel(150, 64, 20, seed=1950))                           # a lean chain (triangle storage, k_sweep_lean2)
    batches.insert(6, synthetic_panel(120, 64, 20, seed=1951, multiallelic_frac=0.3))    # narrow multiallelic columns only: k_sweep_leanx2
    if reg == 0.0:
        for b in batches:
            if b.n_variants > 3:
                b.kmer_count[::3] = 0
                b.kmer_count[1::17] = 60000
    return batches


@pytest.mark.parametrize("H", [64, 41])
@pytest.mark.parametrize("reg", [0.01, 0.0])
def test_wide_columns_at_64_paths_cost_the_column_not_the_job(reg, H, orc, monkeypatch):
    """Eight chains of 1 ... 330 columns with wide columns (6-12 alleles on the paths) at the ends, at the meeting point and
    scattered, a single-column chain whose column is wide, 41 paths padded to 64 as well, beside a lean chain and a chain of narrow
    multiallelic columns; regularised and unregularised tables (fall-back columns in both halves).  The job stays fused; against the
    oracle and against the chunked job (PG_KERNELS=nowidef) to fp64 rounding; a second run gives the same bits."""
    monkeypatch.setenv("PG_SWEEP_MODE", "fused")
    batches = _panels(reg, H)
    nl = [len(set(r)) for b in batches for r in b.path_allele.reshape(b.n_variants, b.n_paths)]
    assert max(nl) > 5   # (there ARE wide columns)
    args = (6, 108, 54, reg)
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    job = hmm.Job(batches, t, p)
    assert job.sweep_mode()[0] == "fused", job.plan()
    assert "wide columns to their aux slots" in job.plan() and "k_bins_wide" in job.plan(), job.plan()
    job.run()
    got = job.fetch_all()
    job.run()
    again = job.fetch_all()
    job.close()
    monkeypatch.setenv("PG_KERNELS", "nowidef")
    job = hmm.Job(batches, t, p)
    assert job.sweep_mode()[0] == "chunked", job.plan()
    job.run()
    chunked = job.fetch_all()
    job.close()
    monkeypatch.delenv("PG_KERNELS", raising=False)
    for b, r, r2, g in zip(batches, got, again, chunked):
        assert np.array_equal(r.lik, r2.lik) and np.array_equal(r.lik_exp, r2.lik_exp)
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, r, ref)
        assert_parity(b, g, ref)
        _agree(r, g, 1e-10)


def test_wide_columns_at_64_paths_in_a_cohort_job(orc, monkeypatch):
    """The same behind the cohort boundary (pg_cohort_new): three samples over an index of two 64-path contigs with wide columns;
    every (sample, contig) chain against the oracle on that sample's counts."""
    monkeypatch.setenv("PG_SWEEP_MODE", "fused")
    args = (6, 108, 54, 0.01)
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    index = [synthetic_panel(257, 64, 20, seed=1970, multiallelic_frac=0.2, wide_frac=0.04, wide_at=(0, 128, 256)),
             synthetic_panel(90, 64, 20, seed=1971, multiallelic_frac=0.3, wide_frac=0.1)]
    samples = []
    for s in range(3):
        kcs, covs = zip(*[synthetic_sample_counts(ix, seed=1980 + 10 * s + c) for c, ix in enumerate(index)])
        samples.append((list(kcs), list(covs)))
    job = hmm.Job.cohort(index, samples, t, p)
    assert job.sweep_mode()[0] == "fused" and "k_bins_wide" in job.plan(), job.plan()
    job.run()
    for s in range(3):
        for c, ix in enumerate(index):
            b = ix.with_counts(samples[s][0][c], samples[s][1][c])
            assert_parity(b, job.fetch(s * len(index) + c), orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5)))
    job.close()
