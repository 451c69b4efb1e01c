"""The split path (pangenie_amd/csrc/pg_split.h, round 6): what the INDEX alone decides — kept columns, present alleles, the
column list, path -> local allele, transition constants (reference src/columnindexer.cpp:8-33,
src/transitionprobabilitycomputer.cpp:8-19) — is formed once per uploaded index and shared by every sample chain over it
(src/commands.cpp:118-138: one index, only kmer counts / coverage differ per sample); a run forms only the emissions
(src/emissionprobabilitycomputer.cpp:9-53) and the bins (src/hmm.cpp:364-368).  Taken by the 16-path chains of fused jobs;
PG_KERNELS=nosplit keeps the per-sample preparation (k_prep*, k_records, k_bins_lean2 / _x) beside it."""
import numpy as np
import pytest

from pangenie_amd import hmm
from pangenie_amd.panel import default_table_args, synthetic_panel, synthetic_sample_counts
from tests.parity_util import assert_parity

pytestmark = pytest.mark.gpu
LD = np.longdouble


@pytest.fixture(scope="module")
def orc():
    from oracle import pyoracle
    return pyoracle


def _agree(a, c, tol):
    a, c = a.likelihoods_ld(), c.likelihoods_ld()
    den = np.maximum(np.abs(a), np.abs(c))
    worst = float(np.where(den > 0, np.abs(a - c) / np.where(den > 0, den, 1), 0).max()) if a.size else 0.0
    assert worst < tol, worst


def _run(batches, t, p, monkeypatch, kernels):
    monkeypatch.setenv("PG_KERNELS", kernels)
    job = hmm.Job(batches, t, p)
    assert job.sweep_mode()[0] == "fused"
    job.run()
    first = job.fetch_all()
    job.run()   # (a second run over the same resident job: the index-level arrays are formed once)
    again = job.fetch_all()
    job.close()
    monkeypatch.delenv("PG_KERNELS", raising=False)
    for a, b in zip(first, again):
        assert np.array_equal(a.lik, b.lik) and np.array_equal(a.lik_exp, b.lik_exp)
    return first


def _panels(reg):
    out = []
    # all-biallelic chains (split 1), chains with 3-5 allele objects (split 2), wide objects, undefined alleles, objects with more
    # than 32 / 64 k-mers (k_prep_s_w), chains of 0 .. 3 columns
    out.append(synthetic_panel(300, 16, 20, seed=3101, undefined_frac=0.05))
    out.append(synthetic_panel(257, 16, 40, seed=3102, undefined_frac=0.02))                       # K = 40 > 32: every object on k_prep_s_w
    out.append(synthetic_panel(310, 16, 20, seed=3103, multiallelic_frac=0.3, undefined_frac=0.05))
    out.append(synthetic_panel(200, 16, 70, seed=3104, multiallelic_frac=0.3))                     # K = 70 > 64: multiallelic objects on k_prep_s_w too
    out.append(synthetic_panel(280, 16, 20, seed=3105, multiallelic_frac=0.2, wide_frac=0.05, wide_at=(0, 139, 140, 279), undefined_frac=0.05))
    for i, v in enumerate((1, 2, 3)):
        out.append(synthetic_panel(v, 16, 20, seed=3110 + i))
        out.append(synthetic_panel(v, 16, 20, seed=3120 + i, multiallelic_frac=1.0))
    out.append(synthetic_panel(120, 64, 20, seed=3130))                                            # (another width in the same job: not split)
    if reg == 0.0:
        for b in out:
            if b.n_variants > 3:
                b.kmer_count[::3] = 0
                b.kmer_count[1::17] = 60000
    return out


@pytest.mark.parametrize("reg", [0.01, 0.0])
def test_split_path_vs_oracle_and_per_sample_preparation(reg, orc, monkeypatch):
    monkeypatch.setenv("PG_SWEEP_MODE", "fused")
    batches = _panels(reg)
    args = (6, 108, 54, reg)
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    got = _run(batches, t, p, monkeypatch, "small")
    old = _run(batches, t, p, monkeypatch, "small,nosplit")
    for b, r, o in zip(batches, got, old):
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, r, ref)
        assert_parity(b, o, ref)
        _agree(r, o, 1e-10)


def test_split_single_column_chains_incl_a_wide_one(orc, monkeypatch):
    """A chain left with ONE column needs no sweep: its bins are (ordered path pairs of the genotype) x E(genotype), or 1 / H^2 per
    state if that column sums to zero (reference src/hmm.cpp:228-267, 356-368) — also when that column is wide (round 5 refused
    such a chain in a fused job)."""
    monkeypatch.setenv("PG_SWEEP_MODE", "fused")
    batches = [synthetic_panel(1, 16, 20, seed=77, wide_at=(0,), wide_alleles=(12, 12)),
               synthetic_panel(1, 16, 20, seed=78), synthetic_panel(1, 16, 20, seed=79, multiallelic_frac=1.0),
               synthetic_panel(5, 16, 20, seed=80, multiallelic_frac=0.5)]
    assert len(set(batches[0].path_allele.tolist())) > 5
    # one kept column among five variants: paths carry the alternative allele at variant 2 only
    pa = batches[3].path_allele.reshape(5, 16).copy()
    pa[[0, 1, 3, 4], :] = 0
    if not pa[2].any():
        pa[2, 3] = 1
    batches[3].path_allele[:] = pa.reshape(-1)
    for reg in (0.01, 0.0):
        bs = [b.with_counts(b.kmer_count.copy(), b.coverage.copy()) for b in batches]
        if reg == 0.0:
            for b in bs:
                b.kmer_count[:] = 60000   # every product underflows to zero: the uniform fall-back of a single column
        args = (6, 108, 54, reg)
        t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
        got = _run(bs, t, p, monkeypatch, "small")
        for b, r in zip(bs, got):
            ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
            assert r.n_columns == 1
            assert_parity(b, r, ref)


def test_split_two_samples_over_one_index_are_two_single_sample_jobs_bit_for_bit(orc, monkeypatch):
    """two samples with different counts over one index, sharing the index-level arrays, give bit for bit
    what two single-sample jobs give — and both match the oracle."""
    monkeypatch.setenv("PG_SWEEP_MODE", "fused")
    monkeypatch.setenv("PG_KERNELS", "small")
    index = [synthetic_panel(v, 16, 20, seed=4200 + i, multiallelic_frac=m, wide_frac=w, undefined_frac=0.03)
             for i, (v, m, w) in enumerate(((260, 0.0, 0.0), (190, 0.25, 0.03), (1, 0.0, 0.0)))]
    samples = []
    for s in range(2):
        kcs, covs = zip(*[synthetic_sample_counts(ix, seed=7000 + 10 * s + i) for i, ix in enumerate(index)])
        samples.append((list(kcs), list(covs)))
    args = default_table_args()
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    job = hmm.Job.cohort(index, samples, t, p)
    job.run()
    both = job.fetch_all()
    job.close()
    for s in range(2):
        one = hmm.Job.cohort(index, [samples[s]], t, p)
        one.run()
        alone = one.fetch_all()
        one.close()
        for i, ix in enumerate(index):
            r = both[s * len(index) + i]
            assert np.array_equal(r.lik, alone[i].lik) and np.array_equal(r.lik_exp, alone[i].lik_exp)
            assert np.array_equal(r.kept, alone[i].kept) and r.n_columns == alone[i].n_columns
            b = ix.with_counts(samples[s][0][i], samples[s][1][i])
            assert_parity(b, r, orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5)))


def test_split_next_samples_and_a_new_index_on_a_resident_job(orc, monkeypatch):
    """pg_job_upload: new counts on the resident index (no index pass), then a new index of the same shape with other path
    alleles — other kept columns — on the resident job (the index pass runs again): both match the oracle."""
    monkeypatch.setenv("PG_SWEEP_MODE", "fused")
    monkeypatch.setenv("PG_KERNELS", "small")
    index = [synthetic_panel(240, 16, 20, seed=4300, multiallelic_frac=0.2, wide_frac=0.03), synthetic_panel(150, 16, 20, seed=4301)]
    def counts(seed):
        kcs, covs = zip(*[synthetic_sample_counts(ix, seed=seed + i) for i, ix in enumerate(index)])
        return (list(kcs), list(covs))
    args = default_table_args()
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    s0, s1 = [counts(100), counts(200)], [counts(300), counts(400)]
    job = hmm.Job.cohort(index, s0, t, p)
    job.run()
    job.upload(s1)
    job.run()
    got = job.fetch_all()
    job.close()
    for s in range(2):
        for i, ix in enumerate(index):
            b = ix.with_counts(s1[s][0][i], s1[s][1][i])
            assert_parity(b, got[s * 2 + i], orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5)))
    # a plain job, then the same shapes with other path alleles (the arrays are changed in place and uploaded again)
    a = [synthetic_panel(200, 16, 20, seed=4310), synthetic_panel(90, 16, 20, seed=4311, multiallelic_frac=0.3)]
    refs = [orc.genotype_contig(x, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5)) for x in a]
    job = hmm.Job(a, t, p)
    job.run()
    first = job.fetch_all()
    for x, r, ref in zip(a, first, refs):
        assert_parity(x, r, ref)
    for x in a:
        pa = x.path_allele.reshape(x.n_variants, 16)
        pa[::3, :] = 0            # a third of the variants lose every alternative allele: no column there any more
    job.upload()
    job.run()
    second = job.fetch_all()
    job.close()
    for x, r, r0 in zip(a, second, first):
        assert r.n_columns < r0.n_columns
        assert_parity(x, r, orc.genotype_contig(x, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5)))


def test_split_entries_below_the_normal_range_keep_their_precise_products(orc, monkeypatch):
    """A column whose present pairs lie more than 2^1021 apart: the scaled entry of the small one is subnormal or zero, so the
    bins kernel cannot rebuild its (mantissa, exponent) product from it — such a column is flagged and its products go to the
    side array (pg_device.h: PG_SREC_FLAG_PRECISE).  The bin of that genotype still carries its full relative precision."""
    monkeypatch.setenv("PG_SWEEP_MODE", "fused")
    args = (6, 108, 54, 0.0)
    for multi in (0.0, 0.4):
        b = synthetic_panel(120, 16, 20, seed=4400 + int(multi * 10), multiallelic_frac=multi, undefined_frac=0.0, zero_kmer_frac=0.0)
        t = hmm.ProbabilityTable(*args)
        ot = orc.OracleTable(*args)
        # copy number 0 of a k-mer with 30 .. 53 reads: 1e-120 instead of the geometric tail — ten such k-mers put a pair 2^-3900 below
        for cov in range(6, 108):
            for cnt in range(30, 54):
                old = t.get(cov, cnt)
                t.modify(cov, cnt, LD("1e-120"), old[1], old[2])
                ot.modify(cov, cnt, LD("1e-120"), old[1], old[2])
        p = hmm.make_params(1.26, False, 1e-5)
        got = _run([b], t, p, monkeypatch, "small")[0]
        old_path = _run([b], t, p, monkeypatch, "small,nosplit")[0]
        ref = orc.genotype_contig(b, ot, orc.make_params(1.26, False, 1e-5))
        lik = ref.lik[ref.lik > 0]
        assert lik.size and float(np.log2(lik.max()) - np.log2(lik.min())) > 1100   # (the test does reach below the normal range)
        assert_parity(b, got, ref)
        assert_parity(b, old_path, ref)


@pytest.mark.parametrize("phase1", ["leanx_tri", "tri1"])
@pytest.mark.parametrize("V", [330, 331, 1, 2])
def test_triangle_storage_of_64_path_chains_with_multiallelic_objects(V, phase1, orc, monkeypatch):
    """Round 6: a 64-path chain with 3-5-allele objects in a fused job stores its (symmetric) columns as upper
    triangles too — phase 1 writes k_sweep_lean_tri's layout (k_sweep_leanx_tri, or the general kernel: "tri1"), phase 2 reads it on
    k_sweep_leanx2 (the lean-2 design with table emissions; partials added up over the waves, k_bins_q) or, PG_KERNELS=noleanx2,
    through the general kernel's triangle ring — instead of leaving triangle storage the moment one object is not biallelic.  Unregularised table as
    well (fall-backs in both halves, re-formed bins); PG_KERNELS=notri (full columns) must agree to fp64 rounding."""
    monkeypatch.setenv("PG_SWEEP_MODE", "fused")
    for seed, reg, multi in ((515, 0.0, 0.3), (516, 0.01, 0.2), (517, 0.0, 1.0)):
        args = (6, 108, 54, reg)
        b = synthetic_panel(V, 64, 20, seed=seed, multiallelic_frac=multi, undefined_frac=0.03)
        if reg == 0.0 and V > 3:
            b.kmer_count[::3] = 0
            b.kmer_count[1::17] = 60000
        t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
        # phase 1 on the lean-x step with triangle stores (k_sweep_leanx_tri, the default) or on the general kernel with them (k_sweep_tri1)
        if phase1 == "tri1":
            monkeypatch.setenv("PG_KERNELS", "noleanx")
        else:
            monkeypatch.delenv("PG_KERNELS", raising=False)
        job = hmm.Job([b, synthetic_panel(100, 64, 20, seed=seed + 50)], t, p)   # (a lean chain in the same job: k_sweep_lean_tri / _lean2)
        assert job.sweep_mode()[0] == "fused"
        if int(np.diff(b.allele_off.astype(np.int64)).max()) > 2:
            assert job.triangle_chains() == 2
        assert ("k_sweep_leanx2" in job.plan() and "k_bins_q" in job.plan()) == (int(np.diff(b.allele_off.astype(np.int64)).max()) > 2), job.plan()
        job.run()
        tri = job.fetch(0)
        job.close()
        # phase 2 on the general kernel's triangle ring + k_bins (PG_KERNELS=noleanx2: round 6's first step) instead of k_sweep_leanx2 + k_bins_q
        monkeypatch.setenv("PG_KERNELS", "noleanx2" + (",noleanx" if phase1 == "tri1" else ""))
        job = hmm.Job([b, synthetic_panel(100, 64, 20, seed=seed + 50)], t, p)
        assert "k_sweep_leanx2" not in job.plan(), job.plan()
        job.run()
        ring = job.fetch(0)
        job.close()
        monkeypatch.setenv("PG_KERNELS", "notri")
        full = hmm.genotype_contig(b, t, p)
        monkeypatch.delenv("PG_KERNELS", raising=False)
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, tri, ref)
        assert_parity(b, ring, ref)
        assert_parity(b, full, ref)
        _agree(tri, full, 1e-11)
        _agree(tri, ring, 1e-11)


def test_leanx2_cohort_chains_of_awkward_lengths(orc, monkeypatch):
    """k_sweep_leanx2 / k_bins_q behind the cohort boundary (pg_cohort_new: SURVEY §8(f)-1): three samples over an index of 64-path
    contigs with 3-5-allele objects whose column counts sit on and beside the kernel's blocks — records in blocks of sixteen, bins
    sixteen columns per workgroup and four per wave, a three-step rotation of the partner buffers — incl. chains of one, two and
    three columns.  Every (sample, contig) chain against the oracle on that sample's counts, and against the general kernel's
    triangle ring (PG_KERNELS=noleanx2) to fp64 rounding; a second run of the resident job gives the same bits."""
    from pangenie_amd.panel import synthetic_sample_counts
    monkeypatch.setenv("PG_SWEEP_MODE", "fused")
    args = default_table_args()
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    lengths = (1, 2, 3, 4, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 95, 96, 97, 129, 257)
    index = [synthetic_panel(V, 64, 20, seed=7000 + V, multiallelic_frac=0.6, undefined_frac=0.02) for V in lengths]
    samples = []
    for s in range(3):
        kcs, covs = zip(*[synthetic_sample_counts(ix, seed=7100 + 100 * s + c) for c, ix in enumerate(index)])
        samples.append((list(kcs), list(covs)))
    job = hmm.Job.cohort(index, samples, t, p)
    assert job.sweep_mode()[0] == "fused"
    assert "k_sweep_leanx2" in job.plan() and "k_bins_q" in job.plan(), job.plan()
    job.run()
    first = job.fetch_all()
    job.run()
    again = job.fetch_all()
    job.close()
    monkeypatch.setenv("PG_KERNELS", "noleanx2")
    job = hmm.Job.cohort(index, samples, t, p)
    assert "k_sweep_leanx2" not in job.plan(), job.plan()
    job.run()
    ring = job.fetch_all()
    job.close()
    monkeypatch.delenv("PG_KERNELS", raising=False)
    for s in range(3):
        for c, ix in enumerate(index):
            i = s * len(index) + c
            assert np.array_equal(first[i].lik, again[i].lik) and np.array_equal(first[i].lik_exp, again[i].lik_exp)
            b = ix.with_counts(samples[s][0][c], samples[s][1][c])
            assert_parity(b, first[i], orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5)))
            _agree(first[i], ring[i], 1e-11)
