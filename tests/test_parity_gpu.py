"""Parity tests proper (need a MI355X): HIP path through the C ABI vs
  (1) the reference's own known-answer vectors (tests/golden/), tolerance 1e-7 absolute
      like the reference's tests (reference tests/utils.cpp:9-11);
  (2) the CPU oracle on seeded synthetic panels: <= 1e-6 relative on every unnormalised
      genotype likelihood, identical genotype calls (BASELINE.json north_star);
  (3) size-independent properties at BASELINE.json's full single-GPU size.
"""
import numpy as np
import pytest

from pangenie_amd import hmm
from pangenie_amd.genotyping_result import normalized_bins
from pangenie_amd.panel import default_table_args, flatten, synthetic_panel
from tests.fixtures_util import build_batch, build_variant, fill_table, triple
from tests.parity_util import assert_parity, calls, log_parity, rel_errors

pytestmark = pytest.mark.gpu
TOL = 1e-7


@pytest.fixture(scope="module")
def orc():
    from oracle import pyoracle
    return pyoracle


def hip_table(spec, orc):
    t = hmm.ProbabilityTable(default=True) if spec["default"] else hmm.ProbabilityTable(*spec["args"])
    return fill_table(t, spec, orc.copynumber_regularized)


def oracle_table(spec, orc):
    t = orc.OracleTable(default=True) if spec["default"] else orc.OracleTable(*spec["args"])
    return fill_table(t, spec, orc.copynumber_regularized)


def run_case(case, orc):
    batch = build_batch(case["variants"], case["hmm"]["only_paths"])
    h = case["hmm"]
    res = hmm.genotype_contig(batch, hip_table(case["table"], orc),
                              hmm.make_params(h["recombrate"], h["uniform"], h["effective_N"]))
    return batch, res


def test_device_visible():
    assert hmm._lib.load_hip().pg_hmm_device_count() >= 1


def test_hmm_known_answers(golden, orc):
    for case in golden["hmm"]:
        batch, res = run_case(case, orc)
        out = res.genotyping_results()
        if case["hmm"]["normalize"]:
            for g in out:
                g.normalize()
        got = [triple(g) for g in out]
        assert np.allclose(got, case["expected_likelihoods"], rtol=0, atol=TOL), (case["name"], got)
        if "expected_coverage" in case:
            assert [g.coverage() for g in out] == case["expected_coverage"]
            assert [g.nr_unique_kmers() for g in out] == case["expected_n_kmers"]
        if "expected_specific" in case:
            spec = [triple(g.get_specific_likelihoods(d)) for g, d in zip(out, case["defined_alleles"])]
            assert np.allclose(spec, case["expected_specific"], rtol=0, atol=TOL), case["name"]
        if "expected_gt" in case:
            assert [list(g.get_likeliest_genotype()) for g in out] == case["expected_gt"]
        if "expected_after_normalize" in case:
            for g in out:
                g.normalize()
            assert np.allclose([triple(g) for g in out], case["expected_after_normalize"], rtol=0, atol=TOL)
        # and bin-for-bin against the oracle
        h = case["hmm"]
        ref = orc.genotype_contig(batch, oracle_table(case["table"], orc),
                                  orc.make_params(h["recombrate"], h["uniform"], h["effective_N"]))
        assert_parity(batch, res, ref)


def test_hmm_combine(golden, orc):
    by_name = {c["name"]: c for c in golden["hmm"]}
    a = run_case(by_name[golden["combine"]["first"]], orc)[1].genotyping_results()
    b = run_case(by_name[golden["combine"]["second"]], orc)[1].genotyping_results()
    for g in a + b:
        g.normalize()
    expect = [np.add(triple(x), triple(y)) for x, y in zip(a, b)]
    for x, y in zip(a, b):
        x.combine(y)
    assert np.allclose([triple(x) for x in a], expect, rtol=0, atol=TOL)


def test_emission_known_answers(golden, orc):
    for case in golden["emission"]:
        batch = flatten([build_variant(case["variant"])])
        E, _ = hmm.emission_table(batch, hip_table(case["table"], orc), 0)
        ids = list(batch.allele_id)
        for key, val in case["expected"].items():
            a, b = (int(x) for x in key.split(","))
            assert abs(float(E[ids.index(a), ids.index(b)]) - val) < TOL, (case["name"], key)


def test_transition_known_answers(golden):
    for c in golden["transition"]:
        t = hmm.transition_probs(c["from"], c["to"], c["recombrate"], c["nr_paths"], c["uniform"], c["effective_N"])
        q = c["recomb_prob"]
        p = q + c["no_recomb_minus_recomb"]
        assert np.allclose(t, [p * p, p * q, q * q], rtol=0, atol=TOL)
    assert list(hmm.transition_probs(1, 2, 1.26, 5, True, 0.25)) == [1.0, 1.0, 1.0]


def test_emission_vs_oracle_on_the_fly_entries(orc):
    # coverage / counts outside the precomputed box exercise the device's closed form
    b = synthetic_panel(40, 6, 24, seed=11, multiallelic_frac=0.5, undefined_frac=0.3)
    b.kmer_count[::7] = 300
    b.kmer_count[3::11] = 2000
    b.coverage[::5] = 3
    b.coverage[1::9] = 400
    args = default_table_args()
    th, to = hmm.ProbabilityTable(*args), orc.OracleTable(*args)
    for v in range(0, b.n_variants, 3):
        Eh, zh = hmm.emission_table(b, th, v)
        Eo, zo = orc.emission_table(b, to, v)
        assert zh == zo
        den = np.maximum(np.abs(Eh), np.abs(Eo))
        rel = np.where(den > 0, np.abs(Eh - Eo) / np.where(den > 0, den, 1), 0)
        assert float(rel.max()) < 1e-9, (v, float(rel.max()))


PANELS = [
    # V, H, K, multiallelic, kwargs
    (3000, 16, 20, 0.0, {}),
    (1500, 13, 20, 0.0, {}),
    (64, 1, 20, 0.0, {}),
    (700, 2, 12, 0.0, {}),
    (800, 32, 20, 0.0, {}),
    (500, 27, 30, 0.3, {}),
    (600, 64, 20, 0.0, {}),
    (400, 50, 24, 0.3, {}),
    (150, 128, 20, 0.0, {}),
    (120, 100, 20, 0.2, {}),
    (1000, 16, 40, 0.4, {"undefined_frac": 0.2, "zero_kmer_frac": 0.2}),
    (1, 16, 20, 0.0, {}), (2, 16, 20, 0.0, {}), (3, 64, 20, 0.0, {}), (4, 64, 20, 0.0, {}), (5, 128, 20, 0.0, {}),
]


@pytest.mark.parametrize("V,H,K,multi,kw", PANELS)
def test_panels_vs_oracle(V, H, K, multi, kw, orc):
    b = synthetic_panel(V, H, K, seed=1000 + V + H, multiallelic_frac=multi, **kw)
    args = default_table_args()
    res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5))
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    assert_parity(b, res, ref)


@pytest.mark.parametrize("recomb,uniform,N", [(1.26, True, 1e-5), (0.001, False, 1e-5), (1e-9, False, 1e-5),
                                              (446.287102628, False, 0.25), (1.26, False, 25000.0)])
def test_transition_regimes_vs_oracle(recomb, uniform, N, orc):
    b = synthetic_panel(400, 24, 16, seed=77, multiallelic_frac=0.2)
    args = default_table_args()
    res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params(recomb, uniform, N))
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(recomb, uniform, N))
    assert_parity(b, res, ref)


def test_zero_recombination_documented_range(orc):
    """recombrate == 0: no mixing at all, every state evolves on its own (a setting the reference's CLI
    never produces: recombrate is fixed at 1.26, src/pangenie-genotype.cpp:33-45).  The stored fp64
    columns then keep 2^-1400 of a column's sum where the reference's long double keeps 2^-16445
    (include/pangenie_hmm.h, numeric contract): bins within 350 decades of their variant's largest bin
    are held to the usual 1e-6 relative, calls are identical; deeper bins may flush to 0."""
    b = synthetic_panel(400, 24, 16, seed=77, multiallelic_frac=0.2)
    args = default_table_args()
    res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params(0.0, False, 1e-5))
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(0.0, False, 1e-5))
    got = res.likelihoods_ld()
    rel = rel_errors(b, got, ref.lik)
    go = b.geno_off.astype(np.int64)
    mx = np.repeat(np.maximum.reduceat(np.concatenate([ref.lik, np.zeros(1, ref.lik.dtype)]), go[:-1])[:b.n_variants],
                   np.diff(go))
    near = ref.lik > mx * np.longdouble(10.0) ** -350
    assert float(rel[near].max()) < 1e-6
    assert (calls(b, got) == calls(b, ref.lik)).all()


def test_unregularized_table_zero_emissions_vs_oracle(orc):
    # regularization 0: exact zeros appear, forward/backward uniform fallbacks and all_zeros fire
    b = synthetic_panel(600, 16, 20, seed=5)
    b.kmer_count[::3] = 0
    b.kmer_count[1::17] = 60000
    args = (6, 108, 54, 0.0)
    res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5))
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    assert_parity(b, res, ref)


@pytest.mark.parametrize("K", [1, 2, 5, 64])
def test_chunk_boundaries_with_fallback_columns(K, orc, monkeypatch):
    """Chunked sweep mode with tiny chunks: every few columns the recursion is resumed from a
    stored column, on a panel whose unregularised table produces exact zeros (forward columns
    that fall back to uniform, backward columns that are all zero) — so fallbacks land on, before
    and behind chunk boundaries.  Same bar as everywhere: 1e-6 relative, identical calls."""
    monkeypatch.setenv("PG_SWEEP_MODE", "chunked")
    monkeypatch.setenv("PG_CHUNK_COLS", str(K))
    args = (6, 108, 54, 0.0)
    for seed, H, multi in ((5, 16, 0.0), (6, 64, 0.3), (7, 27, 0.2)):
        b = synthetic_panel(160, H, 20, seed=seed, multiallelic_frac=multi)
        b.kmer_count[::3] = 0
        b.kmer_count[1::17] = 60000
        res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5))
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, res, ref)


def test_many_alleles_per_variant_vs_oracle(orc):
    """UniqueKmers objects with up to 32 alleles of which the selected paths carry <= 5: the
    emission products (and the all_zeros rule) run over all A(A+1)/2 pairs, 64 at a time."""
    b = synthetic_panel(300, 16, 128, seed=21, multiallelic_frac=0.6, max_alleles=32, undefined_frac=0.1)
    assert int(np.diff(b.allele_off).max()) > 20
    args = default_table_args()
    res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5))
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    assert_parity(b, res, ref)
    for v in (int(np.argmax(np.diff(b.allele_off))), 7):
        Eh, zh = hmm.emission_table(b, hmm.ProbabilityTable(*args), v)
        Eo, zo = orc.emission_table(b, orc.OracleTable(*args), v)
        den = np.maximum(np.abs(Eh), np.abs(Eo))
        assert zh == zo and float(np.where(den > 0, np.abs(Eh - Eo) / np.where(den > 0, den, 1), 0).max()) < 1e-9


def test_emission_dominated_corner_vs_oracle(orc):
    # many k-mers per allele (K = 32 per allele, multiallelic): emission scale ~1e-150 per column
    b = synthetic_panel(300, 16, 160, seed=9, multiallelic_frac=1.0)
    args = default_table_args()
    res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5))
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    assert_parity(b, res, ref)


def test_no_columns_and_empty(orc):
    b = synthetic_panel(50, 8, 10, seed=2)
    b.path_allele[:] = 0  # every selected path carries the reference allele -> no columns
    args = default_table_args()
    res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params())
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params())
    assert res.n_columns == 0 == ref.n_columns
    assert_parity(b, res, ref)
    assert not res.n_kmers.any() and not res.coverage.any()  # reference src/hmm.cpp:94


@pytest.mark.parametrize("mode,hs", [("fused", [16, 64, 64, 32, 16, 128]), ("chunked", [16, 64, 64, 32, 16, 128])])
def test_multi_contig_job_matches_single_calls(orc, monkeypatch, mode, hs):
    """One resident job over contigs of different length and haplotype count, in both sweep modes.
    Bitwise comparisons are made within one mode: the two modes sum in a different order and
    differ in the last bits."""
    monkeypatch.setenv("PG_SWEEP_MODE", mode)
    monkeypatch.setenv("PG_CHUNK_COLS", "64")  # several chunks even on these small panels
    args = default_table_args()
    t = hmm.ProbabilityTable(*args)
    p = hmm.make_params(1.26, False, 1e-5)
    batches = [synthetic_panel(300 + 50 * i, H, 20, seed=40 + i) for i, H in enumerate(hs)]
    job = hmm.Job(batches, t, p)
    job.run()
    job.run()  # re-running a resident job must give the same answer
    for i, b in enumerate(batches):
        multi = job.fetch(i)
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, multi, ref)
        single = hmm.genotype_contig(b, t, p)
        assert (single.lik == multi.lik).all() and (single.lik_exp == multi.lik_exp).all()  # deterministic
    job.close()


@pytest.mark.parametrize("H,V", [(215, 60), (200, 40), (300, 24), (700, 10)])
def test_many_paths_generic_kernel_vs_oracle(H, V, orc):
    """More than 128 selected paths (HP = 256 / 512 / 1024: the generic sweep kernel).  H = 215 is the
    reference's own integration fixture (all paths in one subset, tests/CommandsTest.cpp:31)."""
    b = synthetic_panel(V, H, 20, seed=500 + H, multiallelic_frac=0.2)
    args = default_table_args()
    res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5))
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    assert_parity(b, res, ref)


def test_fixture_shape_215_paths_44_alleles_vs_oracle(orc):
    """The shape of the reference's region fixture (215 paths, first record 62
    k-mers and 44 alleles, most of them undefined and carried by single paths): wide columns with up to 44
    distinct alleles on the selected paths, on the generic kernel."""
    b = synthetic_panel(40, 215, 62, seed=44, multiallelic_frac=0.5, max_alleles=44, local_alts=43, undefined_frac=0.3)
    pa = b.path_allele.reshape(40, 215)
    assert int(np.diff(b.allele_off).max()) >= 40 and max(len(set(r)) for r in pa) > 32
    for reg in (0.01, 0.0):
        args = (6, 108, 54, reg)
        res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5))
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, res, ref)


@pytest.mark.parametrize("H", [64, 100, 128])
def test_generic_kernel_cross_checks_register_kernels(H, orc, monkeypatch):
    """PG_KERNELS=generic runs HP = 64 / 128 on the independently written generic kernel: both must
    match the oracle, and each other to fp64 rounding."""
    b = synthetic_panel(260, H, 24, seed=70 + H, multiallelic_frac=0.3)
    args = (6, 108, 54, 0.0)
    b.kmer_count[::3] = 0
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    monkeypatch.setenv("PG_SWEEP_MODE", "chunked")
    monkeypatch.setenv("PG_CHUNK_COLS", "37")
    reg = hmm.genotype_contig(b, t, p)
    monkeypatch.setenv("PG_KERNELS", "generic")
    gen = hmm.genotype_contig(b, t, p)
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    assert_parity(b, reg, ref)
    assert_parity(b, gen, ref)
    a, c = reg.likelihoods_ld(), gen.likelihoods_ld()
    den = np.maximum(np.abs(a), np.abs(c))
    assert float(np.where(den > 0, np.abs(a - c) / np.where(den > 0, den, 1), 0).max()) < 1e-11


@pytest.mark.parametrize("K", [1, 2, 7, 64, 4096])
def test_lean_kernel_biallelic_h64_vs_oracle_and_general(K, orc, monkeypatch):
    """All-biallelic H = 64 chains of chunked jobs run their store-only phases on k_sweep_lean (MFMA totals behind the
    LDS exchange; the pipelined formulation measured at par in round 4 lives under tools/lean_pipe/, outside the product).  Unregularised table: forward columns that fall back to uniform and
    all-zero backward columns, on, before and behind chunk and record-block boundaries (330 and 131 columns: blocks of
    64 records).  The lean kernel and the general kernel must match the oracle and agree with each other to fp64
    rounding."""
    monkeypatch.setenv("PG_SWEEP_MODE", "chunked")
    monkeypatch.setenv("PG_CHUNK_COLS", str(K))
    for seed, reg, V in ((5, 0.0, 330), (6, 0.01, 330), (7, 0.0, 131), (8, 0.01, 3)):
        args = (6, 108, 54, reg)
        b = synthetic_panel(V, 64, 20, seed=seed)
        if reg == 0.0:
            b.kmer_count[::3] = 0
            b.kmer_count[1::17] = 60000
        t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
        monkeypatch.delenv("PG_KERNELS", raising=False)
        lean = hmm.genotype_contig(b, t, p)
        monkeypatch.setenv("PG_KERNELS", "general")
        gen = hmm.genotype_contig(b, t, p)
        monkeypatch.delenv("PG_KERNELS", raising=False)
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, lean, ref)
        assert_parity(b, gen, ref)
        a, c = lean.likelihoods_ld(), gen.likelihoods_ld()
        den = np.maximum(np.abs(a), np.abs(c))
        assert float(np.where(den > 0, np.abs(a - c) / np.where(den > 0, den, 1), 0).max()) < 1e-11


@pytest.mark.parametrize("mode,K", [("chunked", 1), ("chunked", 2), ("chunked", 7), ("chunked", 19), ("chunked", 4096), ("fused", 0)])
def test_leanx_kernel_narrow_columns_vs_oracle_and_general(mode, K, orc, monkeypatch):
    """HP = 128 chains — and HP = 64 chains with multiallelic objects — whose objects have at most five alleles run their
    store-only phases on k_sweep_leanx (full records through LDS in blocks of 16, emissions by table lookup, biallelic and
    multiallelic columns on one path).  128 / 100 paths and 64 / 50 paths (phantom rows / columns), 30 % multiallelic, regularised and unregularised table (uniform
    fall-backs and all-zero backward columns on, before and behind chunk and record-block boundaries).  Both kernels
    must match the oracle and agree with each other to fp64 rounding."""
    monkeypatch.setenv("PG_SWEEP_MODE", mode)
    if K:
        monkeypatch.setenv("PG_CHUNK_COLS", str(K))
    for seed, H, reg, V in ((5, 128, 0.0, 150), (6, 128, 0.01, 131), (7, 100, 0.0, 90), (8, 100, 0.01, 77), (9, 64, 0.0, 140), (10, 50, 0.01, 117)):
        args = (6, 108, 54, reg)
        b = synthetic_panel(V, H, 20, seed=seed, multiallelic_frac=0.3)
        if reg == 0.0:
            b.kmer_count[::3] = 0
            b.kmer_count[1::17] = 60000
        t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
        monkeypatch.setenv("PG_KERNELS", "leanx")   # (also in phase 1 of the fused mode, where the general kernel is the default)
        lx = hmm.genotype_contig(b, t, p)
        monkeypatch.setenv("PG_KERNELS", "noleanx")
        gen = hmm.genotype_contig(b, t, p)
        monkeypatch.delenv("PG_KERNELS", raising=False)
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, lx, ref)
        assert_parity(b, gen, ref)
        a, c = lx.likelihoods_ld(), gen.likelihoods_ld()
        den = np.maximum(np.abs(a), np.abs(c))
        assert float(np.where(den > 0, np.abs(a - c) / np.where(den > 0, den, 1), 0).max()) < 1e-11


@pytest.mark.parametrize("K", [1, 2, 7, 64, 4096])
def test_small16_kernel_biallelic_h16_vs_oracle_and_general(K, orc, monkeypatch):
    """All-biallelic H = 16 chains (BASELINE configs[1]) run their store-only phases on k_sweep_small16: four half-chains
    per wave, one per DPP row, nothing exchanged outside the row.  Unregularised table: forward columns that fall back to
    uniform and all-zero backward columns, on, before and behind chunk boundaries.  The small and the general kernel
    must both match the oracle and agree with each other to fp64 rounding."""
    monkeypatch.setenv("PG_SWEEP_MODE", "chunked")
    monkeypatch.setenv("PG_CHUNK_COLS", str(K))
    for seed, reg in ((15, 0.0), (16, 0.01)):
        args = (6, 108, 54, reg)
        b = synthetic_panel(330, 16, 20, seed=seed)
        if reg == 0.0:
            b.kmer_count[::3] = 0
            b.kmer_count[1::17] = 60000
        t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
        monkeypatch.setenv("PG_KERNELS", "small")   # (by default only jobs with hundreds of such chains take this kernel)
        small = hmm.genotype_contig(b, t, p)
        monkeypatch.setenv("PG_KERNELS", "nosmall")
        gen = hmm.genotype_contig(b, t, p)
        monkeypatch.delenv("PG_KERNELS", raising=False)
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, small, ref)
        assert_parity(b, gen, ref)
        a, c = small.likelihoods_ld(), gen.likelihoods_ld()
        den = np.maximum(np.abs(a), np.abs(c))
        assert float(np.where(den > 0, np.abs(a - c) / np.where(den > 0, den, 1), 0).max()) < 1e-11


@pytest.mark.parametrize("H", [16, 32])
def test_class_sums_in_the_single_wave_sweeps_vs_oracle_and_partials(H, orc, monkeypatch):
    """Fused jobs, 16 / 32 paths, every object biallelic (DevContig::cls4): phase 2 writes the four class sums of a column
    and k_bins_lean2 forms the bins (one thread per column), instead of 1 KB of per-thread partials reduced by k_bins.
    Regularised and unregularised table (forward fall-backs whose bins are re-formed from the stored backward column),
    chains of 1 ... 300 variants; both forms must match the oracle and agree with each other to fp64 rounding."""
    monkeypatch.setenv("PG_SWEEP_MODE", "fused")
    for seed, reg, V in ((5, 0.0, 300), (6, 0.01, 257), (7, 0.0, 1), (8, 0.01, 2), (9, 0.0, 3)):
        args = (6, 108, 54, reg)
        b = synthetic_panel(V, H, 20, seed=seed)
        if reg == 0.0 and V > 3:
            b.kmer_count[::3] = 0
            b.kmer_count[1::17] = 60000
        t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
        monkeypatch.delenv("PG_KERNELS", raising=False)
        cls = hmm.genotype_contig(b, t, p)
        monkeypatch.setenv("PG_KERNELS", "nocls4")
        old = hmm.genotype_contig(b, t, p)
        monkeypatch.delenv("PG_KERNELS", raising=False)
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, cls, ref)
        assert_parity(b, old, ref)
        a, c = cls.likelihoods_ld(), old.likelihoods_ld()
        den = np.maximum(np.abs(a), np.abs(c))
        assert float(np.where(den > 0, np.abs(a - c) / np.where(den > 0, den, 1), 0).max()) < 1e-11


@pytest.mark.parametrize("H", [17, 20, 21, 24, 25, 29, 32])
def test_short_columns_of_fused_jobs_vs_oracle_and_full_columns(H, orc, monkeypatch):
    """Fused jobs at 17 ... 32 paths (HP = 32; 17 = a user-chosen panel size of 16 + the reference path — the default is 15 + 1 = 16): phase 1 stores, and the loader of
    phase 2 fetches, only the first DevContig::live = H rounded up to 4 rows and lanes of a column (the rest of the LDS ring is
    zeroed once); the posterior partials of a wave's two halves are added in registers and only real paths' entries are
    written / read.  Regularised and unregularised table (fall-back columns re-formed from the stored backward column, which
    is read below H only), 1 ... 260 variants, multiallelic objects: must match the oracle, and PG_KERNELS=fullcols (whole
    32 x 32 columns) bit for bit — rows and lanes at or above H only ever contribute exact zeros."""
    monkeypatch.setenv("PG_SWEEP_MODE", "fused")
    for seed, reg, V in ((5, 0.0, 260), (6, 0.01, 257), (7, 0.0, 1), (8, 0.01, 2), (9, 0.0, 3), (10, 0.01, 9)):
        args = (6, 108, 54, reg)
        b = synthetic_panel(V, H, 20, seed=seed, multiallelic_frac=0.3, undefined_frac=0.05)
        if reg == 0.0 and V > 3:
            b.kmer_count[::3] = 0
        t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
        monkeypatch.delenv("PG_KERNELS", raising=False)
        short = hmm.genotype_contig(b, t, p)
        monkeypatch.setenv("PG_KERNELS", "fullcols")
        full = hmm.genotype_contig(b, t, p)
        monkeypatch.delenv("PG_KERNELS", raising=False)
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, short, ref)
        assert_parity(b, full, ref)
        assert np.array_equal(short.likelihoods_ld(), full.likelihoods_ld())


@pytest.mark.parametrize("mode", ["chunked", "fused"])
def test_small16_kernel_many_chains_of_different_lengths(mode, orc, monkeypatch):
    """Chains of very different lengths share waves (the trip count is the longest row's; finished rows run on with
    their stores parked): 11 chains — a partial last wave — incl. chains of one and two columns, next to H = 64 and
    multiallelic H = 16 chains that take other kernels; every chain equals its single-call result bit for bit and
    matches the oracle."""
    monkeypatch.setenv("PG_SWEEP_MODE", mode)
    monkeypatch.setenv("PG_CHUNK_COLS", "97")
    monkeypatch.setenv("PG_KERNELS", "small")
    sizes = [700, 3, 260, 1, 2, 510, 64, 65, 33, 400, 129]
    batches = [synthetic_panel(v, 16, 20, seed=300 + i) for i, v in enumerate(sizes)]
    batches.insert(4, synthetic_panel(150, 64, 20, seed=350))
    batches.insert(8, synthetic_panel(200, 16, 20, seed=351, multiallelic_frac=0.3))
    args = default_table_args()
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    job = hmm.Job(batches, t, p)
    job.run()
    got = job.fetch_all()
    job.run()
    again = job.fetch_all()
    job.close()
    for b, r, r2 in zip(batches, got, again):
        assert np.array_equal(r.lik, r2.lik) and np.array_equal(r.lik_exp, r2.lik_exp)
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, r, ref)
        alone = hmm.genotype_contig(b, t, p)
        if mode == "chunked":   # (a lone call runs chunked: the same kernels — PG_KERNELS=small is still set —, the same bits)
            assert np.array_equal(r.lik, alone.lik) and np.array_equal(r.lik_exp, alone.lik_exp)


@pytest.mark.parametrize("reg", [0.01, 0.0])
def test_small16_kernel_phase2_of_fused_jobs_vs_oracle_and_general(reg, orc, monkeypatch):
    """Fused jobs of all-biallelic 16-path chains (DevContig::small == 2): phase 2 runs on k_sweep_small16 as well — partner
    columns prefetched into registers three steps ahead, the posterior added by row allele with exact 0 / 1 multipliers and
    by column allele inside the 16-lane row, four class sums per column for k_bins_lean2 — and k_records writes the compact
    records only.  Thirteen chains of 1 ... 640 columns in one job (a partial last wave; the one-column chain stays on the
    general kernel), regularised and unregularised table (fall-back columns re-formed from the stored backward column):
    every chain matches the oracle, and PG_KERNELS=small,nosmall2 (phase 2 on the general kernel) to fp64 rounding."""
    monkeypatch.setenv("PG_SWEEP_MODE", "fused")
    sizes = [640, 3, 260, 1, 2, 511, 64, 65, 33, 400, 129, 7, 300]
    batches = [synthetic_panel(v, 16, 20, seed=700 + i, undefined_frac=0.05) for i, v in enumerate(sizes)]
    if reg == 0.0:
        for b in batches:
            if b.n_variants > 3:
                b.kmer_count[::3] = 0
    args = (6, 108, 54, reg)
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    out = {}
    for kern in ("small", "small,nosmall2"):
        monkeypatch.setenv("PG_KERNELS", kern)
        job = hmm.Job(batches, t, p)
        job.run()
        out[kern] = job.fetch_all()
        job.close()
    monkeypatch.delenv("PG_KERNELS", raising=False)
    for b, r, g in zip(batches, out["small"], out["small,nosmall2"]):
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, r, ref)
        assert_parity(b, g, ref)
        a, c = r.likelihoods_ld(), g.likelihoods_ld()
        den = np.maximum(np.abs(a), np.abs(c))
        assert float(np.where(den > 0, np.abs(a - c) / np.where(den > 0, den, 1), 0).max()) < 1e-11


@pytest.mark.parametrize("lean2", ["1", "0"])
@pytest.mark.parametrize("C_odd", [False, True])
def test_triangle_storage_fused_lean_vs_oracle_and_full_columns(C_odd, lean2, orc, monkeypatch):
    """Fused jobs store the (symmetric) columns of lean chains as upper triangles — diagonal halved, nothing
    below it — and phase 2 sums the stored half (k_bins doubles).  Unregularised table: forward fall-backs in
    both halves (the stored uniform column, the re-formed bins of k_bins) and all-zero backward columns go
    through the triangle path; PG_KERNELS=notri (full columns) must give the same bins to fp64 rounding."""
    monkeypatch.setenv("PG_SWEEP_MODE", "fused")
    l2 = [] if lean2 == "1" else ["nolean2"]   # phase 2 on k_sweep_lean2 (default) or on the general kernel's triangle ring
    def kernels(*more):
        toks = l2 + list(more)
        if toks:
            monkeypatch.setenv("PG_KERNELS", ",".join(toks))
        else:
            monkeypatch.delenv("PG_KERNELS", raising=False)
    for seed, reg in ((15, 0.0), (16, 0.01), (17, 0.0)):
        args = (6, 108, 54, reg)
        b = synthetic_panel(331 if C_odd else 330, 64, 20, seed=seed)
        if reg == 0.0:
            b.kmer_count[::3] = 0
            b.kmer_count[1::17] = 60000
        t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
        kernels()
        job = hmm.Job([b], t, p)
        job.run()
        assert job.sweep_mode()[0] == "fused"
        tri = job.fetch(0)
        job.close()
        kernels("notri")
        full = hmm.genotype_contig(b, t, p)
        kernels()
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, tri, ref)
        assert_parity(b, full, ref)
        a, c = tri.likelihoods_ld(), full.likelihoods_ld()
        den = np.maximum(np.abs(a), np.abs(c))
        assert float(np.where(den > 0, np.abs(a - c) / np.where(den > 0, den, 1), 0).max()) < 1e-11


def test_limits_are_reported_not_silently_wrong():
    b = synthetic_panel(4, 1100, 10, seed=1)
    with pytest.raises(hmm.PanGenieError) as e:
        hmm.genotype_contig(b, hmm.ProbabilityTable(*default_table_args()), hmm.make_params())
    assert e.value.code == hmm._lib.PG_ERR_UNSUPPORTED


def test_many_alleles_per_object_vs_oracle(orc):
    """More than 32 (up to 256) alleles in one UniqueKmers object: presence bitmap of 256 slots."""
    b = synthetic_panel(60, 16, 160, seed=3, multiallelic_frac=1.0, max_alleles=90)
    assert int(np.diff(b.allele_off).max()) > 64
    args = default_table_args()
    res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5))
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    assert_parity(b, res, ref)


def test_deep_bins_keep_relative_precision(orc, monkeypatch):
    """The round-1 soak failure, now a test: 16-allele columns, H = 65, K = 128, unregularised table —
    genotype bins hundreds to thousands of decades below their variant's largest bin.  Every bin holds
    1e-6 relative, in the chunked and in the fused mode (narrow panel for the latter), and the two modes
    agree with each other."""
    args = (6, 108, 54, 0.0)
    wide = synthetic_panel(150, 65, 128, seed=99, multiallelic_frac=0.6, max_alleles=17, local_alts=15, undefined_frac=0.05)
    narrow = synthetic_panel(300, 65, 128, seed=98, multiallelic_frac=0.6, max_alleles=5)
    for b in (wide, narrow):
        b.kmer_count[::3] = 0
        b.kmer_count[1::17] = 300
    ref = orc.genotype_contig(wide, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    nz = ref.lik[ref.lik > 0]
    assert float(np.log10(nz.max() / nz.min())) > 1000  # the panel really reaches that deep
    res = hmm.genotype_contig(wide, hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5))
    assert_parity(wide, res, ref)
    refn = orc.genotype_contig(narrow, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    out = {}
    for mode in ("fused", "chunked"):
        monkeypatch.setenv("PG_SWEEP_MODE", mode)
        monkeypatch.setenv("PG_CHUNK_COLS", "64")
        out[mode] = hmm.genotype_contig(narrow, hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5))
        assert_parity(narrow, out[mode], refn)
    a, c = out["fused"].likelihoods_ld(), out["chunked"].likelihoods_ld()
    den = np.maximum(np.abs(a), np.abs(c))
    assert float(np.where(den > 0, np.abs(a - c) / np.where(den > 0, den, 1), 0).max()) < 1e-10


def test_resident_job_survives_table_modify_and_destroy(orc):
    """a job keeps its own device copy of the ProbabilityTable."""
    args = default_table_args()
    b = synthetic_panel(200, 16, 20, seed=4)
    t = hmm.ProbabilityTable(*args)
    p = hmm.make_params(1.26, False, 1e-5)
    job_a = hmm.Job([b], t, p)
    t.modify(27, 13, 0.3, 0.3, 0.4)
    job_b = hmm.Job([b], t, p)
    job_b.run()
    del t
    job_a.run()
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    assert_parity(b, job_a.fetch(0), ref)
    to = orc.OracleTable(*args)
    to.modify(27, 13, 0.3, 0.3, 0.4)
    assert_parity(b, job_b.fetch(0), orc.genotype_contig(b, to, orc.make_params(1.26, False, 1e-5)))
    job_a.close(); job_b.close()


@pytest.mark.parametrize("mode", ["fused", "chunked"])
def test_cohort_job_vs_oracle(mode, orc, monkeypatch):
    """8 samples x 3 contigs against ONE index uploaded once (pg_cohort_new); every
    (sample, contig) chain against the oracle run on that sample's counts; per-sample upload moves
    2 K + 2 bytes per variant."""
    from pangenie_amd.panel import synthetic_sample_counts
    monkeypatch.setenv("PG_SWEEP_MODE", mode)
    monkeypatch.setenv("PG_CHUNK_COLS", "128")
    args = default_table_args()
    index = [synthetic_panel(350, 64, 20, seed=60), synthetic_panel(240, 16, 20, seed=61, multiallelic_frac=0.3),
             synthetic_panel(300, 32, 24, seed=62, multiallelic_frac=0.2)]
    n_samples = 8
    samples = []
    for s in range(n_samples):
        kcs, covs = zip(*[synthetic_sample_counts(ix, seed=900 + 10 * s + c) for c, ix in enumerate(index)])
        samples.append((list(kcs), list(covs)))
    job = hmm.Job.cohort(index, samples, hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5))
    assert job.n_chains == n_samples * len(index)
    up = job.upload_bytes()
    per_sample = sum(2 * int(ix.kmer_off[-1]) + 2 * ix.n_variants for ix in index)
    # per sample exactly 2 K + 2 bytes per variant; the index (+ genotype offsets + the table) goes up once
    assert up["samples"] == n_samples * per_sample
    assert 0 < up["index"] < sum(ix.nbytes() + 8 * (ix.n_variants + 1) for ix in index) + (1 << 20)
    job.run()
    for s in range(n_samples):
        for c, ix in enumerate(index):
            b = ix.with_counts(samples[s][0][c], samples[s][1][c])
            ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
            assert_parity(b, job.fetch(s * len(index) + c), ref)
    # the next batch of samples: counts only, the index stays resident
    job.upload(samples[::-1])
    assert job.upload_bytes() == {"index": 0, "samples": n_samples * per_sample}
    job.run()
    b = index[1].with_counts(samples[-1][0][1], samples[-1][1][1])
    assert_parity(b, job.fetch(1), orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5)))
    job.close()


def test_cohort_upload_overlapped_with_the_run(orc):
    """pg_job_upload_begin / _end: the next batch of samples is copied into the job's second set of per-sample arrays
    while the current batch is genotyped.  Three batches through the pipeline (both sets used twice): every batch's
    results are bit for bit those of a fresh cohort job on that batch, and match the oracle; misuse is an error."""
    from pangenie_amd.panel import synthetic_sample_counts
    args = default_table_args()
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    index = [synthetic_panel(260, 64, 20, seed=70), synthetic_panel(180, 16, 20, seed=71, multiallelic_frac=0.3)]
    n_samples = 70   # (140 chains: more than one staging piece)

    def batch(k):
        out = []
        for s in range(n_samples):
            kcs, covs = zip(*[synthetic_sample_counts(ix, seed=5000 * k + 10 * s + c) for c, ix in enumerate(index)])
            out.append((list(kcs), list(covs)))
        return out

    batches = [batch(k) for k in range(4)]
    fresh = []
    for k in range(4):
        j = hmm.Job.cohort(index, batches[k], t, p)
        j.run()
        fresh.append([j.fetch(c) for c in range(j.n_chains)])
        j.close()
    job = hmm.Job.cohort(index, batches[0], t, p)
    with pytest.raises(hmm.PanGenieError):
        job.upload_end()                       # nothing in flight
    for k in range(4):
        if k + 1 < 4:
            job.upload_begin(batches[k + 1])   # copies while batch k runs
            with pytest.raises(hmm.PanGenieError):
                job.upload_begin(batches[k + 1])   # one at a time
        job.run()
        got = [job.fetch(c) for c in range(job.n_chains)]
        for c in range(job.n_chains):
            assert np.array_equal(got[c].lik, fresh[k][c].lik) and np.array_equal(got[c].lik_exp, fresh[k][c].lik_exp)
            assert np.array_equal(got[c].coverage, fresh[k][c].coverage) and np.array_equal(got[c].kept, fresh[k][c].kept)
        if k + 1 < 4:
            job.upload_end()
            with pytest.raises(hmm.PanGenieError):
                job.fetch(0)                   # the previous results are gone once the new batch is current
    for c in (0, 1, job.n_chains - 1):
        s_, ci = divmod(c, len(index))
        b = index[ci].with_counts(batches[3][s_][0][ci], batches[3][s_][1][ci])
        assert_parity(b, got[c], orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5)))
    assert job.upload_bytes()["samples"] == n_samples * sum(2 * int(ix.kmer_off[-1]) + 2 * ix.n_variants for ix in index)
    job.close()


@pytest.mark.parametrize("H,V,local_alts", [(16, 300, 12), (64, 200, 20), (27, 150, 8), (128, 60, 31)])
def test_wide_columns_vs_oracle(H, V, local_alts, orc):
    """Columns with more than 5 (up to 32) distinct alleles on the selected paths: emission table in
    the side buffer, posteriors by k_post in blocks of row alleles (always the chunked mode)."""
    b = synthetic_panel(V, H, 128, seed=31 + H, multiallelic_frac=0.5, max_alleles=32, local_alts=local_alts,
                        undefined_frac=0.05)
    pa = b.path_allele.reshape(V, H)
    assert max(len(set(r)) for r in pa) > 5
    args = default_table_args()
    res = hmm.genotype_contig(b, hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5))
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    assert_parity(b, res, ref)


def test_c_abi_gather_single_rank_rccl_loopback(monkeypatch):
    """The multi-GPU exchange behind the C ABI (pg_comm_* / pg_hmm_gather) on ONE GPU: a world-size-1 RCCL
    communicator, the rank's packed posteriors sent to itself through ncclSend / ncclRecv
    (PG_GATHER_LOOPBACK), compared with what pg_job_fetch returns — and the plain world-1 path, which
    is a device-to-device copy."""
    import ctypes as C
    import torch
    lib = hmm._lib.load_hip()
    batches = [synthetic_panel(300, 16, 20, seed=81), synthetic_panel(200, 64, 20, seed=82, multiallelic_frac=0.3)]
    job = hmm.Job(batches, hmm.ProbabilityTable(*default_table_args()), hmm.make_params(1.26, False, 1e-5))
    job.run()
    _, _, n = job.packed_results()
    want_l = np.concatenate([job.fetch(i).lik for i in range(2)])
    want_e = np.concatenate([job.fetch(i).lik_exp for i in range(2)])
    assert n == want_l.size
    err = C.create_string_buffer(512)
    uid = (C.c_uint8 * 128)()
    assert lib.pg_comm_unique_id(uid, err, 512) == 0, err.value
    comm = C.c_void_p()
    assert lib.pg_comm_init(uid, 1, 0, 0, C.byref(comm), err, 512) == 0, err.value
    assert lib.pg_comm_rank(comm) == 0 and lib.pg_comm_world(comm) == 1
    plan = (C.c_uint64 * 1)(n)
    for loop in (False, True):
        if loop:
            monkeypatch.setenv("PG_GATHER_LOOPBACK", "1")
        got_l = torch.zeros(n, dtype=torch.float64, device="cuda")
        got_e = torch.zeros(n, dtype=torch.int32, device="cuda")
        rc = lib.pg_hmm_gather(comm, job.h, 0, plan, C.c_void_p(got_l.data_ptr()), C.c_void_p(got_e.data_ptr()), err, 512)
        assert rc == 0, err.value
        assert np.array_equal(got_l.cpu().numpy(), want_l) and np.array_equal(got_e.cpu().numpy(), want_e)
    lib.pg_comm_destroy(comm)
    job.close()


def test_abi_gather_helper_as_bench_uses_it(monkeypatch):
    """pangenie_amd.dist.AbiGather (what `bench.py --gpus N` gathers with): a world-size-1 communicator made from an id,
    the job's packed posteriors through ncclSend / ncclRecv into the root's buffers — with torch (and its own RCCL) in
    the process."""
    import torch
    from pangenie_amd.dist import AbiGather
    monkeypatch.setenv("PG_GATHER_LOOPBACK", "1")
    batches = [synthetic_panel(250, 64, 20, seed=91), synthetic_panel(180, 16, 20, seed=92)]
    job = hmm.Job(batches, hmm.ProbabilityTable(*default_table_args()), hmm.make_params(1.26, False, 1e-5))
    job.run()
    _, _, n = job.packed_results()
    g = AbiGather(0, 1, 0)
    g.gather(job, [n])
    torch.cuda.synchronize()
    want_l = np.concatenate([job.fetch(i).lik for i in range(2)])
    want_e = np.concatenate([job.fetch(i).lik_exp for i in range(2)])
    assert np.array_equal(g.lik_all.cpu().numpy()[:n], want_l) and np.array_equal(g.exp_all.cpu().numpy()[:n], want_e)
    g.gather(job, [n])  # (buffers are reused)
    g.close()
    job.close()


def _check_normalised(b, r):
    n = normalized_bins(b, r.likelihoods_ld())
    sums = np.add.reduceat(np.concatenate([n, np.zeros(1, n.dtype)]), b.geno_off[:-1].astype(np.int64))[:b.n_variants]
    kept = r.kept.astype(bool)
    assert np.allclose(sums[kept].astype(float), 1.0, atol=1e-12)
    assert not np.isnan(n.astype(float)).any()
    return n


def test_config3_whole_genome_full_size(orc):
    """BASELINE.json configs[3] at full size on ONE GPU: 24 contigs (human chromosome proportions), 5 M
    variants, 64 haplotypes.  Size-independent properties on the full job (determinism, normalised
    posteriors sum to 1, every variant accounted for) and oracle parity of the same device path on the
    first 2 000 variants of three of its contigs."""
    from bench import genome_contig_sizes
    sizes = genome_contig_sizes(5_000_000)
    batches = [synthetic_panel(sizes[i], 64, 20, seed=12345 + 1000 * i) for i in range(24)]
    t = hmm.ProbabilityTable(*default_table_args())
    p = hmm.make_params(1.26, False, 1e-5)
    job = hmm.Job(batches, t, p)
    job.run()
    first = [job.fetch(i) for i in (0, 11, 23)]
    job.run()
    for k, i in enumerate((0, 11, 23)):
        r2 = job.fetch(i)
        assert (first[k].lik == r2.lik).all() and (first[k].lik_exp == r2.lik_exp).all()
        assert r2.n_columns == int(r2.kept.sum()) > 0.9 * sizes[i]
        _check_normalised(batches[i], r2)
    whole = job.fetch(20)   # the shortest contig (chr21's share: ~76 k variants) — a WHOLE chain of the 24-chain job, below
    job.close()
    hmm._lib.load_hip().pg_hmm_release_cache()
    args = default_table_args()
    # every bin of that chain against the oracle: ~38 k columns per half-chain, ten chunks of 4096 handed over, the
    # two directions meeting in the middle — as it ran INSIDE the 24-chain job, not re-run alone (parity gap 1)
    assert sizes[20] > 70_000
    ref = orc.genotype_contig(batches[20], orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    worst = assert_parity(batches[20], whole, ref)
    log_parity(f"whole chain 20 of genome24_h64 ({sizes[20]} variants x 64 paths, {whole.n_columns} columns) vs oracle: worst relative error {worst:.3e}")
    for i in (0, 11, 23):
        sl = batches[i].slice(0, 2000)
        res = hmm.genotype_contig(sl, t, p)
        ref = orc.genotype_contig(sl, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(sl, res, ref)


def test_config4_shape_h128_multiallelic(orc):
    """BASELINE.json configs[4] shape, one GPU's slice: 60 k variants x 128 haplotypes, 20 % multiallelic.
    Properties at full slice size + oracle parity on a 300-variant piece."""
    b = synthetic_panel(60_000, 128, 20, seed=4242, multiallelic_frac=0.2)
    t = hmm.ProbabilityTable(*default_table_args())
    p = hmm.make_params(1.26, False, 1e-5)
    job = hmm.Job([b], t, p)
    job.run()
    r1 = job.fetch(0)
    job.run()
    r2 = job.fetch(0)
    job.close()
    assert (r1.lik == r2.lik).all() and (r1.lik_exp == r2.lik_exp).all()
    _check_normalised(b, r1)
    sl = b.slice(30_000, 30_300)
    args = default_table_args()
    assert_parity(sl, hmm.genotype_contig(sl, t, p), orc.genotype_contig(sl, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5)))


def test_config4_one_gpu_share_full_size(orc):
    """BASELINE.json configs[4] at ONE GPU's real size: the share rank 0 gets when the 24 contigs of the 5 M-variant,
    128-haplotype, 20 %-multiallelic panel are spread over 8 GPUs by longest-processing-time-first (what
    `bench.py --gpus 8 --workload hprc_h128` gives each rank): ~625 k variants, ~82 GB of columns, the large-arena
    multi-chunk path at HP = 128 with wide and narrow multiallelic columns.  Size-independent properties on the full
    job (determinism, normalised posteriors, every variant accounted for) + oracle parity of the same device path on
    slices from the start, the middle and the end of its longest chain."""
    from bench import genome_contig_sizes
    from pangenie_amd.dist import assign_chains
    sizes = genome_contig_sizes(5_000_000)
    mine = assign_chains([float(v) * 128 * 128 for v in sizes], 8)[0]
    assert 550_000 < sum(sizes[i] for i in mine) < 700_000
    batches = [synthetic_panel(sizes[i], 128, 20, seed=12345 + 1000 * i, multiallelic_frac=0.2) for i in mine]
    t = hmm.ProbabilityTable(*default_table_args())
    p = hmm.make_params(1.26, False, 1e-5)
    job = hmm.Job(batches, t, p)
    assert job.device_bytes() > 80e9
    job.run()
    first = job.fetch_all()
    job.run()
    again = job.fetch_all()
    job.close()
    hmm._lib.load_hip().pg_hmm_release_cache()
    for b, r1, r2 in zip(batches, first, again):
        assert (r1.lik == r2.lik).all() and (r1.lik_exp == r2.lik_exp).all()
        assert r2.n_columns == int(r2.kept.sum()) > 0.9 * b.n_variants
        _check_normalised(b, r2)
    args = default_table_args()
    big = batches[0]
    V = big.n_variants
    for lo in (0, V // 2 - 150, V - 300):
        sl = big.slice(lo, lo + 300)
        assert_parity(sl, hmm.genotype_contig(sl, t, p), orc.genotype_contig(sl, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5)))


def test_config1_full_size_50k_variants_16_paths(orc):
    """BASELINE.json configs[1] at its stated size: one contig, 50 000 variants x 16 haplotypes, ~20 unique k-mers per
    variant.  Size-independent checks on the whole chain (determinism, normalised posteriors sum to 1 at every kept
    variant, kept-column count), the oracle on three 400-variant windows of it run as contigs of their own (the same
    kernels, prologue and resume code at a size the oracle finishes in seconds), and the chunked and the fused mode of
    the whole chain agreeing to fp64 rounding with identical calls."""
    V, H = 50_000, 16
    b = synthetic_panel(V, H, 20, seed=12345)
    args = default_table_args()
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    job = hmm.Job([b], t, p)
    job.run()
    r1 = job.fetch(0)
    job.run()
    r2 = job.fetch(0)
    job.close()
    assert (r1.lik == r2.lik).all() and (r1.lik_exp == r2.lik_exp).all()
    n1 = normalized_bins(b, r1.likelihoods_ld())
    sums = np.add.reduceat(n1, b.geno_off[:-1].astype(np.int64))
    kept = r1.kept.astype(bool)
    assert np.allclose(sums[kept].astype(float), 1.0, atol=1e-12)
    assert r1.n_columns == int(kept.sum()) > 0.9 * V
    for lo in (0, 24_800, V - 400):
        sl = b.slice(lo, lo + 400)
        assert_parity(sl, hmm.genotype_contig(sl, t, p), orc.genotype_contig(sl, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5)))
    import os
    os.environ["PG_SWEEP_MODE"] = "fused"
    try:
        rf = hmm.genotype_contig(b, t, p)
    finally:
        os.environ.pop("PG_SWEEP_MODE", None)
    # the WHOLE chain against the oracle, every bin, in both sweep modes (the oracle needs ~1 s for it)
    ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    w_chunked, w_fused = assert_parity(b, r1, ref), assert_parity(b, rf, ref)
    log_parity(f"configs[1] whole chain (50 000 x 16) vs oracle: worst relative error chunked {w_chunked:.3e}, fused {w_fused:.3e}")
    nf = normalized_bins(b, rf.likelihoods_ld())
    rel = rel_errors(b, nf, n1)
    assert float(rel[n1 > 1e-200].max()) < 1e-9
    assert (calls(b, nf) == calls(b, n1)).all()


def test_full_size_properties(orc):
    """BASELINE.json configs[2] shape (200k variants x 64 haplotypes): size-independent checks.
    (a) determinism; (b) normalised posteriors sum to 1; (c) reversibility: the Li-Stephens
    chain with uniform start is time-reversible, so genotyping the mirrored contig must give
    the same normalised posteriors for every variant; (d) the WHOLE chain against the oracle, every bin
    (~45 s of oracle on a worker thread while the device does (c))."""
    from concurrent.futures import ThreadPoolExecutor
    V, H = 200_000, 64
    b = synthetic_panel(V, H, 20, seed=12345)
    t = hmm.ProbabilityTable(*default_table_args())
    p = hmm.make_params(1.26, False, 1e-5)
    pool = ThreadPoolExecutor(1)   # (ctypes drops the GIL for the duration of the oracle call)
    ref_future = pool.submit(orc.genotype_contig, b, orc.OracleTable(*default_table_args()), orc.make_params(1.26, False, 1e-5))
    job = hmm.Job([b], t, p)
    job.run()
    r1 = job.fetch(0)
    job.run()
    r2 = job.fetch(0)
    job.close()
    assert (r1.lik == r2.lik).all() and (r1.lik_exp == r2.lik_exp).all()
    n1 = normalized_bins(b, r1.likelihoods_ld())
    G = np.diff(b.geno_off.astype(np.int64))
    sums = np.add.reduceat(n1, b.geno_off[:-1].astype(np.int64))
    kept = r1.kept.astype(bool)
    assert np.allclose(sums[kept].astype(float), 1.0, atol=1e-12)
    assert r1.n_columns == int(kept.sum()) > 0.9 * V

    # mirrored contig
    pos = b.variant_pos.astype(np.int64)
    rev = synthetic_panel(1, H, 20, seed=1)  # container
    idx = np.arange(V)[::-1]
    K = np.diff(b.kmer_off.astype(np.int64)); A = np.diff(b.allele_off.astype(np.int64))
    def regroup(arr, off, n):
        parts = np.split(arr, off[1:-1].astype(np.int64))
        return np.concatenate([parts[i] for i in idx]) if len(parts) else arr
    from pangenie_amd.panel import ContigBatch
    koff = np.zeros(V + 1, np.uint32); np.cumsum(K[idx], out=koff[1:])
    aoff = np.zeros(V + 1, np.uint32); np.cumsum(A[idx], out=aoff[1:])
    rb = ContigBatch(H, (pos.max() - pos[idx] + 10000).astype(np.uint64), b.coverage[idx], koff,
                     regroup(b.kmer_count, b.kmer_off, V), aoff, regroup(b.allele_id, b.allele_off, V),
                     regroup(b.allele_flags, b.allele_off, V), regroup(b.allele_kmer_off, b.allele_off, V),
                     regroup(b.allele_kmer_mask, b.allele_off, V),
                     b.path_allele.reshape(V, H)[idx].reshape(-1))
    rr = hmm.genotype_contig(rb, t, p)
    nr = normalized_bins(rb, rr.likelihoods_ld())
    # map reversed bins back to forward order
    goff_r = rb.geno_off.astype(np.int64)
    parts = np.split(nr, goff_r[1:-1])
    back = np.concatenate([parts[V - 1 - v] for v in range(V)])
    rel = rel_errors(b, back, n1)
    big = n1 > 1e-200
    assert float(rel[big].max()) < 1e-6, float(rel[big].max())
    assert (calls(b, back) == calls(b, n1)).all()
    worst = assert_parity(b, r1, ref_future.result())
    pool.shutdown()
    log_parity(f"configs[2] whole chain (200 000 x 64, {r1.n_columns} columns) vs oracle: worst relative error {worst:.3e}")


def test_whole_chain_h128_multiallelic_vs_oracle(orc):
    """One whole 20 000-variant chain at 128 paths, a fifth of the objects multiallelic (BASELINE configs[4]'s shape) — every
    bin against the oracle, chunked (k_sweep_leanx + k_post, three chunks per half-chain) and fused (the general kernel's
    phase 2): column 10 000 of a 20 000-column chain, not a 300-variant window re-run alone."""
    b = synthetic_panel(20_000, 128, 20, seed=20128, multiallelic_frac=0.2)
    args = default_table_args()
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(1)
    ref_future = pool.submit(orc.genotype_contig, b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
    import os
    got = {}
    for mode in ("chunked", "fused"):
        os.environ["PG_SWEEP_MODE"] = mode
        try:
            got[mode] = hmm.genotype_contig(b, t, p)
        finally:
            os.environ.pop("PG_SWEEP_MODE", None)
    ref = ref_future.result()
    pool.shutdown()
    worst = {m: assert_parity(b, r, ref) for m, r in got.items()}
    log_parity(f"whole chain 20 000 x 128, 20 % multiallelic ({ref.n_columns} columns) vs oracle: worst relative error chunked {worst['chunked']:.3e}, fused {worst['fused']:.3e}")


@pytest.mark.parametrize("shape", [(700, 5, 20), (600, 16, 12), (500, 30, 32), (400, 64, 20), (300, 33, 7)], ids=lambda s: "V%d_H%d_K%d" % s)
def test_prep_four_variants_per_wave_vs_one(shape, orc, monkeypatch):
    """k_prep_bi (chains of biallelic objects: four variants per wave, DPP folds) against k_prep (PG_KERNELS=prepwave: one
    variant per wave) and against the oracle: same kept columns and present alleles, likelihoods equal to rounding
    (the products are taken in a different order).  Undefined alleles, k-mer-less variants, zero counts, and a table
    without regularisation (exact zeros: the all_zeros rule, uniform columns) are in."""
    V, H, K = shape
    batch = synthetic_panel(V, H, K, seed=31 + V, undefined_frac=0.05, zero_kmer_frac=0.05)
    batch.path_allele.reshape(V, H)[V // 2] = 1          # a variant whose selected paths all carry the ALT allele
    batch.path_allele.reshape(V, H)[V // 2 + 1, :] = 0   # and one that is not a column at all
    batch._c = None
    for targs in (default_table_args(), (6, 108, 54, 0.0)):
        table, otab = hmm.ProbabilityTable(*targs), orc.OracleTable(*targs)
        prm = hmm.make_params(1.26, False, 1e-5)
        monkeypatch.delenv("PG_KERNELS", raising=False)
        four = hmm.genotype_contig(batch, table, prm)
        monkeypatch.setenv("PG_KERNELS", "prepwave")
        one = hmm.genotype_contig(batch, table, prm)
        monkeypatch.delenv("PG_KERNELS", raising=False)
        assert four.n_columns == one.n_columns and np.array_equal(four.kept, one.kept)
        assert np.array_equal(four.allele_present, one.allele_present)
        a, b = four.likelihoods_ld(), one.likelihoods_ld()
        denom = np.maximum(np.abs(a), np.abs(b))
        rel = np.where(denom > 0, np.abs(a - b) / np.where(denom > 0, denom, 1), 0)
        assert float(rel.max()) < 1e-12, float(rel.max())
        assert_parity(batch, four, orc.genotype_contig(batch, otab, orc.make_params(1.26, False, 1e-5)))


@pytest.mark.parametrize("shape", [(600, 17, 20, 0.2), (500, 64, 20, 0.3), (400, 30, 12, 0.45), (300, 16, 40, 0.2), (1000, 16, 40, 0.4)], ids=lambda s: "V%d_H%d_K%d_m%g" % s)
def test_prep_mixed_chains_two_allele_objects_on_the_fast_kernel(shape, orc, monkeypatch):
    """Chains with multiallelic objects (HPRC-style panels, the 15 + 1 sampled paths): k_prep_bi takes the two-allele
    objects with at most 32 k-mers, k_prep the others (DevContig::prep_fast == 2) — against k_prep alone (PG_KERNELS=prepwave)
    and the oracle; K = 40 puts two-allele objects with more than 32 k-mers on k_prep as well."""
    V, H, K, multi = shape
    # (the 1000-variant shape is test_panels_vs_oracle's: a fifth of the objects without k-mers, a fifth with an undefined allele —
    #  waves of k_prep_m4 whose first object is no column at all next to objects with 40 k-mers: the case a shuffle butterfly lost)
    heavy = V == 1000
    batch = synthetic_panel(V, H, K, seed=(1000 + V + H) if heavy else 77 + V, multiallelic_frac=multi, undefined_frac=0.2 if heavy else 0.05, zero_kmer_frac=0.2 if heavy else 0.05)
    for targs in (default_table_args(), (6, 108, 54, 0.0)):
        table, otab = hmm.ProbabilityTable(*targs), orc.OracleTable(*targs)
        prm = hmm.make_params(1.26, False, 1e-5)
        monkeypatch.delenv("PG_KERNELS", raising=False)
        mixed = hmm.genotype_contig(batch, table, prm)
        monkeypatch.setenv("PG_KERNELS", "prepwave")
        one = hmm.genotype_contig(batch, table, prm)
        monkeypatch.delenv("PG_KERNELS", raising=False)
        assert mixed.n_columns == one.n_columns and np.array_equal(mixed.kept, one.kept)
        assert np.array_equal(mixed.allele_present, one.allele_present)
        a, b = mixed.likelihoods_ld(), one.likelihoods_ld()
        denom = np.maximum(np.abs(a), np.abs(b))
        rel = np.where(denom > 0, np.abs(a - b) / np.where(denom > 0, denom, 1), 0)
        assert float(rel.max()) < 1e-11, float(rel.max())
        assert_parity(batch, mixed, orc.genotype_contig(batch, otab, orc.make_params(1.26, False, 1e-5)))
