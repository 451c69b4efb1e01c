"""Pin the CPU oracle (oracle/pg_oracle.c) on the reference's own known-answer tests.

Tolerance = the reference's own: |a-b| < 1e-7 absolute (reference tests/utils.cpp:9-11).
"""
import numpy as np
import pytest

from oracle import pyoracle as orc
from pangenie_amd.genotyping_result import results_from_flat
from pangenie_amd.panel import flatten
from tests.fixtures_util import build_batch, build_variant, fill_table, triple

TOL = 1e-7


def make_table(spec):
    if spec["default"]:
        t = orc.OracleTable(default=True)
    else:
        t = orc.OracleTable(*spec["args"])
    return fill_table(t, spec, orc.copynumber_regularized)


def run_case(case):
    batch = build_batch(case["variants"], case["hmm"]["only_paths"])
    table = make_table(case["table"])
    h = case["hmm"]
    params = orc.make_params(h["recombrate"], h["uniform"], h["effective_N"])
    r = orc.genotype_contig(batch, table, params)
    res = results_from_flat(batch, r.lik, r.kept, r.allele_present, r.n_kmers, r.coverage)
    return batch, r, res


def test_hmm_known_answers(golden):
    for case in golden["hmm"]:
        _, raw, res = run_case(case)
        if case["hmm"]["normalize"]:
            for g in res:
                g.normalize()
        got = [triple(g) for g in res]
        assert np.allclose(got, case["expected_likelihoods"], rtol=0, atol=TOL), case["name"]
        if "expected_coverage" in case:
            assert [g.coverage() for g in res] == case["expected_coverage"]
            assert [g.nr_unique_kmers() for g in res] == case["expected_n_kmers"]
        if "expected_specific" in case:
            spec = [triple(g.get_specific_likelihoods(d)) for g, d in zip(res, case["defined_alleles"])]
            assert np.allclose(spec, case["expected_specific"], rtol=0, atol=TOL), case["name"]
        if "expected_gt" in case:
            assert [list(g.get_likeliest_genotype()) for g in res] == case["expected_gt"]
        if "expected_after_normalize" in case:
            for g in res:
                g.normalize()
            assert np.allclose([triple(g) for g in res], case["expected_after_normalize"], rtol=0, atol=TOL)


def test_viterbi_known_answer_and_forms(golden):
    """HMM(run_phasing=true): the reference's own haplotype pin (tests/HMMTest.cpp:393-439) on both forms of the oracle's
    Viterbi, then form 1 (O(H^2)) == form 0 (the reference's O(H^4) loop) on every HMM fixture and on seeded panels."""
    for case in golden["hmm"]:
        batch = build_batch(case["variants"], case["hmm"]["only_paths"])
        table = make_table(case["table"])
        h = case["hmm"]
        params = orc.make_params(h["recombrate"], h["uniform"], h["effective_N"], run_genotyping=False, run_phasing=True)
        a = orc.viterbi_contig(batch, table, params, form=0)
        b = orc.viterbi_contig(batch, table, params, form=1)
        for name in ("hap1", "hap2", "kept", "n_kmers", "coverage"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), (case["name"], name)
        assert a.n_columns == b.n_columns
        if "expected_haplotype1" in case:
            e1, e2 = case["expected_haplotype1"], case["expected_haplotype2"]
            got1, got2 = a.hap1.tolist(), a.hap2.tolist()
            assert (got1 == e1 and got2 == e2) or (got1 == e2 and got2 == e1), case["name"]
    from pangenie_amd.panel import synthetic_panel
    table = orc.OracleTable(6, 108, 54, 0.01)
    for seed, (V, H, multi) in enumerate([(60, 5, 0.0), (80, 12, 0.3), (50, 20, 0.2), (40, 7, 0.5)]):
        batch = synthetic_panel(V, H, seed=400 + seed, multiallelic_frac=multi)
        for recomb, effn, uni in ((1.26, 25000.0, False), (1.26, 1e-5, False), (0.0, 25000.0, False), (1.26, 25000.0, True)):
            params = orc.make_params(recomb, uni, effn, run_genotyping=False, run_phasing=True)
            a = orc.viterbi_contig(batch, table, params, form=0)
            b = orc.viterbi_contig(batch, table, params, form=1)
            assert np.array_equal(a.hap1, b.hap1) and np.array_equal(a.hap2, b.hap2), (seed, recomb, effn, uni)
            assert a.n_columns == b.n_columns > 0


def test_hmm_combine(golden):
    by_name = {c["name"]: c for c in golden["hmm"]}
    a = run_case(by_name[golden["combine"]["first"]])[2]
    b = run_case(by_name[golden["combine"]["second"]])[2]
    for g in a + b:
        g.normalize()
    expect = [np.add(triple(x), triple(y)) for x, y in zip(a, b)]
    for x, y in zip(a, b):
        x.combine(y)
    assert np.allclose([triple(x) for x in a], expect, rtol=0, atol=TOL)


def test_emission_known_answers(golden):
    for case in golden["emission"]:
        batch = flatten([build_variant(case["variant"])])
        table = make_table(case["table"])
        E, _ = orc.emission_table(batch, table, 0)
        ids = list(batch.allele_id)
        for key, val in case["expected"].items():
            a, b = (int(x) for x in key.split(","))
            assert abs(float(E[ids.index(a), ids.index(b)]) - val) < TOL, (case["name"], key)


def test_transition_known_answers(golden):
    for c in golden["transition"]:
        t = orc.transition_probs(c["from"], c["to"], c["recombrate"], c["nr_paths"], c["uniform"], c["effective_N"])
        q = c["recomb_prob"]
        p = q + c["no_recomb_minus_recomb"]
        assert np.allclose(t.astype(float), [p * p, p * q, q * q], rtol=0, atol=TOL)
    assert list(orc.transition_probs(1, 2, 1.26, 5, True, 0.25).astype(float)) == [1.0, 1.0, 1.0]


def test_probability_table_known_answers(golden):
    for c in golden["probability_table"]:
        t = orc.OracleTable(*c["args"])
        for cov, count, exp in c["expected"]:
            assert np.allclose(t.get(cov, count).astype(float), exp, rtol=0, atol=TOL), (c["ref"], cov, count)


def test_copynumber_known_answers(golden):
    for c in golden["copynumber"]:
        got = orc.copynumber_regularized(*c["cn"], c["reg"]).astype(float)
        assert np.allclose(got, c["expected"], rtol=0, atol=TOL)


def test_column_indexer(golden):
    c = golden["column_indexer"]
    uks = [build_variant(v) for v in c["variants"]]
    table = orc.OracleTable(0, 30, 30, 0.0)
    params = orc.make_params(1.26, False, 0.25)
    w = c["with_only_paths"]
    batch = flatten(uks, w["only_paths"])
    assert batch.n_paths == w["nr_paths"]
    assert batch.path_allele.reshape(-1, batch.n_paths).tolist() == w["path_alleles"]
    r = orc.genotype_contig(batch, table, params)
    assert r.n_columns == w["n_columns"]
    assert list(np.nonzero(r.kept)[0]) == w["column_variants"]
    a = c["all_paths"]
    batch = flatten(uks, None)
    r = orc.genotype_contig(batch, table, params)
    assert (batch.n_paths, r.n_columns, list(np.nonzero(r.kept)[0])) == (a["nr_paths"], a["n_columns"], a["column_variants"])
    with pytest.raises(RuntimeError):
        flatten(uks, [7, 9])  # no paths -> "column ... is not covered by any paths"


def test_probability_table_bit_for_bit_against_the_reference_translation_units():
    """oracle/_ref/libref_table.so is the reference's OWN src/probabilitytable.cpp + src/copynumber.cpp (no cereal
    includes: they compile where they lie, `make -C oracle ref`).  The oracle's table — and the product's host-side
    table behind pg_table_get, which feeds the device its (mantissa, exponent) pairs — must give the very same long
    doubles: inside the precomputed box, on the fly outside it, with and without regularisation, default-constructed."""
    if orc.ref_table_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference)")
    from pangenie_amd import hmm

    def same(a, b):
        return all((x == y) or (np.isnan(x) and np.isnan(y)) for x, y in zip(a, b))
    for args in ((6, 108, 54, 0.01), (4, 72, 36, 0.01), (0, 1, 21, 0.0), (5, 40, 30, 0.0), (0, 300, 10, 0.001)):
        ref, o, t = orc.RefTable(*args), orc.OracleTable(*args), hmm.ProbabilityTable(*args)
        covs = list(range(0, 130, 1)) + [200, 299, 300, 1000, 65535]
        counts = list(range(0, 70)) + [100, 255, 1000, 65535]
        for cov in covs:
            for count in counts[::3] if cov % 7 else counts:
                r = ref.get(cov, count)
                assert same(o.get(cov, count), r), (args, cov, count)
                assert same(t.get(cov, count), r), (args, cov, count)
    ref, o, t = orc.RefTable(default=True), orc.OracleTable(default=True), hmm.ProbabilityTable(default=True)
    for cov, count in ((0, 0), (1, 0), (10, 3), (27, 27), (40, 0), (500, 499)):
        assert same(o.get(cov, count), ref.get(cov, count)) and same(t.get(cov, count), ref.get(cov, count)), (cov, count)
    rng = np.random.default_rng(5)
    for _ in range(500):
        cn = rng.random(3) * 10.0 ** rng.integers(-30, 1, size=3)
        reg = float(10.0 ** rng.integers(-6, 0))
        assert same(orc.copynumber_regularized(cn[0], cn[1], cn[2], reg), orc.ref_copynumber_regularized(cn[0], cn[1], cn[2], reg))
