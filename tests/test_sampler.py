"""HaplotypeSampler: oracle pinned on the reference's own tests (CPU), HIP path
bit-exact against the oracle and the same known answers (GPU)."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import pyoracle as orc  # checker only
from pangenie_amd import panel as pn
from pangenie_amd import sampler as smp

GOLD = json.loads((Path(__file__).parent / "golden" / "sampler_known_answers.json").read_text())


def build_uk(spec):
    uk = (pn.BiallelicUniqueKmers if spec["type"] == "bi" else pn.MultiallelicUniqueKmers)(spec["pos"], spec["paths"])
    for a in spec["undefined"]:
        uk.set_undefined_allele(a)
    for count, alleles in spec["kmers"]:
        uk.insert_kmer(count, alleles)
    uk.set_coverage(spec["coverage"])
    return uk


def sampler_panel(n_variants, n_paths, seed, max_alleles=4, undefined=0.02):
    """Random panel shaped like the sampler's input: many paths, few alleles, counts around the
    'present' threshold of 3 so that the costs spread over 0..25."""
    rng = np.random.default_rng(seed)
    uks = []
    pos = 1000
    mosaic = rng.integers(0, 8, n_paths)
    for v in range(n_variants):
        pos += int(rng.integers(1, 3000))
        A = int(rng.integers(2, max_alleles + 1))
        if v and rng.random() < 0.3:
            idx = rng.integers(0, n_paths, max(1, n_paths // 16))
            mosaic[idx] = rng.integers(0, 8, idx.size)
        founders = rng.integers(0, A, 8)
        p2a = founders[mosaic]
        noise = rng.random(n_paths) < 0.05
        p2a = np.where(noise, rng.integers(0, A, n_paths), p2a)
        uk = pn.MultiallelicUniqueKmers(pos, p2a.tolist()) if A > 2 or rng.random() < 0.3 else pn.BiallelicUniqueKmers(pos, p2a.tolist())
        for a in sorted(set(p2a.tolist())):
            if rng.random() < undefined:
                uk.set_undefined_allele(a)
        for a in sorted(set(p2a.tolist())):
            for _ in range(int(rng.integers(0, 6))):
                if uk.size() < 14:
                    uk.insert_kmer(int(rng.choice([0, 1, 2, 3, 5, 9, 27])), [a])
        uk.set_coverage(int(rng.integers(5, 40)))
        uks.append(uk)
    return uks


# --------------------------------------------------------------------------- #
#  CPU: the oracle against the reference's known answers
# --------------------------------------------------------------------------- #
def test_oracle_column_minima_known_answers():
    for c in GOLD["column_minima"]:
        got = orc.sampler_column_minima(c["column"], c["mask"])
        assert got == (c["first_id"], c["second_id"], c["first_val"], c["second_val"])


def test_oracle_emission_costs_known_answers():
    for c in GOLD["emissions"]:
        b = pn.flatten([build_uk(c["variant"])])
        costs = orc.sampler_emission_costs(b)
        got = {str(int(a)): int(x) for a, x in zip(b.allele_id, costs)}
        assert got == c["costs"], c["name"]


def test_oracle_transition_cost_known_answer():
    for c in GOLD["transitions"]:
        expected = int(np.trunc(-10.0 * np.log10(c["recomb_prob"])))  # `unsigned int expected_cost = -10.0 * log10(recomb_prob)`
        assert orc.sampler_transition_cost(c["from"], c["to"], c["recombrate"], c["nr_paths"], c["effective_N"]) == expected


def test_oracle_transition_cost_matches_reference_translation_unit():
    """oracle/_ref/libref_transitions.so is the reference's own samplingtransitions.cpp (built here where
    /root/reference exists; absent on the GPU box)."""
    if orc.ref_transition_cost(1, 2, 1.26, 5) is None:
        pytest.skip("oracle/_ref not built (no /root/reference)")
    rng = np.random.default_rng(7)
    n = 0
    for _ in range(20000):
        a = int(rng.integers(0, 250_000_000))
        d = int(rng.choice([1, 2, 5, 10, 50, 100, 1000, 10_000, 1_000_000, 50_000_000])) + int(rng.integers(0, 100))
        H = int(rng.choice([2, 5, 16, 64, 100, 215, 500, 1000, 5000]))
        N = float(rng.choice([0.25, 1.0, 25000.0, 1e6]))
        r = float(rng.choice([1.26, 0.01, 5.0]))
        assert orc.sampler_transition_cost(a, a + d, r, H, N) == orc.ref_transition_cost(a, a + d, r, H, N)
        n += 1
    assert n == 20000


def test_oracle_viterbi_known_answers():
    for c in GOLD["viterbi"]:
        b = pn.flatten([build_uk(s) for s in c["panel"]])
        sampled, best = orc.sampler_run(b, c["size"])
        assert best.tolist() == c["best_scores"], c["name"]
        assert sampled.tolist() == c["sampled_paths"][: c["size"]], c["name"]


def check_updated(c, uks_or_batch, sampled):
    if "updated" not in c:
        return
    # object mirror
    uks = [build_uk(s) for s in c["panel"]]
    for v, uk in enumerate(uks):
        uk.update_paths([int(p[v]) for p in sampled])
    flat = pn.flatten(uks)
    for v, (uk, want) in enumerate(zip(uks, c["updated"])):
        assert uk.size() == want["size"]
        assert [uk.get_readcount_of(i) for i in range(uk.size())] == want["counts"]
        for k, p in want["on_path"]:
            assert uk.kmer_on_path(k, p)
    # flat (vectorised) restatement gives the same batch as flattening the updated objects
    got = pn.flatten([build_uk(s) for s in c["panel"]]).update_paths(np.array(sampled))
    for f in ("variant_pos", "coverage", "kmer_off", "kmer_count", "allele_off", "allele_id", "allele_flags",
              "allele_kmer_off", "allele_kmer_mask", "path_allele"):
        assert np.array_equal(getattr(got, f), getattr(flat, f)), f
    assert got.n_paths == flat.n_paths


def test_update_paths_known_answers():
    for c in GOLD["viterbi"]:
        check_updated(c, None, c["sampled_paths"])


def test_update_paths_flat_matches_objects_random():
    for seed in range(6):
        uks = sampler_panel(40, 23, seed)
        rng = np.random.default_rng(100 + seed)
        sampled = rng.integers(0, 23, (5, 40))
        flat = pn.flatten(uks).update_paths(sampled)
        for v, uk in enumerate(uks):
            uk.update_paths(sampled[:, v].tolist())
        want = pn.flatten(uks)
        for f in ("kmer_off", "kmer_count", "allele_off", "allele_id", "allele_flags", "allele_kmer_off", "allele_kmer_mask", "path_allele"):
            assert np.array_equal(getattr(flat, f), getattr(want, f)), (seed, f)


def test_sampled_paths_mask_and_recombination():
    g = GOLD["mask_indexes"]
    s = smp.SampledPaths(g["sampled_paths"])
    for c in g["cases"]:
        assert s.mask_indexes(c["column"], c["max_index"]) == [bool(x) for x in c["mask"]]
    for col, mx in g["throws"]:
        with pytest.raises(RuntimeError):
            s.mask_indexes(col, mx)
    r = GOLD["recombination"]
    s = smp.SampledPaths(r["sampled_paths"])
    for col, path, want in r["cases"]:
        assert s.recombination(col, path) == bool(want)


def test_oracle_sampler_properties():
    """Size-independent properties: a pass never re-picks a path an earlier pass holds at that column;
    the first pass's score is the true optimum (brute force over all path sequences on a tiny panel)."""
    uks = sampler_panel(60, 30, 3)
    b = pn.flatten(uks)
    sampled, best = orc.sampler_run(b, 8)
    for v in range(b.n_variants):
        assert len(set(sampled[:, v].tolist())) == 8
    # brute force on a tiny panel
    uks = sampler_panel(5, 4, 11)
    b = pn.flatten(uks)
    cost = orc.sampler_emission_costs(b)
    import itertools
    bestv = None
    for seq in itertools.product(range(4), repeat=5):
        t = 0
        for v, p in enumerate(seq):
            a = b.path_allele[v * 4 + p]
            lo, hi = int(b.allele_off[v]), int(b.allele_off[v + 1])
            t += int(cost[lo + list(b.allele_id[lo:hi]).index(a)])
            if v and seq[v - 1] != p:
                t += orc.sampler_transition_cost(int(b.variant_pos[v - 1]), int(b.variant_pos[v]), 1.26, 4)
        bestv = t if bestv is None else min(bestv, t)
    _, best = orc.sampler_run(b, 1)
    assert int(best[0]) == bestv


# --------------------------------------------------------------------------- #
#  GPU: the HIP path against the known answers and the oracle (bit-exact)
# --------------------------------------------------------------------------- #
@pytest.mark.gpu
def test_hip_column_minima_known_answers():
    for c in GOLD["column_minima"]:
        assert smp.column_minima(c["column"], c["mask"]) == (c["first_id"], c["second_id"], c["first_val"], c["second_val"])
    rng = np.random.default_rng(5)
    for n in (2, 63, 64, 65, 300, 5000):
        col = rng.integers(0, 50, n).astype(np.uint32)
        col[rng.random(n) < 0.05] = 0xFFFFFFFF
        mask = rng.random(n) < 0.8
        assert smp.column_minima(col, mask) == orc.sampler_column_minima(col, mask)
    assert smp.column_minima([7, 7], [False, False]) == orc.sampler_column_minima([7, 7], [0, 0])


@pytest.mark.gpu
def test_hip_costs_known_answers():
    for c in GOLD["emissions"]:
        b = pn.flatten([build_uk(c["variant"])])
        e = smp.SamplingEmissions(b, 0)
        for a, want in c["costs"].items():
            assert e.get_emission_cost(int(a)) == want
    for c in GOLD["transitions"]:
        t = smp.SamplingTransitions(c["from"], c["to"], c["recombrate"], c["nr_paths"], c["effective_N"])
        assert t.compute_transition_cost(False) == 0
        assert t.compute_transition_cost(True) == int(np.trunc(-10.0 * np.log10(c["recomb_prob"])))
    e = smp.SamplingEmissions(pn.flatten([build_uk(GOLD["emissions"][2]["variant"])]), 0)
    e.penalize(0, 10); assert e.get_emission_cost(0) == 11
    e.penalize(0, 10); assert e.get_emission_cost(0) == 21
    e.penalize(0, 10); assert e.get_emission_cost(0) == 25


@pytest.mark.gpu
def test_hip_viterbi_known_answers():
    for c in GOLD["viterbi"]:
        b = pn.flatten([build_uk(s) for s in c["panel"]])
        h = smp.HaplotypeSampler(b, c["size"], add_reference=c["add_reference"])
        assert h.best_scores == c["best_scores"], c["name"]
        assert h.get_sampled_paths().sampled_paths == c["sampled_paths"], c["name"]
        if "updated" in c:
            uks = [build_uk(s) for s in c["panel"]]
            for v, uk in enumerate(uks):
                uk.update_paths([p[v] for p in c["sampled_paths"]])
            want = pn.flatten(uks)
            for f in ("kmer_off", "kmer_count", "allele_off", "allele_id", "allele_kmer_off", "allele_kmer_mask", "path_allele"):
                assert np.array_equal(getattr(h.panel, f), getattr(want, f)), (c["name"], f)
    h = smp.HaplotypeSampler(pn.flatten([build_uk(s) for s in GOLD["viterbi"][0]["panel"]]), 0)  # size 0: nothing happens
    assert h.best_scores == [] and h.get_sampled_paths().sampled_paths == []


@pytest.mark.gpu
@pytest.mark.parametrize("n_variants,n_paths,size,seed", [
    (50, 3, 2, 0), (200, 16, 5, 1), (300, 64, 15, 2), (257, 100, 15, 3), (500, 215, 15, 4), (400, 256, 8, 5),
    (300, 257, 6, 6), (150, 1000, 15, 7), (60, 1300, 4, 8), (33, 4500, 3, 9), (1, 40, 5, 10), (2, 40, 39, 11),
])
def test_hip_sampler_matches_oracle(n_variants, n_paths, size, seed):
    b = pn.flatten(sampler_panel(n_variants, n_paths, seed))
    want_paths, want_best = orc.sampler_run(b, size)
    h = smp.HaplotypeSampler(b, size)
    assert h.best_scores == want_best.tolist()
    assert np.array_equal(h.sampled, want_paths)
    for v in range(0, n_variants, max(1, n_variants // 7)):
        assert len(set(h.sampled[:, v].tolist())) == size


@pytest.mark.gpu
def test_hip_sampler_penalty_and_rates():
    b = pn.flatten(sampler_panel(300, 90, 21))
    for pen, rate, N in ((0, 1.26, 25000.0), (10, 0.01, 25000.0), (25, 5.0, 1.0), (65530, 1.26, 25000.0), (3, 1.26, 1e-9)):
        want_paths, want_best = orc.sampler_run(b, 6, rate, N, pen)
        h = smp.HaplotypeSampler(b, 6, rate, N, allele_penalty=pen)
        assert h.best_scores == want_best.tolist(), (pen, rate, N)
        assert np.array_equal(h.sampled, want_paths), (pen, rate, N)


@pytest.mark.gpu
def test_hip_sampler_saturation():
    """Coincident positions make the recombination cost saturate (undefined in the reference, pinned to
    UINT_MAX in oracle and product): saturating adds, UINT_MAX entries never chosen as minima."""
    uks = sampler_panel(40, 12, 31)
    for v in (10, 11, 25):
        uks[v].variant_pos = uks[v - 1].variant_pos
    b = pn.flatten(uks)
    want_paths, want_best = orc.sampler_run(b, 5)
    h = smp.HaplotypeSampler(b, 5)
    assert h.best_scores == want_best.tolist()
    assert np.array_equal(h.sampled, want_paths)


@pytest.mark.gpu
def test_hip_sampler_batch_of_contigs():
    """pg_sampler_run_batch: contigs of different lengths (and one without variants) in one call — one workgroup per
    contig and pass — give what the oracle gives for each of them alone."""
    batches = [pn.flatten(sampler_panel(v, 90, 40 + i)) for i, v in enumerate((130, 17, 400, 1, 64))]
    empty = pn.flatten(sampler_panel(3, 90, 99)).slice(0, 0)
    batches.insert(2, empty)
    sampled, best = smp.sample_contigs(batches, 7)
    for b, s, bs in zip(batches, sampled, best):
        if b.n_variants == 0:
            assert s.shape == (7, 0)
            continue
        want_paths, want_best = orc.sampler_run(b, 7)
        assert np.array_equal(s, want_paths) and bs.tolist() == want_best.tolist()


@pytest.mark.gpu
def test_hip_sampler_errors():
    b = pn.flatten(sampler_panel(10, 5, 1))
    with pytest.raises(RuntimeError):
        smp.HaplotypeSampler(b, 5)  # as many passes as paths: every path is masked in the last pass
    one = pn.flatten([pn.BiallelicUniqueKmers(10, [0])])
    with pytest.raises(RuntimeError):
        smp.HaplotypeSampler(one, 1)


@pytest.mark.gpu
def test_sampler_then_genotyping_matches_oracle():
    """The sampler's output panel feeds the genotyping HMM (reference src/commands.cpp:148-151 then
    :160-175): HIP sampler + HIP HMM against oracle sampler + oracle HMM."""
    from pangenie_amd import hmm
    from tests.parity_util import assert_parity
    uks = sampler_panel(120, 150, 77, max_alleles=3)
    b = pn.flatten(uks)
    h = smp.HaplotypeSampler(b, 15)
    want_paths, _ = orc.sampler_run(b, 15)
    assert np.array_equal(h.sampled, want_paths)
    sub = h.panel
    assert sub.n_paths == 15
    args = (6, 108, 54, 0.0)
    res = hmm.genotype_contig(sub, hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 25000.0))
    want = orc.genotype_contig(sub, orc.OracleTable(*args), orc.make_params(1.26, False, 25000.0))
    assert_parity(sub, res, want)


# --------------------------------------------------------------------------- #
#  the reference's own 215-path fixture (tests/data/region_UniqueKmersList.cereal, kept as data under tests/golden)
# --------------------------------------------------------------------------- #
def _fixture_panel():
    from pangenie_amd import cereal_io
    return cereal_io.load(Path(__file__).parent / "golden" / "region_UniqueKmersList.cereal").unique_kmers["chr1"]


def test_oracle_sampler_on_the_reference_fixture():
    """215 paths, 44 / 45 alleles per record, most of them undefined (cost 50): the sampler's real input shape
    (the reference enables sampling above 100 paths, src/commands.cpp:283-287)."""
    uks = _fixture_panel()
    b = pn.flatten(uks)
    assert b.n_paths == 215 and b.n_variants == 2
    cost = orc.sampler_emission_costs(b)
    assert set(np.unique(cost).tolist()) <= {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 25, 50} and (cost == 50).sum() >= 80
    sampled, best = orc.sampler_run(b, 15)
    for v in range(2):
        assert len(set(sampled[:, v].tolist())) == 15
    assert (np.diff(best.astype(np.int64)) >= 0).all()  # every pass is at best as good as the one before
    # the object mirror and the flat restatement of update_paths agree on the reduced panel
    flat = b.update_paths(np.vstack([sampled, np.zeros((1, 2), np.uint32)]))
    for v, uk in enumerate(uks):
        uk.update_paths(sampled[:, v].tolist() + [0])
    want = pn.flatten(uks)
    for f in ("kmer_off", "kmer_count", "allele_off", "allele_id", "allele_flags", "allele_kmer_off", "allele_kmer_mask", "path_allele"):
        assert np.array_equal(getattr(flat, f), getattr(want, f)), f


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["fast", "general"])
def test_hip_sampler_and_hmm_on_the_reference_fixture(kernel, monkeypatch):
    """Index archive -> sampler (15 paths + reference) -> HMM -> VCF sample column, HIP against oracle, on
    reference-held data."""
    from pangenie_amd import hmm
    from pangenie_amd.genotyping_result import results_from_flat, vcf_sample_field
    from tests.parity_util import assert_parity
    if kernel == "general":
        monkeypatch.setenv("PG_SAMPLER_KERNEL", "general")
    uks = _fixture_panel()
    b = pn.flatten(uks)
    want_paths, want_best = orc.sampler_run(b, 15)
    h = smp.HaplotypeSampler(b, 15, add_reference=True)
    assert h.kernel == (0 if kernel == "general" else 1)
    assert h.best_scores == want_best.tolist()
    assert np.array_equal(h.sampled[:15], want_paths) and (h.sampled[15] == 0).all()
    sub = h.panel
    assert sub.n_paths == 16
    table, params = (18 // 4, 18 * 4, 2 * 18, 0.01), (1.26, False, 1e-5)  # tests/CommandsTest.cpp:20-35
    res = hmm.genotype_contig(sub, hmm.ProbabilityTable(*table), hmm.make_params(*params))
    ref = orc.genotype_contig(sub, orc.OracleTable(*table), orc.make_params(*params))
    assert_parity(sub, res, ref)
    got = res.genotyping_results()
    want = results_from_flat(sub, ref.lik, ref.kept, ref.allele_present, ref.n_kmers, ref.coverage)
    for g, w, d, na in zip(got, want, ([0, 1], [0, 1, 2]), (44, 45)):
        g.normalize(); w.normalize()
        assert vcf_sample_field(g, d, na) == vcf_sample_field(w, d, na)


# --------------------------------------------------------------------------- #
#  sampler -> update_paths -> job with the panel staying on the device (pg_sampler_then_job)
# --------------------------------------------------------------------------- #
PANEL_FIELDS = ("kmer_off", "kmer_count", "allele_off", "allele_id", "allele_flags", "allele_kmer_off", "allele_kmer_mask", "path_allele")


@pytest.mark.gpu
@pytest.mark.parametrize("add_reference", [False, True])
def test_sampler_then_job_panel_is_the_hosts_update_paths_bit_for_bit(add_reference):
    """The reduced panel the device forms (ku_count / ku_write) must be, array for array, what update_paths gives on the
    host (pangenie_amd.panel.ContigBatch.update_paths, itself checked against the object mirror of the reference's
    update_paths above) — on the reference's 215-path fixture and on seeded multiallelic panels, several contigs in one
    call (one of them without variants); the job's posteriors are then those of a job made from the host's panel, bit
    for bit, and match the oracle's sampler -> HMM."""
    from pangenie_amd import hmm
    from tests.parity_util import assert_parity
    fixture = pn.flatten(_fixture_panel())
    panels = [fixture, pn.flatten(sampler_panel(150, 120, 5, max_alleles=4)), pn.flatten(sampler_panel(90, 64, 6, max_alleles=2)),
              pn.flatten(sampler_panel(40, 33, 7, max_alleles=6)).slice(0, 0), pn.flatten(sampler_panel(300, 215, 8, max_alleles=3))]
    for b in panels[1:]:
        b.kmer_count[::3] = 1
    size = 12
    table, params = (18 // 4, 18 * 4, 2 * 18, 0.01), (1.26, False, 1e-5)
    t, p = hmm.ProbabilityTable(*table), hmm.make_params(*params)
    job, sampled, best = smp.sample_then_job(panels, size, t, p, add_reference=add_reference)
    job.run()
    got = job.fetch_all()
    for g, b in enumerate(panels):
        if b.n_variants == 0:
            assert job.batches[g].n_variants == 0
            continue
        want_paths, want_best = orc.sampler_run(b, size)
        assert np.array_equal(sampled[g], want_paths) and best[g].tolist() == want_best.tolist()
        rows = np.vstack([want_paths, np.zeros((1, b.n_variants), np.uint32)]) if add_reference else want_paths
        host = b.update_paths(rows)
        dev = job.batches[g]
        assert dev.n_paths == host.n_paths == size + int(add_reference)
        for f in PANEL_FIELDS:
            assert np.array_equal(getattr(dev, f), getattr(host, f)), (g, f)
        alone = hmm.genotype_contig(host, t, p)
        assert np.array_equal(got[g].lik, alone.lik) and np.array_equal(got[g].lik_exp, alone.lik_exp)
        assert np.array_equal(got[g].kept, alone.kept) and got[g].n_columns == alone.n_columns
        assert np.array_equal(got[g].coverage, alone.coverage) and np.array_equal(got[g].n_kmers, alone.n_kmers)
        ref = orc.genotype_contig(host, orc.OracleTable(*table), orc.make_params(*params))
        assert_parity(host, alone, ref)
    job.close()


@pytest.mark.gpu
def test_sampler_then_job_limits_and_errors():
    from pangenie_amd import hmm
    t = hmm.ProbabilityTable(6, 108, 54, 0.01)
    b = pn.flatten(sampler_panel(30, 20, 3, max_alleles=2))
    with pytest.raises(hmm.PanGenieError):
        smp.sample_then_job([b], 20, t)          # more passes than paths
    with pytest.raises(hmm.PanGenieError):
        smp.sample_then_job([b], 0, t)           # no pass at all
    job, _, _ = smp.sample_then_job([b], 5, t, want_paths=False)
    job.run()
    assert job.fetch(0).n_columns > 0
    job.close()
