"""The drop-in the way the reference drives it: one HMM constructor per (contig x subset) on N thread-pool workers at a
time, all sharing one ProbabilityTable (reference src/commands.cpp:949-978, run_genotyping :155-185).  Here: 24 host
threads call pg_hmm_genotype_contig at once — mixed shapes (16 / 30 / 64 / 128 paths, multiallelic and biallelic,
one phasing call, one malformed batch) — and every result is checked against the CPU oracle.  Concurrent calls are
merged into one device job by the library (include/pangenie_hmm.h); each caller must get exactly what it would have
got alone, and an error must stay with its caller."""
import numpy as np
import pytest

from pangenie_amd import hmm
from pangenie_amd.panel import default_table_args, synthetic_panel
from tests.parity_util import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import pyoracle
    return pyoracle


def mixed_batches():
    shapes = [(900, 64, 0.0), (400, 16, 0.0), (300, 128, 0.2), (700, 64, 0.0), (250, 30, 0.3), (1200, 16, 0.1),
              (500, 64, 0.0), (350, 64, 0.25), (150, 128, 0.0), (800, 16, 0.0), (450, 64, 0.0), (200, 5, 0.0),
              (650, 64, 0.0), (300, 32, 0.2), (1000, 64, 0.0), (120, 128, 0.2), (550, 16, 0.0), (380, 64, 0.0),
              (270, 64, 0.0), (600, 48, 0.1), (330, 64, 0.0), (90, 2, 0.0), (1, 64, 0.0), (0, 64, 0.0)]
    out = [synthetic_panel(max(V, 1), H, 20, seed=500 + i, multiallelic_frac=m) for i, (V, H, m) in enumerate(shapes)]
    out[-1] = out[-1].slice(0, 0)  # a contig without variants
    return out


def test_24_threads_mixed_shapes_vs_oracle(orc):
    batches = mixed_batches()
    args = default_table_args()
    table = hmm.ProbabilityTable(*args)
    otable = orc.OracleTable(*args)
    prm = hmm.make_params(1.26, False, 1e-5)
    before = hmm.coalesce_stats()
    for rounds in range(2):  # (second round: arenas come from the pool)
        got = hmm.genotype_contigs_threaded(batches, table, prm)
        for b, r in zip(batches, got):
            assert not isinstance(r, Exception), r
            ref = orc.genotype_contig(b, otable, orc.make_params(1.26, False, 1e-5))
            assert_parity(b, r, ref)
    after = hmm.coalesce_stats()
    assert after["calls"] - before["calls"] == 2 * len(batches)
    # 24 calls in flight together: far fewer device jobs than calls
    assert after["merged_jobs"] - before["merged_jobs"] <= len(batches), after
    assert after["largest_merge"] >= 2, after


def test_threads_equal_single_calls_bit_for_bit():
    """chains of a merged job are independent: a caller gets the very bits it gets alone"""
    batches = [synthetic_panel(600 + 37 * i, 64, 20, seed=40 + i) for i in range(6)] + \
              [synthetic_panel(500, 16, 20, seed=90, multiallelic_frac=0.2)]
    table = hmm.ProbabilityTable(*default_table_args())
    prm = hmm.make_params(1.26, False, 1e-5)
    alone = [hmm.genotype_contig(b, table, prm) for b in batches]
    together = hmm.genotype_contigs_threaded(batches, table, prm)
    for a, t in zip(alone, together):
        assert not isinstance(t, Exception), t
        assert np.array_equal(a.lik, t.lik) and np.array_equal(a.lik_exp, t.lik_exp)
        assert np.array_equal(a.kept, t.kept) and a.n_columns == t.n_columns


def test_threads_mixed_parameters_phasing_and_an_error_stay_with_their_callers(orc):
    args = default_table_args()
    table = hmm.ProbabilityTable(*args)
    otable = orc.OracleTable(*args)
    batches = [synthetic_panel(400 + 50 * i, 64 if i % 2 else 16, 20, seed=700 + i) for i in range(10)]
    phase = synthetic_panel(300, 30, 20, seed=77, multiallelic_frac=0.2)
    bad = synthetic_panel(200, 16, 20, seed=78)
    bad.path_allele = bad.path_allele.copy()
    bad._c = None
    bad.path_allele[5 * 16 + 3] = 999  # an allele id the variant does not have: PG_ERR_INVALID, for THIS caller only
    other = synthetic_panel(350, 64, 20, seed=79)  # different transition parameters: never merged with the rest
    all_b = batches + [phase, bad, other]
    p_std = hmm.make_params(1.26, False, 1e-5)
    p_phase = hmm.make_params(1.26, False, 1e-5, run_genotyping=False, run_phasing=True)
    p_other = hmm.make_params(2.0, False, 25000.0)
    plist = [p_std] * len(batches) + [p_phase, p_std, p_other]
    got = hmm.genotype_contigs_threaded(all_b, table, plist)
    for b, r in zip(batches, got):
        assert not isinstance(r, Exception), r
        assert_parity(b, r, orc.genotype_contig(b, otable, orc.make_params(1.26, False, 1e-5)))
    r_phase, r_bad, r_other = got[len(batches):]
    assert not isinstance(r_phase, Exception), r_phase
    want = orc.viterbi_contig(phase, otable, orc.make_params(1.26, False, 1e-5, run_genotyping=False, run_phasing=True), form=1)
    assert np.array_equal(r_phase.haplotype_1, want.hap1) and np.array_equal(r_phase.haplotype_2, want.hap2)
    assert isinstance(r_bad, hmm.PanGenieError) and r_bad.code == -1, r_bad
    assert not isinstance(r_other, Exception), r_other
    assert_parity(other, r_other, orc.genotype_contig(other, otable, orc.make_params(2.0, False, 25000.0)))


def test_fewer_workers_than_contigs(orc):
    """-t smaller than the number of contigs (src/commands.cpp:949-953): workers come back for more"""
    args = default_table_args()
    table = hmm.ProbabilityTable(*args)
    otable = orc.OracleTable(*args)
    batches = [synthetic_panel(300 + 40 * i, 64, 20, seed=900 + i) for i in range(9)]
    got = hmm.genotype_contigs_threaded(batches, table, hmm.make_params(1.26, False, 1e-5), n_threads=4)
    for b, r in zip(batches, got):
        assert not isinstance(r, Exception), r
        assert_parity(b, r, orc.genotype_contig(b, otable, orc.make_params(1.26, False, 1e-5)))


def test_an_announced_call_that_never_arrives_costs_the_wait_bound_once(orc):
    """pg_hmm_announce promises a call; a worker that announces and then never calls (it threw while flattening its
    UniqueKmers, or an external caller forgot) makes the first leader wait for it — bounded by PG_COALESCE_WAIT_MS (250
    ms), after which the batch runs without it.  Results are unaffected; the cost is measured and printed.  Un-announced
    callers next to announced ones (fewer announcements than callers) simply join the batch in flight."""
    import threading
    import time
    lib = hmm._lib.load_hip()
    batches = mixed_batches()[:6]
    args = default_table_args()
    table, prm = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    solo = [hmm.genotype_contig(b, table, prm) for b in batches]

    def round_of_calls(n_announced):
        out = [None] * len(batches)
        for _ in range(n_announced):
            hmm.announce(0)

        def work(i):
            out[i] = hmm.genotype_contig(batches[i], table, prm, announced=i < n_announced)
        ts = [threading.Thread(target=work, args=(i,)) for i in range(len(batches))]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return out, time.perf_counter() - t0

    round_of_calls(len(batches))                       # (warm: arenas into the pool)
    ok, t_all = round_of_calls(len(batches))           # every caller announced
    half, t_half = round_of_calls(len(batches) // 2)   # fewer announcements than callers
    hmm.announce(0)                                    # one announcement nobody honours ...
    late, t_late = round_of_calls(len(batches))
    lib.pg_hmm_retract(0)                              # (... taken back, whatever the leader did with it)
    for got in (ok, half, late):
        for g, s in zip(got, solo):
            assert np.array_equal(g.lik, s.lik) and np.array_equal(g.lik_exp, s.lik_exp) and g.n_columns == s.n_columns
    print("one-shot round of %d calls: all announced %.1f ms, half announced %.1f ms, with a call announced but never made %.1f ms"
          % (len(batches), t_all * 1e3, t_half * 1e3, t_late * 1e3))
    assert t_late < t_all + 0.6      # bounded: the 250 ms wait once, not per caller
    assert t_half < t_all + 0.2      # un-announced callers cost nothing extra


def test_merged_one_shot_job_pipelines_its_upload_behind_the_long_chains(orc, monkeypatch):
    """Round 6: a merged one-shot job (upload and run inside one call) uploads the inputs of its LONG chains first, starts their
    preparation and phase 1, and lets the other chains' inputs cross PCIe meanwhile; their index pass, preparation and phase 1
    then run on the job's second stream (pg_shim.cpp: pg_job::pipeline).  PG_PIPELINE_MIN_MB=0 puts these small contigs through
    it (by default only jobs with >= 128 MB of inputs take it); the callers get the bits they get with PG_NO_PIPELINE=1 and
    alone, and match the oracle."""
    args = default_table_args()
    table, otable = hmm.ProbabilityTable(*args), orc.OracleTable(*args)
    prm = hmm.make_params(1.26, False, 1e-5)
    # two long chains (group A), five shorter ones (group B), one of them with multiallelic objects (k_sweep_leanx beside k_sweep_lean)
    batches = [synthetic_panel(v, 64, 20, seed=2100 + i, multiallelic_frac=0.2 if i == 4 else 0.0) for i, v in enumerate((1000, 980, 500, 400, 300, 450, 350))]
    monkeypatch.setenv("PG_PIPELINE_MIN_MB", "0")
    piped = hmm.genotype_contigs_threaded(batches, table, prm)
    again = hmm.genotype_contigs_threaded(batches, table, prm)
    monkeypatch.setenv("PG_NO_PIPELINE", "1")
    plain = hmm.genotype_contigs_threaded(batches, table, prm)
    monkeypatch.delenv("PG_NO_PIPELINE")
    monkeypatch.delenv("PG_PIPELINE_MIN_MB")
    for b, r, r2, q in zip(batches, piped, again, plain):
        assert not isinstance(r, Exception), r
        assert not isinstance(r2, Exception) and not isinstance(q, Exception)
        assert np.array_equal(r.lik, q.lik) and np.array_equal(r.lik_exp, q.lik_exp) and r.n_columns == q.n_columns
        assert np.array_equal(r.lik, r2.lik) and np.array_equal(r.lik_exp, r2.lik_exp)
        assert np.array_equal(r.kept, q.kept)
        assert_parity(b, r, orc.genotype_contig(b, otable, prm if False else orc.make_params(1.26, False, 1e-5)))


def test_upload_run_in_one_call_equals_upload_then_run(orc, monkeypatch):
    """pg_job_upload_run on a resident job: the same bits as pg_job_upload + pg_job_run, pipelined (PG_PIPELINE_MIN_MB=0 lets
    these small contigs take the pipelined path) or not, twice in a row."""
    args = default_table_args()
    table = hmm.ProbabilityTable(*args)
    prm = hmm.make_params(1.26, False, 1e-5)
    batches = [synthetic_panel(v, 64, 20, seed=2300 + i) for i, v in enumerate((900, 890, 300, 420, 333, 10, 505))]
    monkeypatch.setenv("PG_PIPELINE_MIN_MB", "0")
    job = hmm.Job(batches, table, prm)
    assert job.sweep_mode()[0] == "chunked"
    job.run()
    first = job.fetch_all()
    for _ in range(2):
        job.upload_run()
        piped = job.fetch_all()
        for a, b in zip(first, piped):
            assert np.array_equal(a.lik, b.lik) and np.array_equal(a.lik_exp, b.lik_exp) and a.n_columns == b.n_columns
    job.upload()
    job.run()
    for a, b in zip(first, job.fetch_all()):
        assert np.array_equal(a.lik, b.lik) and np.array_equal(a.lik_exp, b.lik_exp)
    job.close()
