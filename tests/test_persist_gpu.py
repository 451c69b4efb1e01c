"""The persistent chunked phase 2 (round 6, opt-in: PG_KERNELS=persist): chunked jobs whose chains are all lean chains (64 paths,
biallelic) run the second half of every half-chain as ONE launch of k_sweep_lean<4> beside ONE launch of k_post_loop, chunks
handed over through DevContig::sync (pangenie_amd/csrc/pg_kernels.hip: chunk_spin / chunk_publish).  The default keeps a launch
per chunk (k_sweep_lean<3> + k_post): measured at par or ahead on the whole-genome job (profiles/r06_persist.txt).  Both must match the oracle; where no column falls back to uniform they carry the
same bits (a column resumed from memory and one carried in registers hand the same partial sums to the next step)."""
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


def _rel(a, c):
    a, c = a.likelihoods_ld(), c.likelihoods_ld()
    den = np.maximum(np.abs(a), np.abs(c))
    return float(np.where(den > 0, np.abs(a - c) / np.where(den > 0, den, 1), 0).max())


def _job_result(b, t, p, persistent):
    job = hmm.Job([b], t, p)
    assert ("k_post_loop" in job.plan()) == persistent, job.plan()
    job.run()
    r = job.fetch(0)
    job.close()
    return r


@pytest.mark.parametrize("K", [2, 64, 128, 4096])
def test_persistent_phase2_vs_launch_per_chunk_and_oracle(K, orc, monkeypatch):
    """One job, six lean chains of different length (incl. one of three variants) and chunk sizes that put
    several, one or no chunk boundary into a half-chain.  Persistent = launch per chunk bit for bit, both match the oracle; a
    second run of the resident job (the hand-over words are zeroed per run) gives the same bits."""
    monkeypatch.setenv("PG_SWEEP_MODE", "chunked")
    monkeypatch.setenv("PG_CHUNK_COLS", str(K))
    args = default_table_args()
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    batches = [synthetic_panel(v, 64, 20, seed=300 + i) for i, v in enumerate((700, 333, 3, 1290, 64, 515))]
    monkeypatch.setenv("PG_KERNELS", "persist")
    job = hmm.Job(batches, t, p)
    assert "k_sweep_lean<4>" in job.plan() and "k_post_loop" in job.plan(), job.plan()
    job.run()
    first = job.fetch_all()
    job.run()
    again = job.fetch_all()
    assert "persistent phase 2: 2 run(s), 0 more" in job.plan(), job.plan()   # (both runs really took the persistent pair)
    job.close()
    monkeypatch.delenv("PG_KERNELS", raising=False)
    job = hmm.Job(batches, t, p)
    assert "k_sweep_lean<3>" in job.plan() and "k_post_loop" not in job.plan(), job.plan()
    job.run()
    chunks = job.fetch_all()
    job.close()
    for b, r, r2, c in zip(batches, first, again, chunks):
        assert np.array_equal(r.lik, r2.lik) and np.array_equal(r.lik_exp, r2.lik_exp)
        assert np.array_equal(r.lik, c.lik) and np.array_equal(r.lik_exp, c.lik_exp)
        assert_parity(b, r, orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5)))


@pytest.mark.parametrize("K", [2, 6, 64])
def test_persistent_phase2_with_fallback_columns(K, orc, monkeypatch):
    """Unregularised table: forward columns that fall back to uniform and all-zero backward columns on, before and behind chunk
    boundaries.  (A fall-back exactly at a boundary is resumed from the stored uniform column by the launch-per-chunk path and
    taken inside the step by the persistent one: same value, another order of additions — rounding apart.)"""
    monkeypatch.setenv("PG_SWEEP_MODE", "chunked")
    monkeypatch.setenv("PG_CHUNK_COLS", str(K))
    args = (6, 108, 54, 0.0)
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    for seed, V in ((5, 330), (7, 131), (9, 64)):
        b = synthetic_panel(V, 64, 20, seed=seed)
        b.kmer_count[::3] = 0
        b.kmer_count[1::17] = 60000
        monkeypatch.setenv("PG_KERNELS", "persist")
        pers = _job_result(b, t, p, True)
        monkeypatch.delenv("PG_KERNELS", raising=False)
        chunks = _job_result(b, t, p, False)
        ref = orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5))
        assert_parity(b, pers, ref)
        assert_parity(b, chunks, ref)
        assert _rel(pers, chunks) < 1e-11


def test_persistent_phase2_only_for_all_lean_jobs(orc, monkeypatch):
    """A chunked job with a chain that is not a lean chain, or an odd chunk size with more than one chunk, keeps a launch per chunk."""
    monkeypatch.setenv("PG_SWEEP_MODE", "chunked")
    monkeypatch.setenv("PG_CHUNK_COLS", "64")
    monkeypatch.setenv("PG_KERNELS", "persist")
    args = default_table_args()
    t, p = hmm.ProbabilityTable(*args), hmm.make_params(1.26, False, 1e-5)
    mixed = [synthetic_panel(400, 64, 20, seed=1), synthetic_panel(400, 16, 20, seed=2)]
    job = hmm.Job(mixed, t, p)
    assert "k_post_loop" not in job.plan(), job.plan()
    job.run()
    got = job.fetch_all()
    job.close()
    for b, r in zip(mixed, got):
        assert_parity(b, r, orc.genotype_contig(b, orc.OracleTable(*args), orc.make_params(1.26, False, 1e-5)))
    monkeypatch.setenv("PG_CHUNK_COLS", "37")
    job = hmm.Job(mixed[:1], t, p)
    assert "k_post_loop" not in job.plan(), job.plan()
    job.close()
