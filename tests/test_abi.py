"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol
include/pangenie_hmm.h declares; host-side pieces (ProbabilityTable in long double,
geno offsets, argument checking) behave like the reference.  No device compute here."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from pangenie_amd import _lib, build
from pangenie_amd.panel import synthetic_panel

ROOT = Path(__file__).resolve().parent.parent
TOL = 1e-7


@pytest.fixture(scope="module")
def lib():
    build.build_hip()
    return _lib.load_hip()


def test_every_declared_symbol_is_exported(lib):
    header = (ROOT / "include" / "pangenie_hmm.h").read_text()
    declared = set(re.findall(r"\b(pg_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.HIP_ABI_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_sampler_symbols_are_exported_and_host_costs(lib):
    """include/pangenie_sampler.h: every declared symbol is exported; the two host-side cost functions
    (no device involved) agree bit for bit with the oracle."""
    from oracle import pyoracle as orc
    from pangenie_amd import sampler as smp
    header = (ROOT / "include" / "pangenie_sampler.h").read_text()
    declared = set(re.findall(r"\b(pg_sampler_[a-z_]+)\s*\(", header))
    assert declared == set(smp.SAMPLER_ABI_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    b = synthetic_panel(200, 12, 20, seed=9, multiallelic_frac=0.4)
    b.kmer_count[::2] = 1
    assert np.array_equal(smp.emission_costs(b), orc.sampler_emission_costs(b))
    rng = np.random.default_rng(1)
    for _ in range(2000):
        a, d = int(rng.integers(0, 2 ** 28)), int(rng.integers(1, 10 ** 6))
        H = int(rng.integers(2, 3000))
        assert smp.SamplingTransitions(a, a + d, 1.26, H).cost == orc.sampler_transition_cost(a, a + d, 1.26, H)
    # argument errors are raised before any device work
    with pytest.raises(RuntimeError):
        smp.HaplotypeSampler(synthetic_panel(5, 1, 4, seed=1), 1)


def test_struct_layout_matches_header(lib):
    # sizes the C compiler gives the header's structs (x86-64 SysV)
    assert C.sizeof(_lib.PgContigBatch) == 8 + 10 * 8
    assert C.sizeof(_lib.PgHmmParams) == 16 + 8 + 4 * 4 + 8  # long double, double, 4 ints, tail pad to 16
    assert C.sizeof(_lib.PgContigResult) == 6 * 8 + 8 + 2 * 8  # 6 pointers, 2 u32, haplotype_1 / haplotype_2


def test_probability_table_known_answers(lib, golden):
    from pangenie_amd.hmm import ProbabilityTable
    for c in golden["probability_table"]:
        t = ProbabilityTable(*c["args"])
        for cov, count, exp in c["expected"]:
            assert np.allclose(t.get(cov, count).astype(float), exp, rtol=0, atol=TOL)
    t = ProbabilityTable(0, 1, 21, 0.0)
    t.modify(0, 10, 0.1, 0.9, 0.1)
    assert np.allclose(t.get(0, 10).astype(float), [0.1, 0.9, 0.1])
    with pytest.raises(RuntimeError):
        t.modify(3, 10, 0.1, 0.9, 0.1)  # outside the precomputed box, as the reference throws


def test_probability_table_matches_oracle_bitwise(lib):
    from oracle import pyoracle as orc
    from pangenie_amd.hmm import ProbabilityTable
    t, o = ProbabilityTable(6, 108, 54, 0.01), orc.OracleTable(6, 108, 54, 0.01)
    for cov, count in [(6, 0), (27, 13), (27, 53), (107, 20), (3, 5), (27, 200), (500, 60000), (0, 0)]:
        a, b = t.get(cov, count), o.get(cov, count)
        assert all((x == y) or (np.isnan(x) and np.isnan(y)) for x, y in zip(a, b)), (cov, count)


def test_geno_offsets(lib):
    b = synthetic_panel(300, 8, 20, seed=3, multiallelic_frac=0.4)
    out = np.zeros(b.n_variants + 1, np.uint64)
    assert lib.pg_hmm_geno_offsets(C.byref(b.as_c()), out.ctypes.data_as(_lib.u64p)) == 0
    assert (out == b.geno_off).all()


def test_argument_errors_without_device(lib):
    from pangenie_amd import hmm
    b = synthetic_panel(10, 4, 6, seed=1)
    t = hmm.ProbabilityTable(6, 108, 54, 0.01)
    # the device Viterbi takes at most 64 selected paths: refused before any device work
    b65 = synthetic_panel(10, 65, 6, seed=1)
    with pytest.raises(hmm.PanGenieError) as e:
        hmm.genotype_contig(b65, t, hmm.make_params(run_phasing=True))
    assert e.value.code == _lib.PG_ERR_UNSUPPORTED
    # no selected paths -> the reference's ColumnIndexer error
    b0 = synthetic_panel(10, 4, 6, seed=1)
    b0.n_paths = 0
    b0.path_allele = b0.path_allele[:0]
    b0._c = None
    with pytest.raises(hmm.PanGenieError) as e:
        hmm.genotype_contig(b0, t, hmm.make_params())
    assert e.value.code == _lib.PG_ERR_NO_PATHS
    if lib.pg_hmm_device_count() == 0:
        # no CPU fallback: without a GPU the product path must fail loudly
        with pytest.raises(hmm.PanGenieError):
            hmm.genotype_contig(b, t, hmm.make_params())


def test_malformed_batches_are_rejected_on_the_host(lib):
    """offsets are validated (O(V)) before anything reaches the device, with specific codes."""
    from pangenie_amd import hmm
    t = hmm.ProbabilityTable(6, 108, 54, 0.01)

    def code(b):
        b._c = None
        with pytest.raises(hmm.PanGenieError) as e:
            hmm.genotype_contig(b, t, hmm.make_params())
        return e.value.code

    b = synthetic_panel(12, 4, 6, seed=1)
    b.kmer_off[5] = b.kmer_off[7] + 3          # not monotonic: a huge K after unsigned wrap on the device
    assert code(b) == _lib.PG_ERR_INVALID
    b = synthetic_panel(12, 4, 6, seed=1)
    b.allele_off[4] = b.allele_off[3]          # a variant without alleles
    assert code(b) == _lib.PG_ERR_INVALID
    b = synthetic_panel(12, 4, 6, seed=1)
    b.allele_off[0] = 1
    assert code(b) == _lib.PG_ERR_INVALID
    # limits: more than 256 alleles in one object, more than 1024 selected paths
    b = synthetic_panel(8, 4, 6, seed=1, multiallelic_frac=1.0, max_alleles=400)
    if int(np.diff(b.allele_off).max()) > 256:
        assert code(b) == _lib.PG_ERR_UNSUPPORTED
    assert code(synthetic_panel(3, 1100, 4, seed=1)) == _lib.PG_ERR_UNSUPPORTED
