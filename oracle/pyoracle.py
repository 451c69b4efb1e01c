"""TEST INFRASTRUCTURE ONLY — ctypes wrapper of oracle/pg_oracle.c.

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Never imported by the pangenie_amd package.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from pangenie_amd._lib import (PgContigBatch, PgHmmParams, i32p, ldp, u8p, u16p, u32p, u64p)

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "_build" / "libpg_oracle.so"


class PgoResult(C.Structure):
    _fields_ = [
        ("lik", ldp),
        ("kept", u8p),
        ("allele_present", u8p),
        ("n_kmers", u16p),
        ("coverage", u16p),
        ("n_columns", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


_lib = None


def build():
    subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)


def load():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        lib = C.CDLL(str(LIB_PATH))
        ld = C.c_longdouble
        lib.pgo_table_create.argtypes = [C.c_uint16, C.c_uint16, C.c_uint16, ld]
        lib.pgo_table_create.restype = C.c_void_p
        lib.pgo_table_create_default.argtypes = []
        lib.pgo_table_create_default.restype = C.c_void_p
        lib.pgo_table_modify.argtypes = [C.c_void_p, C.c_uint16, C.c_uint16, ld, ld, ld]
        lib.pgo_table_modify.restype = C.c_int
        lib.pgo_table_get.argtypes = [C.c_void_p, C.c_uint16, C.c_uint16, ldp]
        lib.pgo_table_get.restype = None
        lib.pgo_table_destroy.argtypes = [C.c_void_p]
        lib.pgo_table_destroy.restype = None
        lib.pgo_copynumber_regularized.argtypes = [ld, ld, ld, ld, ldp]
        lib.pgo_copynumber_regularized.restype = None
        lib.pgo_transition_probs.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_uint32,
                                             C.c_int, ld, ldp]
        lib.pgo_transition_probs.restype = None
        lib.pgo_emission_table.argtypes = [C.POINTER(PgContigBatch), C.c_void_p, C.c_uint32,
                                           ldp, i32p]
        lib.pgo_emission_table.restype = C.c_int
        lib.pgo_genotype_contig.argtypes = [C.POINTER(PgContigBatch), C.c_void_p,
                                            C.POINTER(PgHmmParams), C.POINTER(PgoResult)]
        lib.pgo_genotype_contig.restype = C.c_int
        lib.pgo_viterbi_contig.argtypes = [C.POINTER(PgContigBatch), C.c_void_p, C.POINTER(PgHmmParams), C.c_int,
                                           u16p, u16p, u8p, u32p, u16p, u16p]
        lib.pgo_viterbi_contig.restype = C.c_int
        lib.pgo_geno_offsets.argtypes = [C.POINTER(PgContigBatch), u64p]
        lib.pgo_geno_offsets.restype = None
        lib.pgo_sampler_emission_costs.argtypes = [C.POINTER(PgContigBatch), u16p]
        lib.pgo_sampler_emission_costs.restype = None
        lib.pgo_sampler_transition_cost.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_uint32, ld]
        lib.pgo_sampler_transition_cost.restype = C.c_uint32
        lib.pgo_sampler_column_minima.argtypes = [u32p, u8p, C.c_uint32, u32p]
        lib.pgo_sampler_column_minima.restype = None
        lib.pgo_sampler_run.argtypes = [C.POINTER(PgContigBatch), C.c_uint32, C.c_double, ld, C.c_uint16, u32p, u32p]
        lib.pgo_sampler_run.restype = C.c_int
        _lib = lib
    return _lib


def _ld3():
    """(array, pointer) for a `long double out3[3]` argument — the numpy array keeps all 64 mantissa bits (an element
    of a ctypes long double array reads back as a Python float, i.e. a double)."""
    a = np.zeros(3, dtype=np.longdouble)
    return a, a.ctypes.data_as(ldp)


def _ld(x) -> C.c_longdouble:
    """Python float / str / np.longdouble -> c_longdouble without a double round-trip (ctypes' own conversion of a
    np.longdouble goes through a Python float)."""
    return C.c_longdouble.from_buffer_copy(np.array([np.longdouble(x)], dtype=np.longdouble).tobytes())


class OracleTable:
    """ProbabilityTable restatement (reference src/probabilitytable.cpp)."""

    def __init__(self, cov_min=0, cov_max=0, count_max=0, regularization=0.0, default=False):
        lib = load()
        if default:
            self.h = lib.pgo_table_create_default()
        else:
            self.h = lib.pgo_table_create(cov_min, cov_max, count_max, _ld(regularization))
        self.lib = lib

    def modify(self, cov, count, p0, p1, p2):
        rc = self.lib.pgo_table_modify(self.h, cov, count, _ld(p0), _ld(p1), _ld(p2))
        if rc:
            raise RuntimeError("ProbabilityTable::modify_probability: no precomputed values for these parameters.")

    def get(self, cov, count) -> np.ndarray:
        out, ptr = _ld3()
        self.lib.pgo_table_get(self.h, cov, count, ptr)
        return out

    def __del__(self):
        try:
            self.lib.pgo_table_destroy(self.h)
        except Exception:
            pass


def copynumber_regularized(cn0, cn1, cn2, reg) -> np.ndarray:
    out, ptr = _ld3()
    load().pgo_copynumber_regularized(_ld(cn0), _ld(cn1), _ld(cn2), _ld(reg), ptr)
    return out


def transition_probs(from_pos, to_pos, recombrate, nr_paths, uniform, effective_N) -> np.ndarray:
    out, ptr = _ld3()
    load().pgo_transition_probs(from_pos, to_pos, recombrate, nr_paths, int(uniform), _ld(effective_N), ptr)
    return out


def emission_table(batch, table: OracleTable, v: int):
    A = int(batch.allele_off[v + 1] - batch.allele_off[v])
    out = np.zeros(A * A, dtype=np.longdouble)
    az = C.c_int32(0)
    rc = load().pgo_emission_table(C.byref(batch.as_c()), table.h, v,
                                   out.ctypes.data_as(ldp), C.byref(az))
    assert rc == 0
    return out.reshape(A, A), bool(az.value)


def make_params(recombrate=1.26, uniform=False, effective_N=25000.0, run_genotyping=True,
                run_phasing=False) -> PgHmmParams:
    p = PgHmmParams()
    p.effective_N = _ld(effective_N)
    p.recombrate = float(recombrate)
    p.uniform = int(uniform)
    p.run_genotyping = int(run_genotyping)
    p.run_phasing = int(run_phasing)
    return p


class OracleResult:
    def __init__(self, batch):
        V = batch.n_variants
        self.geno_off = batch.geno_off
        self.lik = np.zeros(int(self.geno_off[-1]), dtype=np.longdouble)
        self.kept = np.zeros(V, np.uint8)
        self.allele_present = np.zeros(int(batch.allele_off[-1]) if V else 0, np.uint8)
        self.n_kmers = np.zeros(V, np.uint16)
        self.coverage = np.zeros(V, np.uint16)
        self.n_columns = 0


def genotype_contig(batch, table: OracleTable, params: PgHmmParams) -> OracleResult:
    """HMM::HMM(..., normalize=false) restated on the CPU in long double."""
    r = OracleResult(batch)

    def p(a, t):
        if a.size == 0:
            a = np.zeros(1, a.dtype)
        return a, a.ctypes.data_as(t)
    keep = []
    c = PgoResult()
    for name, typ in (("lik", ldp), ("kept", u8p), ("allele_present", u8p),
                      ("n_kmers", u16p), ("coverage", u16p)):
        arr, ptr = p(getattr(r, name), typ)
        keep.append(arr)
        setattr(c, name, ptr)
    rc = load().pgo_genotype_contig(C.byref(batch.as_c()), table.h, C.byref(params), C.byref(c))
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    r.n_columns = int(c.n_columns)
    return r


class OraclePhasing:
    """hap1 / hap2 [V]: Viterbi haplotype alleles at kept variants (0 elsewhere)."""

    def __init__(self, V):
        n = max(V, 1)
        self.hap1 = np.zeros(n, np.uint16)
        self.hap2 = np.zeros(n, np.uint16)
        self.kept = np.zeros(n, np.uint8)
        self.n_kmers = np.zeros(n, np.uint16)
        self.coverage = np.zeros(n, np.uint16)
        self.n_columns = 0


def viterbi_contig(batch, table: OracleTable, params: PgHmmParams, form: int = 0) -> OraclePhasing:
    """HMM::compute_viterbi_path restated in long double (reference src/hmm.cpp:112-173, :408-511);
    form 0 = the reference's O(H^4) loop, form 1 = the same maxima in O(H^2)."""
    V = batch.n_variants
    r = OraclePhasing(V)
    nc = np.zeros(1, np.uint32)
    rc = load().pgo_viterbi_contig(C.byref(batch.as_c()), table.h, C.byref(params), int(form),
                                   r.hap1.ctypes.data_as(u16p), r.hap2.ctypes.data_as(u16p), r.kept.ctypes.data_as(u8p),
                                   nc.ctypes.data_as(u32p), r.n_kmers.ctypes.data_as(u16p), r.coverage.ctypes.data_as(u16p))
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    r.n_columns = int(nc[0])
    for name in ("hap1", "hap2", "kept", "n_kmers", "coverage"):
        setattr(r, name, getattr(r, name)[:V])
    return r


# --------------------------------------------------------------------------- #
#  HaplotypeSampler restatement (pg_sampler_oracle.c)
# --------------------------------------------------------------------------- #
def sampler_emission_costs(batch) -> np.ndarray:
    n = int(batch.allele_off[-1])
    out = np.zeros(max(1, n), np.uint16)
    load().pgo_sampler_emission_costs(C.byref(batch.as_c()), out.ctypes.data_as(u16p))
    return out[:n]


def sampler_transition_cost(from_pos, to_pos, recombrate, nr_paths, effective_N=25000.0) -> int:
    return int(load().pgo_sampler_transition_cost(int(from_pos), int(to_pos), float(recombrate), int(nr_paths), _ld(effective_N)))


def sampler_column_minima(column, mask):
    col = np.ascontiguousarray(column, np.uint32)
    m = np.ascontiguousarray(mask, np.uint8)
    out = np.zeros(4, np.uint32)
    load().pgo_sampler_column_minima(col.ctypes.data_as(u32p), m.ctypes.data_as(u8p), col.size, out.ctypes.data_as(u32p))
    return tuple(int(x) for x in out)


def sampler_run(batch, size, recombrate=1.26, effective_N=25000.0, allele_penalty=10):
    """-> (sampled_paths [size, V] u32, best_scores [size] u32)"""
    V = batch.n_variants
    sampled = np.zeros((size, max(V, 1)), np.uint32)
    best = np.zeros(max(size, 1), np.uint32)
    rc = load().pgo_sampler_run(C.byref(batch.as_c()), size, float(recombrate), _ld(effective_N), int(allele_penalty),
                                sampled.ctypes.data_as(u32p), best.ctypes.data_as(u32p))
    if rc:
        raise RuntimeError(f"oracle sampler error {rc}")
    return sampled[:, :V], best[:size]


_ref_transitions = None


def ref_transition_cost(from_pos, to_pos, recombrate, nr_paths, effective_N=25000.0):
    """The reference's OWN SamplingTransitions translation unit (oracle/_ref/libref_transitions.so, built by
    `make ref` where /root/reference exists).  Returns None when the library is not there."""
    global _ref_transitions
    path = HERE / "_ref" / "libref_transitions.so"
    if _ref_transitions is None:
        if not path.exists():
            return None
        lib = C.CDLL(str(path))
        lib.ref_sampling_transition_cost.argtypes = [C.c_ulonglong, C.c_ulonglong, C.c_double, C.c_ushort, C.c_longdouble]
        lib.ref_sampling_transition_cost.restype = C.c_uint
        _ref_transitions = lib
    return int(_ref_transitions.ref_sampling_transition_cost(int(from_pos), int(to_pos), float(recombrate), int(nr_paths), _ld(effective_N)))


# --------------------------------------------------------------------------- #
#  oracle/_ref/libref_table.so: the reference's OWN ProbabilityTable + CopyNumber translation units
#  (built by `make -C oracle ref` from /root/reference/src where they lie); None where it was never built
# --------------------------------------------------------------------------- #
_ref_table = None


def ref_table_lib():
    global _ref_table
    path = HERE / "_ref" / "libref_table.so"
    if _ref_table is None:
        if not path.exists():
            return None
        lib = C.CDLL(str(path))
        ld = C.c_longdouble
        lib.ref_table_create.argtypes = [C.c_uint16, C.c_uint16, C.c_uint16, ld]
        lib.ref_table_create.restype = C.c_void_p
        lib.ref_table_create_default.argtypes = []
        lib.ref_table_create_default.restype = C.c_void_p
        lib.ref_table_destroy.argtypes = [C.c_void_p]
        lib.ref_table_destroy.restype = None
        lib.ref_table_get.argtypes = [C.c_void_p, C.c_uint16, C.c_uint16, ldp]
        lib.ref_table_get.restype = None
        lib.ref_copynumber_regularized.argtypes = [ld, ld, ld, ld, ldp]
        lib.ref_copynumber_regularized.restype = None
        _ref_table = lib
    return _ref_table


class RefTable:
    """ProbabilityTable of the reference's own compiled code (reference src/probabilitytable.cpp)."""

    def __init__(self, cov_min=0, cov_max=0, count_max=0, regularization=0.0, default=False):
        self.lib = ref_table_lib()
        self.h = self.lib.ref_table_create_default() if default else self.lib.ref_table_create(cov_min, cov_max, count_max, _ld(regularization))

    def get(self, cov, count) -> np.ndarray:
        out, ptr = _ld3()
        self.lib.ref_table_get(self.h, cov, count, ptr)
        return out

    def __del__(self):
        try:
            if self.h:
                self.lib.ref_table_destroy(self.h)
                self.h = None
        except Exception:
            pass


def ref_copynumber_regularized(cn0, cn1, cn2, reg) -> np.ndarray:
    out, ptr = _ld3()
    ref_table_lib().ref_copynumber_regularized(_ld(cn0), _ld(cn1), _ld(cn2), _ld(reg), ptr)
    return out
