"""ctypes view of include/pangenie_hmm.h (the product C ABI).

The product library is `pangenie_amd/csrc/libpangenie_hmm.so` (HIP, gfx950).  It
is loaded lazily and loading FAILS LOUDLY when it is missing: there is no CPU
fallback anywhere in the package (the CPU oracle lives under oracle/ and is
never imported from here).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import os
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HIP_LIB_PATH = ROOT / "pangenie_amd" / "csrc" / "libpangenie_hmm.so"

PG_OK = 0
PG_ERR_INVALID = -1
PG_ERR_NO_PATHS = -2
PG_ERR_UNSUPPORTED = -3
PG_ERR_DEVICE = -4
PG_ERR_NOMEM = -5
PG_N_KERNEL_CLASSES = 6

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)
ldp = C.POINTER(C.c_longdouble)


def c_ld(x) -> C.c_longdouble:
    """np.longdouble / float / decimal string -> c_longdouble with all 64 mantissa bits (ctypes' own conversion goes
    through a Python float, i.e. a double).  A string is parsed AS a long double: c_ld("0.01") is the reference's
    0.01L (its default sampling_effective_N), which no double holds."""
    v = np.longdouble(x) if isinstance(x, str) else x
    return C.c_longdouble.from_buffer_copy(np.array([v], dtype=np.longdouble).tobytes())


def ld_out(n: int):
    """(array, pointer) for `long double out[n]` arguments: read the array, not a ctypes element (which is a float)."""
    a = np.zeros(n, dtype=np.longdouble)
    return a, a.ctypes.data_as(ldp)


class PgContigBatch(C.Structure):
    _fields_ = [
        ("n_variants", C.c_uint32),
        ("n_paths", C.c_uint32),
        ("variant_pos", u64p),
        ("coverage", u16p),
        ("kmer_off", u32p),
        ("kmer_count", u16p),
        ("allele_off", u32p),
        ("allele_id", u16p),
        ("allele_flags", u8p),
        ("allele_kmer_off", u16p),
        ("allele_kmer_mask", u32p),
        ("path_allele", u16p),
    ]


class PgHmmParams(C.Structure):
    _fields_ = [
        ("effective_N", C.c_longdouble),
        ("recombrate", C.c_double),
        ("uniform", C.c_int32),
        ("run_genotyping", C.c_int32),
        ("run_phasing", C.c_int32),
        ("reserved", C.c_int32),
    ]


class PgContigResult(C.Structure):
    _fields_ = [
        ("lik", f64p),
        ("lik_exp", i32p),
        ("kept", u8p),
        ("allele_present", u8p),
        ("n_kmers", u16p),
        ("coverage", u16p),
        ("n_columns", C.c_uint32),
        ("reserved", C.c_uint32),
        ("haplotype_1", u16p),
        ("haplotype_2", u16p),
    ]


class PgSampleCounts(C.Structure):
    _fields_ = [
        ("kmer_count", C.POINTER(u16p)),
        ("coverage", C.POINTER(u16p)),
    ]


PG_CALL_ANNOUNCED = 1


class HipExtensionMissing(RuntimeError):
    pass


_hip = None


def _bind_hip(lib):
    ld = C.c_longdouble
    lib.pg_hmm_geno_offsets.argtypes = [C.POINTER(PgContigBatch), u64p]
    lib.pg_hmm_geno_offsets.restype = C.c_int
    lib.pg_table_create.argtypes = [C.c_uint16, C.c_uint16, C.c_uint16, ld]
    lib.pg_table_create.restype = C.c_void_p
    lib.pg_table_create_default.argtypes = []
    lib.pg_table_create_default.restype = C.c_void_p
    lib.pg_table_modify.argtypes = [C.c_void_p, C.c_uint16, C.c_uint16, ld, ld, ld]
    lib.pg_table_modify.restype = C.c_int
    lib.pg_table_get.argtypes = [C.c_void_p, C.c_uint16, C.c_uint16, ldp]
    lib.pg_table_get.restype = C.c_int
    lib.pg_table_destroy.argtypes = [C.c_void_p]
    lib.pg_table_destroy.restype = None
    lib.pg_hmm_device_count.argtypes = []
    lib.pg_hmm_device_count.restype = C.c_int
    lib.pg_hmm_version.argtypes = []
    lib.pg_hmm_version.restype = C.c_char_p
    lib.pg_hmm_genotype_contig.argtypes = [
        C.POINTER(PgContigBatch), C.c_void_p, C.POINTER(PgHmmParams), C.c_int,
        C.POINTER(PgContigResult), C.c_char_p, C.c_size_t]
    lib.pg_hmm_genotype_contig.restype = C.c_int
    lib.pg_job_create.argtypes = [
        C.c_int, C.c_uint32, C.POINTER(PgContigBatch), C.c_void_p,
        C.POINTER(PgHmmParams), C.c_char_p, C.c_size_t]
    lib.pg_job_create.restype = C.c_void_p
    lib.pg_job_run.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    lib.pg_job_run.restype = C.c_int
    lib.pg_job_fetch.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(PgContigResult),
                                 C.c_char_p, C.c_size_t]
    lib.pg_job_fetch.restype = C.c_int
    lib.pg_job_device_results.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p),
                                          u64p, C.POINTER(C.c_void_p), u64p]
    lib.pg_job_device_results.restype = C.c_int
    lib.pg_job_index_ms.argtypes = [C.c_void_p]
    lib.pg_job_index_ms.restype = C.c_double
    lib.pg_job_plan.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.pg_job_plan.restype = C.c_size_t
    lib.pg_job_kernel_ms.argtypes = [C.c_void_p, f64p]
    lib.pg_job_kernel_ms.restype = C.c_int
    lib.pg_job_kernel_name.argtypes = [C.c_int]
    lib.pg_job_kernel_name.restype = C.c_char_p
    lib.pg_job_profile_counters.argtypes = [C.c_void_p, C.c_uint32, u64p]
    lib.pg_job_profile_counters.restype = C.c_int
    lib.pg_job_viterbi_ms.argtypes = [C.c_void_p]
    lib.pg_job_viterbi_ms.restype = C.c_double
    lib.pg_job_device_bytes.argtypes = [C.c_void_p]
    lib.pg_job_device_bytes.restype = C.c_uint64
    lib.pg_job_sweep_mode.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    lib.pg_job_sweep_mode.restype = C.c_int
    lib.pg_job_triangle_chains.argtypes = [C.c_void_p]
    lib.pg_job_triangle_chains.restype = C.c_uint32
    lib.pg_job_destroy.argtypes = [C.c_void_p]
    lib.pg_job_destroy.restype = None
    lib.pg_job_new.argtypes = [C.c_int, C.c_uint32, C.POINTER(PgContigBatch), C.c_void_p, C.POINTER(PgHmmParams),
                               C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    lib.pg_job_new.restype = C.c_int
    lib.pg_cohort_new.argtypes = [C.c_int, C.c_uint32, C.POINTER(PgContigBatch), C.c_uint32, C.POINTER(PgSampleCounts),
                                  C.c_void_p, C.POINTER(PgHmmParams), C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    lib.pg_cohort_new.restype = C.c_int
    lib.pg_job_n_chains.argtypes = [C.c_void_p]
    lib.pg_job_n_chains.restype = C.c_uint32
    lib.pg_job_upload.argtypes = [C.c_void_p, C.POINTER(PgContigBatch), C.POINTER(PgSampleCounts), C.c_char_p, C.c_size_t]
    lib.pg_job_upload.restype = C.c_int
    lib.pg_job_upload_run.argtypes = [C.c_void_p, C.POINTER(PgContigBatch), C.POINTER(PgSampleCounts), C.c_char_p, C.c_size_t]
    lib.pg_job_upload_run.restype = C.c_int
    lib.pg_job_upload_begin.argtypes = [C.c_void_p, C.POINTER(PgSampleCounts), C.c_char_p, C.c_size_t]
    lib.pg_job_upload_begin.restype = C.c_int
    lib.pg_job_upload_end.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.pg_job_upload_end.restype = C.c_int
    lib.pg_job_host_seconds.argtypes = [C.c_void_p, f64p]
    lib.pg_job_host_seconds.restype = C.c_int
    lib.pg_job_upload_bytes.argtypes = [C.c_void_p, u64p]
    lib.pg_job_upload_bytes.restype = C.c_int
    lib.pg_job_packed_results.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), u64p]
    lib.pg_job_packed_results.restype = C.c_int
    lib.pg_hmm_release_cache.argtypes = []
    lib.pg_hmm_release_cache.restype = None
    lib.pg_comm_unique_id.argtypes = [u8p, C.c_char_p, C.c_size_t]
    lib.pg_comm_unique_id.restype = C.c_int
    lib.pg_comm_init.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    lib.pg_comm_init.restype = C.c_int
    lib.pg_comm_init_all.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    lib.pg_comm_init_all.restype = C.c_int
    lib.pg_comm_rank.argtypes = [C.c_void_p]
    lib.pg_comm_rank.restype = C.c_int
    lib.pg_comm_world.argtypes = [C.c_void_p]
    lib.pg_comm_world.restype = C.c_int
    lib.pg_comm_destroy.argtypes = [C.c_void_p]
    lib.pg_comm_destroy.restype = None
    lib.pg_hmm_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int, u64p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    lib.pg_hmm_gather.restype = C.c_int
    lib.pg_hmm_gather_all.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, u64p, C.c_void_p, C.c_void_p,
                                      C.c_char_p, C.c_size_t]
    lib.pg_hmm_gather_all.restype = C.c_int
    lib.pg_emission_table.argtypes = [C.POINTER(PgContigBatch), C.c_void_p, C.c_uint32, C.c_int,
                                      ldp, i32p, C.c_char_p, C.c_size_t]
    lib.pg_emission_table.restype = C.c_int
    lib.pg_transition_probs.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_uint32, C.c_int,
                                        ld, C.c_int, f64p, C.c_char_p, C.c_size_t]
    lib.pg_transition_probs.restype = C.c_int
    lib.pg_job_fetch_all.argtypes = [C.c_void_p, C.POINTER(PgContigResult), C.c_char_p, C.c_size_t]
    lib.pg_job_fetch_all.restype = C.c_int
    lib.pg_job_panel_sizes.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), u64p, u64p]
    lib.pg_job_panel_sizes.restype = C.c_int
    lib.pg_job_fetch_panel.argtypes = [C.c_void_p, C.c_uint32, u32p, u16p, u32p, u16p, u8p, u16p, u32p, u16p, C.c_char_p, C.c_size_t]
    lib.pg_job_fetch_panel.restype = C.c_int
    lib.pg_hmm_announce.argtypes = [C.c_int]
    lib.pg_hmm_announce.restype = None
    lib.pg_hmm_retract.argtypes = [C.c_int]
    lib.pg_hmm_retract.restype = None
    lib.pg_hmm_coalesce_stats.argtypes = [u64p]
    lib.pg_hmm_coalesce_stats.restype = C.c_int
    return lib


# Every symbol include/pangenie_hmm.h declares (checked by tests/test_abi.py).
HIP_ABI_SYMBOLS = [
    "pg_hmm_geno_offsets", "pg_table_create", "pg_table_create_default", "pg_table_modify",
    "pg_table_get", "pg_table_destroy", "pg_hmm_device_count", "pg_hmm_version",
    "pg_hmm_genotype_contig", "pg_job_create", "pg_job_run", "pg_job_fetch",
    "pg_job_device_results", "pg_job_profile_counters", "pg_job_kernel_ms", "pg_job_index_ms", "pg_job_plan", "pg_job_kernel_name", "pg_job_device_bytes", "pg_job_sweep_mode",
    "pg_job_destroy", "pg_emission_table", "pg_transition_probs",
    "pg_job_new", "pg_cohort_new", "pg_job_n_chains", "pg_job_upload", "pg_job_upload_run", "pg_job_upload_begin", "pg_job_upload_end", "pg_job_host_seconds", "pg_job_upload_bytes",
    "pg_job_packed_results", "pg_hmm_release_cache", "pg_job_triangle_chains", "pg_job_viterbi_ms",
    "pg_comm_unique_id", "pg_comm_init", "pg_comm_init_all", "pg_comm_rank", "pg_comm_world", "pg_comm_destroy",
    "pg_hmm_gather", "pg_hmm_gather_all", "pg_hmm_gather_to_host",
    "pg_hmm_announce", "pg_hmm_retract", "pg_hmm_coalesce_stats", "pg_job_fetch_all", "pg_job_panel_sizes", "pg_job_fetch_panel",
]


def load_hip():
    """Load the HIP product library; raises HipExtensionMissing if it was not built."""
    global _hip
    if _hip is None:
        path = Path(os.environ.get("PANGENIE_HMM_LIB", HIP_LIB_PATH))  # override: instrumented builds (tools/)
        if not path.exists():
            raise HipExtensionMissing(
                f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _hip = _bind_hip(C.CDLL(str(path), mode=os.RTLD_GLOBAL if hasattr(os, "RTLD_GLOBAL") else 0))
    return _hip
