"""Python face of the C ABI (include/pangenie_hmm.h): ProbabilityTable, HMM runs, resident jobs.

All compute happens in libpangenie_hmm.so (HIP kernels, gfx950).  This module only
marshals numpy arrays and rebuilds the reference's `long double` likelihoods from
(lik, lik_exp) — `np.longdouble` is the x87 80-bit type on x86-64.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import (PG_N_KERNEL_CLASSES, PgContigBatch, PgContigResult, PgHmmParams, PgSampleCounts, f64p, i32p,
                   ldp, u8p, u16p, u64p)
from .genotyping_result import GenotypingResult, results_from_flat
from .panel import ContigBatch

LD = np.longdouble
_ERRLEN = 512


class PanGenieError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def make_params(recombrate: float = 1.26, uniform: bool = False, effective_N=25000.0,
                run_genotyping: bool = True, run_phasing: bool = False) -> PgHmmParams:
    """HMM constructor arguments (reference src/hmm.hpp:38)."""
    p = PgHmmParams()
    p.effective_N = _lib.c_ld(effective_N)
    p.recombrate = float(recombrate)
    p.uniform = int(bool(uniform))
    p.run_genotyping = int(bool(run_genotyping))
    p.run_phasing = int(bool(run_phasing))
    return p


class ProbabilityTable:
    """Mirror of the reference ProbabilityTable (src/probabilitytable.hpp:13-29)."""

    def __init__(self, cov_min: int = 0, cov_max: int = 0, count_max: int = 0,
                 regularization=0.0, default: bool = False):
        self._lib = _lib.load_hip()
        if default:
            self.h = self._lib.pg_table_create_default()
        else:
            self.h = self._lib.pg_table_create(cov_min, cov_max, count_max, _lib.c_ld(regularization))
        if not self.h:
            raise MemoryError("pg_table_create failed")

    def modify(self, coverage: int, count: int, p0, p1, p2) -> None:
        rc = self._lib.pg_table_modify(self.h, coverage, count, _lib.c_ld(p0), _lib.c_ld(p1), _lib.c_ld(p2))
        if rc:
            raise RuntimeError("ProbabilityTable::modify_probability: no precomputed values for these parameters.")

    def get(self, coverage: int, count: int) -> np.ndarray:
        out, ptr = _lib.ld_out(3)
        self._lib.pg_table_get(self.h, coverage, count, ptr)
        return out

    def __del__(self):
        try:
            if self.h:
                self._lib.pg_table_destroy(self.h)
                self.h = None
        except Exception:
            pass


class ContigResult:
    """Host buffers of one pg_contig_result + long double reconstruction."""

    def __init__(self, batch: ContigBatch):
        V = batch.n_variants
        self.batch = batch
        self.geno_off = batch.geno_off
        n = int(self.geno_off[-1])
        self.lik = np.zeros(max(n, 1), np.float64)[:n]
        self.lik_exp = np.zeros(max(n, 1), np.int32)[:n]  # one exponent per genotype bin
        self.kept = np.zeros(max(V, 1), np.uint8)[:V]
        nA = int(batch.allele_off[-1]) if V else 0
        self.allele_present = np.zeros(max(nA, 1), np.uint8)[:nA]
        self.n_kmers = np.zeros(max(V, 1), np.uint16)[:V]
        self.coverage = np.zeros(max(V, 1), np.uint16)[:V]
        self.n_columns = 0
        # run_phasing: alleles of the Viterbi path's two haplotypes at kept variants
        self.haplotype_1 = np.zeros(max(V, 1), np.uint16)[:V]
        self.haplotype_2 = np.zeros(max(V, 1), np.uint16)[:V]
        self.run_genotyping = True
        self._c = PgContigResult()
        self._c.lik = self.lik.ctypes.data_as(f64p)
        self._c.lik_exp = self.lik_exp.ctypes.data_as(i32p)
        self._c.kept = self.kept.ctypes.data_as(u8p)
        self._c.allele_present = self.allele_present.ctypes.data_as(u8p)
        self._c.n_kmers = self.n_kmers.ctypes.data_as(u16p)
        self._c.coverage = self.coverage.ctypes.data_as(u16p)
        self._c.haplotype_1 = self.haplotype_1.ctypes.data_as(u16p)
        self._c.haplotype_2 = self.haplotype_2.ctypes.data_as(u16p)

    def likelihoods_ld(self) -> np.ndarray:
        """Unnormalised genotype likelihoods as 80-bit long double: lik[g] * 2^lik_exp[g]."""
        return np.ldexp(self.lik.astype(LD), self.lik_exp.astype(np.int64))

    def genotyping_results(self) -> List[GenotypingResult]:
        return results_from_flat(self.batch, self.likelihoods_ld(), self.kept, self.allele_present,
                                 self.n_kmers, self.coverage, self.haplotype_1, self.haplotype_2,
                                 with_likelihoods=self.run_genotyping)


def genotype_contig(batch: ContigBatch, table: ProbabilityTable, params: Optional[PgHmmParams] = None,
                    device: int = 0, announced: bool = False, into: Optional[ContigResult] = None) -> ContigResult:
    """One blocking call = the body of HMM::HMM for one (contig, path subset)
    (reference src/hmm.cpp:25-63 with normalize=false, as run_genotyping calls it,
    src/commands.cpp:160).  Thread-safe (ctypes drops the GIL for the call); calls in flight together are merged
    into one device job.  `announced`: the caller has called announce(device) for this call (what the C++ adapter's
    constructor does before it flattens).  `into`: result buffers to reuse."""
    lib = _lib.load_hip()
    params = params or make_params()
    if announced:
        q = PgHmmParams()
        C.memmove(C.byref(q), C.byref(params), C.sizeof(PgHmmParams))
        q.reserved = _lib.PG_CALL_ANNOUNCED
        params = q
    try:
        res = into if into is not None else ContigResult(batch)
        err = C.create_string_buffer(_ERRLEN)
        cb, th = batch.as_c(), table.h
    except Exception:
        if announced:   # the C call that would have consumed the announcement is never made: take it back (a leader would
            lib.pg_hmm_retract(device)   # otherwise wait PG_COALESCE_WAIT_MS for a caller that is not coming)
        raise
    rc = lib.pg_hmm_genotype_contig(C.byref(cb), th, C.byref(params), device, C.byref(res._c), err, _ERRLEN)
    if rc:
        raise PanGenieError(rc, err.value.decode(errors="replace"))
    res.n_columns = int(res._c.n_columns)
    res.run_genotyping = bool(params.run_genotyping)
    return res


def announce(device: int = 0) -> None:
    """pg_hmm_announce: a genotype_contig(..., announced=True) call on `device` will follow shortly."""
    _lib.load_hip().pg_hmm_announce(device)


def coalesce_stats() -> dict:
    out = (C.c_uint64 * 3)()
    _lib.load_hip().pg_hmm_coalesce_stats(out)
    return {"merged_jobs": int(out[0]), "calls": int(out[1]), "largest_merge": int(out[2])}


def genotype_contigs_threaded(batches: Sequence[ContigBatch], table: ProbabilityTable, params=None, device: int = 0,
                              n_threads: Optional[int] = None, into: Optional[Sequence[ContigResult]] = None):
    """The reference's calling pattern (src/commands.cpp:949-978): one one-shot call per contig from `n_threads`
    worker threads (default: one per batch).  `params` may be a list (one per batch).  Returns the list of
    ContigResult / raised exceptions, in batch order."""
    import threading
    n = len(batches)
    n_threads = n_threads or n
    plist = list(params) if isinstance(params, (list, tuple)) else [params] * n
    out: list = [None] * n
    nxt = [0]
    lock = threading.Lock()

    n_workers = min(n_threads, n)
    for _ in range(n_workers):  # the pool's workers all pick up their first job at once: announce those calls up front
        announce(device)

    def worker():
        first = True
        while True:
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= n:
                if first:
                    _lib.load_hip().pg_hmm_retract(device)
                return
            if not first:
                announce(device)
            first = False
            try:
                out[i] = genotype_contig(batches[i], table, plist[i], device, announced=True, into=into[i] if into else None)
            except Exception as e:  # noqa: BLE001 - handed to the caller
                out[i] = e

    ts = [threading.Thread(target=worker) for _ in range(n_workers)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return out


class Job:
    """Resident multi-chain job: upload once, run many times (bench / pipelines).

    Job(batches, ...)                      one chain per batch (contig x path subset)
    Job.cohort(index, samples, ...)        chains = samples x contigs over ONE shared index
                                           (include/pangenie_hmm.h: pg_cohort_new); chain id =
                                           sample * n_contigs + contig
    """

    def __init__(self, batches: Sequence[ContigBatch], table: ProbabilityTable,
                 params: Optional[PgHmmParams] = None, device: int = 0, _cohort=None):
        self._lib = _lib.load_hip()
        self.table = table
        self.params = params or make_params()
        self.index = list(batches)
        self._arr = (PgContigBatch * len(self.index))(*[b.as_c() for b in self.index])
        err = C.create_string_buffer(_ERRLEN)
        h = C.c_void_p()
        if _cohort is None:
            self.batches = self.index
            self._samples = None
            rc = self._lib.pg_job_new(device, len(self.index), self._arr, table.h, C.byref(self.params),
                                      C.byref(h), err, _ERRLEN)
        else:
            self._samples, self._keep = self._marshal_samples(_cohort)
            self.batches = [b.with_counts(kc, cov) for (kcs, covs) in _cohort for b, kc, cov in zip(self.index, kcs, covs)]
            rc = self._lib.pg_cohort_new(device, len(self.index), self._arr, len(_cohort), self._samples, table.h,
                                         C.byref(self.params), C.byref(h), err, _ERRLEN)
        if rc:
            raise PanGenieError(rc, err.value.decode(errors="replace"))
        self.h = h.value

    @classmethod
    def from_handle(cls, handle: int, table: ProbabilityTable, params: PgHmmParams, positions: Optional[Sequence[np.ndarray]] = None) -> "Job":
        """A job made elsewhere behind the C ABI (pg_sampler_then_job): its panels are read back from the device
        (pg_job_fetch_panel) so that results can be fetched and interpreted like any other job's."""
        self = cls.__new__(cls)
        self._lib = _lib.load_hip()
        self.table, self.params, self.h = table, params, handle
        self._samples = None
        self.batches = [self.fetch_panel(c) for c in range(self.n_chains)]
        self.index = self.batches
        return self

    def fetch_panel(self, contig: int) -> ContigBatch:
        """pg_job_fetch_panel: the inputs chain `contig` holds on the device as a ContigBatch (positions and coverage are
        not part of what the C ABI returns here: zeros)."""
        V, H, sK, sA = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_uint64()
        if self._lib.pg_job_panel_sizes(self.h, contig, C.byref(V), C.byref(H), C.byref(sK), C.byref(sA)):
            raise PanGenieError(-1, "pg_job_panel_sizes")
        V, H, sK, sA = V.value, H.value, sK.value, sA.value
        z = lambda n, dt: np.zeros(max(int(n), 1), dt)
        koff, aoff = z(V + 1, np.uint32), z(V + 1, np.uint32)
        kc, aid, afl, ako, akm, pa = z(sK, np.uint16), z(sA, np.uint16), z(sA, np.uint8), z(sA, np.uint16), z(sA, np.uint32), z(V * H, np.uint16)
        err = C.create_string_buffer(_ERRLEN)
        p = lambda a, t: a.ctypes.data_as(t)
        rc = self._lib.pg_job_fetch_panel(self.h, contig, p(koff, _lib.u32p), p(kc, u16p), p(aoff, _lib.u32p), p(aid, u16p), p(afl, u8p),
                                          p(ako, u16p), p(akm, _lib.u32p), p(pa, u16p), err, _ERRLEN)
        if rc:
            raise PanGenieError(rc, err.value.decode(errors="replace"))
        return ContigBatch(H, np.zeros(V, np.uint64), np.zeros(V, np.uint16), koff[:V + 1], kc[:sK], aoff[:V + 1], aid[:sA], afl[:sA],
                           ako[:sA], akm[:sA], pa[:V * H])

    @classmethod
    def cohort(cls, index: Sequence[ContigBatch], samples, table: ProbabilityTable,
               params: Optional[PgHmmParams] = None, device: int = 0) -> "Job":
        """samples: list (one entry per sample) of (kmer_counts, coverages), each a list with one uint16
        array per index contig."""
        return cls(index, table, params, device, _cohort=list(samples))

    def _marshal_samples(self, samples):
        n, nc = len(samples), len(self.index)
        arr = (PgSampleCounts * n)()
        keep = []
        for s, (kcs, covs) in enumerate(samples):
            assert len(kcs) == nc and len(covs) == nc
            kc = [np.ascontiguousarray(a, np.uint16) if len(a) else np.zeros(1, np.uint16) for a in kcs]
            cv = [np.ascontiguousarray(a, np.uint16) if len(a) else np.zeros(1, np.uint16) for a in covs]
            pk = (u16p * nc)(*[a.ctypes.data_as(u16p) for a in kc])
            pc = (u16p * nc)(*[a.ctypes.data_as(u16p) for a in cv])
            arr[s].kmer_count = pk
            arr[s].coverage = pc
            keep += [kc, cv, pk, pc]
        return arr, keep

    @property
    def n_chains(self) -> int:
        return int(self._lib.pg_job_n_chains(self.h))

    def upload(self, samples=None) -> None:
        """Re-upload the inputs (same shapes): all arrays of a plain job; for a cohort job the given
        per-sample counts (index stays resident)."""
        err = C.create_string_buffer(_ERRLEN)
        if self._samples is None:
            rc = self._lib.pg_job_upload(self.h, self._arr, None, err, _ERRLEN)
        else:
            if samples is not None:
                self._samples, self._keep = self._marshal_samples(list(samples))
            rc = self._lib.pg_job_upload(self.h, None, self._samples, err, _ERRLEN)
        if rc:
            raise PanGenieError(rc, err.value.decode(errors="replace"))

    def upload_run(self) -> None:
        """pg_job_upload_run: re-upload every input and run, in one call (the upload of the shorter chains hides behind the long
        chains' phase 1 where the job can do that); a plain (non-cohort) job."""
        if self._samples is not None:
            raise PanGenieError(-2, "upload_run: a cohort job uploads samples with upload() / upload_begin()")
        err = C.create_string_buffer(_ERRLEN)
        rc = self._lib.pg_job_upload_run(self.h, self._arr, None, err, _ERRLEN)
        if rc:
            raise PanGenieError(rc, err.value.decode(errors="replace"))

    def upload_begin(self, samples=None) -> None:
        """pg_job_upload_begin: start copying the next batch of samples of a cohort job into the job's second set of
        per-sample arrays while the current batch is genotyped (run()); upload_end() makes it the current one.  Fetch
        the previous run's results before upload_end()."""
        if self._samples is None:
            raise PanGenieError(-2, "upload_begin: not a cohort job")
        # marshalled into locals: when an upload is already in flight the C side refuses this call, and the uploader thread
        # is still reading the FIRST call's arrays — which only self._next_* keep alive
        if samples is not None:
            nxt, keep = self._marshal_samples(list(samples))
        else:
            nxt, keep = self._samples, self._keep
        err = C.create_string_buffer(_ERRLEN)
        rc = self._lib.pg_job_upload_begin(self.h, nxt, err, _ERRLEN)
        if rc:
            raise PanGenieError(rc, err.value.decode(errors="replace"))
        self._next_samples, self._next_keep = nxt, keep

    def upload_end(self) -> None:
        err = C.create_string_buffer(_ERRLEN)
        rc = self._lib.pg_job_upload_end(self.h, err, _ERRLEN)
        if rc:
            raise PanGenieError(rc, err.value.decode(errors="replace"))
        self._samples, self._keep = self._next_samples, self._next_keep

    def host_seconds(self) -> dict:
        out = (C.c_double * 4)()
        self._lib.pg_job_host_seconds(self.h, out)
        return {"alloc_s": out[0], "upload_s": out[1], "run_s": out[2], "fetch_s": out[3]}

    def upload_bytes(self) -> dict:
        out = (C.c_uint64 * 2)()
        self._lib.pg_job_upload_bytes(self.h, out)
        return {"index": int(out[0]), "samples": int(out[1])}

    def packed_results(self):
        """(lik_ptr, lik_exp_ptr, n_lik_total): the posteriors of all chains as two packed device ranges."""
        d_lik, d_exp, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._lib.pg_job_packed_results(self.h, C.byref(d_lik), C.byref(d_exp), C.byref(n))
        return d_lik.value, d_exp.value, int(n.value)

    def run(self, stream: int = 0) -> None:
        err = C.create_string_buffer(_ERRLEN)
        rc = self._lib.pg_job_run(self.h, C.c_void_p(stream) if stream else None, err, _ERRLEN)
        if rc:
            raise PanGenieError(rc, err.value.decode(errors="replace"))

    def fetch(self, contig: int) -> ContigResult:
        res = ContigResult(self.batches[contig])
        err = C.create_string_buffer(_ERRLEN)
        rc = self._lib.pg_job_fetch(self.h, contig, C.byref(res._c), err, _ERRLEN)
        if rc:
            raise PanGenieError(rc, err.value.decode(errors="replace"))
        res.n_columns = int(res._c.n_columns)
        res.run_genotyping = bool(self.params.run_genotyping)
        return res

    def fetch_all(self, into: Optional[Sequence[ContigResult]] = None) -> List[ContigResult]:
        """pg_job_fetch_all: every chain's results with one synchronisation; `into` = buffers to reuse."""
        res = list(into) if into is not None else [ContigResult(b) for b in self.batches]
        arr = (PgContigResult * len(res))(*[r._c for r in res])
        err = C.create_string_buffer(_ERRLEN)
        rc = self._lib.pg_job_fetch_all(self.h, arr, err, _ERRLEN)
        if rc:
            raise PanGenieError(rc, err.value.decode(errors="replace"))
        for r, c in zip(res, arr):
            r.n_columns = int(c.n_columns)
            r.run_genotyping = bool(self.params.run_genotyping)
        return res

    def viterbi_ms(self) -> float:
        """elapsed ms of the Viterbi kernels (run_phasing) of the last run"""
        return float(self._lib.pg_job_viterbi_ms(self.h))

    def device_results(self, contig: int):
        """(lik_ptr, n_lik, lik_exp_ptr, n_variants): device pointers for an RCCL gather."""
        d_lik, d_exp = C.c_void_p(), C.c_void_p()
        n_lik, n_var = C.c_uint64(), C.c_uint64()
        self._lib.pg_job_device_results(self.h, contig, C.byref(d_lik), C.byref(n_lik),
                                        C.byref(d_exp), C.byref(n_var))
        return d_lik.value, n_lik.value, d_exp.value, n_var.value

    def index_ms(self) -> float:
        """pg_job_index_ms: hipEvent time of the last index pass (what the index alone decides, formed once per uploaded index)"""
        return float(self._lib.pg_job_index_ms(self.h))

    def plan(self) -> str:
        """pg_job_plan: which kernels run for which chains of this job"""
        n = int(self._lib.pg_job_plan(self.h, None, 0))
        buf = C.create_string_buffer(n)
        self._lib.pg_job_plan(self.h, buf, n)
        return buf.value.decode(errors="replace")

    def kernel_ms(self) -> dict:
        ms = (C.c_double * PG_N_KERNEL_CLASSES)()
        self._lib.pg_job_kernel_ms(self.h, ms)
        return {self._lib.pg_job_kernel_name(i).decode(): ms[i] for i in range(PG_N_KERNEL_CLASSES)}

    def profile_counters(self, contig: int = 0) -> np.ndarray:
        out = np.zeros(64, np.uint64)
        self._lib.pg_job_profile_counters(self.h, contig, out.ctypes.data_as(u64p))
        return out

    def device_bytes(self) -> int:
        return int(self._lib.pg_job_device_bytes(self.h))

    def sweep_mode(self):
        """("fused" | "chunked", chunk_cols) — see include/pangenie_hmm.h:pg_job_sweep_mode."""
        k = C.c_uint32(0)
        m = self._lib.pg_job_sweep_mode(self.h, C.byref(k))
        return ("chunked" if m == 1 else "fused"), int(k.value)

    def triangle_chains(self) -> int:
        """chains whose columns are stored as upper triangles (fused jobs, lean chains)"""
        return int(self._lib.pg_job_triangle_chains(self.h))

    def close(self):
        if getattr(self, "h", None):
            self._lib.pg_job_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def emission_table(batch: ContigBatch, table: ProbabilityTable, v: int, device: int = 0):
    """EmissionProbabilityComputer of variant v on the device (A x A over all allele slots)."""
    lib = _lib.load_hip()
    A = int(batch.allele_off[v + 1] - batch.allele_off[v])
    out = np.zeros(A * A, dtype=LD)
    az = C.c_int32(0)
    err = C.create_string_buffer(_ERRLEN)
    rc = lib.pg_emission_table(C.byref(batch.as_c()), table.h, v, device, out.ctypes.data_as(ldp),
                               C.byref(az), err, _ERRLEN)
    if rc:
        raise PanGenieError(rc, err.value.decode(errors="replace"))
    return out.reshape(A, A), bool(az.value)


def transition_probs(from_pos: int, to_pos: int, recombrate: float, nr_paths: int, uniform: bool,
                     effective_N, device: int = 0) -> np.ndarray:
    """TransitionProbabilityComputer on the device: {no switch, one switch, two switches}."""
    lib = _lib.load_hip()
    out = (C.c_double * 3)()
    err = C.create_string_buffer(_ERRLEN)
    rc = lib.pg_transition_probs(from_pos, to_pos, recombrate, nr_paths, int(uniform), _lib.c_ld(effective_N),
                                 device, out, err, _ERRLEN)
    if rc:
        raise PanGenieError(rc, err.value.decode(errors="replace"))
    return np.array([out[0], out[1], out[2]])
