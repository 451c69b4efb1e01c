"""Host-side mirror of the reference's HaplotypeSampler interface over the C ABI of
include/pangenie_sampler.h (HIP, gfx950).  No CPU fallback: every entry point that computes goes through
libpangenie_hmm.so and raises when it is missing.

reference: src/haplotypesampler.hpp:16-59 (SampledPaths), :62-96 (HaplotypeSampler),
           src/samplingemissions.hpp, src/samplingtransitions.hpp
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import _lib
from ._lib import PgContigBatch, u8p, u16p, u32p, f64p, c_ld as _c_ld
from .panel import ContigBatch

_bound = False


def _hip():
    global _bound
    lib = _lib.load_hip()
    if not _bound:
        ld = C.c_longdouble
        lib.pg_sampler_emission_costs.argtypes = [C.POINTER(PgContigBatch), u16p]
        lib.pg_sampler_emission_costs.restype = C.c_int
        lib.pg_sampler_transition_cost.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_uint32, ld]
        lib.pg_sampler_transition_cost.restype = C.c_uint32
        lib.pg_sampler_column_minima.argtypes = [u32p, u8p, C.c_uint32, C.c_int, u32p, C.c_char_p, C.c_size_t]
        lib.pg_sampler_column_minima.restype = C.c_int
        lib.pg_sampler_run.argtypes = [C.POINTER(PgContigBatch), C.c_uint32, C.c_double, ld, C.c_uint16, C.c_int,
                                       u32p, u32p, C.c_char_p, C.c_size_t]
        lib.pg_sampler_run.restype = C.c_int
        lib.pg_sampler_run_batch.argtypes = [C.POINTER(PgContigBatch), C.c_uint32, C.c_uint32, C.c_double, ld, C.c_uint16, C.c_int,
                                             C.POINTER(u32p), C.POINTER(u32p), C.c_char_p, C.c_size_t]
        lib.pg_sampler_run_batch.restype = C.c_int
        lib.pg_sampler_then_job.argtypes = [C.POINTER(PgContigBatch), C.c_uint32, C.c_uint32, C.c_int, C.c_double, ld, C.c_uint16,
                                            C.c_void_p, C.c_void_p, C.c_int, C.POINTER(u32p), C.POINTER(u32p),
                                            C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        lib.pg_sampler_then_job.restype = C.c_int
        lib.pg_sampler_last_ms.argtypes = [f64p, C.POINTER(C.c_int)]
        lib.pg_sampler_last_ms.restype = C.c_int
        _bound = True
    return lib


SAMPLER_ABI_SYMBOLS = ["pg_sampler_emission_costs", "pg_sampler_transition_cost", "pg_sampler_column_minima",
                       "pg_sampler_run", "pg_sampler_run_batch", "pg_sampler_last_ms", "pg_sampler_then_job"]
NO_ID = 0xFFFFFFFF


class SampledPaths:
    """reference src/haplotypesampler.hpp:16-59.  sampled_paths[j][v] = path id of sampled path j at variant v."""

    def __init__(self, sampled_paths: Sequence[Sequence[int]] = ()):
        self.sampled_paths = [list(map(int, p)) for p in sampled_paths]

    def mask_indexes(self, column_index: int, max_index: int) -> list[bool]:
        masked = [True] * (max_index + 1)
        for p in self.sampled_paths:
            if column_index >= len(p):
                raise RuntimeError("HaplotypeSampler::SampledPaths::mask_indexes: column_index exceeds number of columns.")
            if p[column_index] > max_index:
                raise RuntimeError("HaplotypeSampler::SampledPaths::mask_indexes: observed index exceeds max_index.")
            masked[p[column_index]] = False
        return masked

    def recombination(self, column_index: int, path_id: int) -> bool:
        if path_id >= len(self.sampled_paths):
            raise RuntimeError("HaplotypeSampler::SampledPaths::recombination: path_id does not exist.")
        if column_index >= len(self.sampled_paths[path_id]):
            raise RuntimeError("HaplotypeSampler::SampledPaths::recombination: column_id does not exist.")
        if column_index > 0:
            return self.sampled_paths[path_id][column_index - 1] != self.sampled_paths[path_id][column_index]
        return False


class SamplingTransitions:
    """reference src/samplingtransitions.cpp:5-23."""

    def __init__(self, from_variant: int, to_variant: int, recomb_rate: float, nr_paths: int, effective_N=25000.0):
        assert from_variant <= to_variant
        self.cost = int(_hip().pg_sampler_transition_cost(int(from_variant), int(to_variant), float(recomb_rate), int(nr_paths),
                                                          _c_ld(effective_N)))

    def compute_transition_cost(self, recombination: bool) -> int:
        return self.cost if recombination else 0


def emission_costs(batch: ContigBatch) -> np.ndarray:
    """SamplingEmissions ctor for every allele slot of the batch (u16 [sumA])."""
    out = np.zeros(max(1, int(batch.allele_off[-1])), np.uint16)
    rc = _hip().pg_sampler_emission_costs(C.byref(batch.as_c()), out.ctypes.data_as(u16p))
    if rc:
        raise RuntimeError(f"pg_sampler_emission_costs: error {rc}")
    return out[: int(batch.allele_off[-1])]


class SamplingEmissions:
    """reference src/samplingemissions.cpp:9-45 for ONE variant of a batch."""

    def __init__(self, batch: ContigBatch, variant: int = 0, _costs: np.ndarray | None = None):
        costs = emission_costs(batch) if _costs is None else _costs
        lo, hi = int(batch.allele_off[variant]), int(batch.allele_off[variant + 1])
        ids = batch.allele_id[lo:hi]
        self.allele_penalties = np.zeros(int(ids.max()) + 1, np.uint16)
        self.allele_penalties[ids] = costs[lo:hi]
        self.default_penalty = 25

    def get_emission_cost(self, allele_id: int) -> int:
        return int(self.allele_penalties[allele_id])

    def penalize(self, allele_id: int, penalty: int):
        v = (int(self.allele_penalties[allele_id]) + int(penalty)) & 0xFFFF
        self.allele_penalties[allele_id] = min(v, self.default_penalty)


def column_minima(column: Sequence[int], mask: Sequence[bool], device: int = 0):
    """HaplotypeSampler::get_column_minima on the device -> (first_id, second_id, first_val, second_val)."""
    col = np.ascontiguousarray(column, np.uint32)
    m = np.ascontiguousarray(mask, np.uint8)
    assert col.size > 1 and col.size == m.size
    out = np.zeros(4, np.uint32)
    err = C.create_string_buffer(256)
    rc = _hip().pg_sampler_column_minima(col.ctypes.data_as(u32p), m.ctypes.data_as(u8p), col.size, device,
                                         out.ctypes.data_as(u32p), err, 256)
    if rc:
        raise RuntimeError(err.value.decode())
    return tuple(int(x) for x in out)


def last_ms():
    """((expand, forward, backtrack) kernel ms of this thread's last run, waves per workgroup of the fast kernel or 0)"""
    ms = np.zeros(3)
    k = C.c_int(0)
    _hip().pg_sampler_last_ms(ms.ctypes.data_as(f64p), C.byref(k))
    return (float(ms[0]), float(ms[1]), float(ms[2])), int(k.value)


def sample_contigs(batches: Sequence[ContigBatch], size: int, recombrate: float = 1.26, effective_N=25000.0,
                   allele_penalty: int = 10, device: int = 0):
    """pg_sampler_run_batch: all contigs of a sample in one call (one workgroup per contig and pass).
    -> (list of sampled paths [size, V_g], list of best scores [size])"""
    n = len(batches)
    arr = (PgContigBatch * n)(*[b.as_c() for b in batches])
    sampled = [np.zeros((size, max(1, b.n_variants)), np.uint32) for b in batches]
    best = [np.zeros(max(1, size), np.uint32) for _ in batches]
    sp = (u32p * n)(*[a.ctypes.data_as(u32p) for a in sampled])
    bp = (u32p * n)(*[a.ctypes.data_as(u32p) for a in best])
    err = C.create_string_buffer(512)
    rc = _hip().pg_sampler_run_batch(arr, n, size, float(recombrate), _c_ld(effective_N), int(allele_penalty), device, sp, bp, err, 512)
    if rc:
        raise RuntimeError(f"pg_sampler_run_batch: {err.value.decode()} (error {rc})")
    return [s[:, : b.n_variants] for s, b in zip(sampled, batches)], [x[:size] for x in best]


def sample_then_job(batches: Sequence[ContigBatch], size: int, table, params=None, add_reference: bool = False,
                    recombrate: float = 1.26, effective_N=25000.0, allele_penalty: int = 10, device: int = 0, want_paths: bool = True):
    """pg_sampler_then_job: the sampler, UniqueKmers::update_paths and the genotyping job's upload without the panel
    leaving the device.  -> (hmm.Job over the reduced panels — job.batches are read back from the device —,
    sampled paths per contig [size, V] or None, best scores per contig [size] or None)."""
    from . import hmm
    n = len(batches)
    arr = (PgContigBatch * n)(*[b.as_c() for b in batches])
    sampled = [np.zeros((size, max(1, b.n_variants)), np.uint32) for b in batches] if want_paths else None
    best = [np.zeros(max(1, size), np.uint32) for _ in batches] if want_paths else None
    sp = (u32p * n)(*[a.ctypes.data_as(u32p) for a in sampled]) if want_paths else None
    bp = (u32p * n)(*[a.ctypes.data_as(u32p) for a in best]) if want_paths else None
    params = params or hmm.make_params()
    err = C.create_string_buffer(512)
    h = C.c_void_p()
    rc = _hip().pg_sampler_then_job(arr, n, size, int(bool(add_reference)), float(recombrate), _c_ld(effective_N), int(allele_penalty),
                                    table.h, C.byref(params), device, sp, bp, C.byref(h), err, 512)
    if rc:
        raise hmm.PanGenieError(rc, err.value.decode(errors="replace"))
    job = hmm.Job.from_handle(h.value, table, params)
    if want_paths:
        return job, [s[:, : b.n_variants] for s, b in zip(sampled, batches)], [x[:size] for x in best]
    return job, None, None


class HaplotypeSampler:
    """reference src/haplotypesampler.cpp:20-77: `size` Viterbi passes over the panel, then (update_unique_kmers)
    the panel reduced to the sampled paths — here returned as `self.panel` (a new ContigBatch) instead of
    mutating the input."""

    def __init__(self, batch: ContigBatch, size: int, recombrate: float = 1.26, effective_N=25000.0,
                 add_reference: bool = False, allele_penalty: int = 10, device: int = 0):
        self.batch = batch
        self.best_scores: list[int] = []
        self._paths = SampledPaths()
        self.kernel_ms = (0.0, 0.0, 0.0)
        self.kernel = 0
        self.panel = batch
        if size < 1:
            return
        V = batch.n_variants
        sampled = np.zeros((size, max(V, 1)), np.uint32)
        best = np.zeros(size, np.uint32)
        err = C.create_string_buffer(512)
        lib = _hip()
        rc = lib.pg_sampler_run(C.byref(batch.as_c()), size, float(recombrate), _c_ld(effective_N), int(allele_penalty),
                                device, sampled.ctypes.data_as(u32p), best.ctypes.data_as(u32p), err, 512)
        if rc:
            raise RuntimeError(f"pg_sampler_run: {err.value.decode()} (error {rc})")
        self.kernel_ms, self.kernel = last_ms()
        sampled = sampled[:, :V]
        self.best_scores = [int(x) for x in best] if V else []
        if add_reference:
            sampled = np.vstack([sampled, np.zeros((1, V), np.uint32)])
        self.sampled = sampled
        self._paths = SampledPaths(sampled.tolist()) if V else SampledPaths()
        if V:
            self.panel = batch.update_paths(sampled)

    def get_sampled_paths(self) -> SampledPaths:
        return self._paths
