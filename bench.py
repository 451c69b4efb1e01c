#!/usr/bin/env python3
"""bench.py — throughput of the genotyping hot path (emissions + forward-backward HMM).

A "step" = one full pass of the device path (k_prep -> k_compact -> k_records -> k_sweep phase 1 ->
phase 2 -> bins, plus the exchange of the posteriors to rank 0 when N > 1) over synthetic contig
batches that are already resident in HBM.  metric = genotyped variants/sec (whole job).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload genome24_h64|chr22_h64|...]

One JSON line on rank 0:
  value              resident rate of the main workload (inputs in HBM when the timed region starts)
  value_end_to_end   the same workload timed from host buffers to host results: H2D of every input +
                     run + D2H of every result, into the arena the job already holds
  roofline           dominant kernel of the main workload against the HBM peak
  cohort             second measurement in the same line: many (sample x contig) chains against ONE
                     shared index (pg_cohort_new) — the regime in which the sweep
                     is HBM-bound; per-rank work fixed (weak scaling), with its own roofline
  cpu_baseline       the CPU port of the reference on a bounded sample (N = 1 only)

N > 1 (launched by the driver with torch.distributed.run, one rank per GPU): the chains of the main
workload are SHARDED over the ranks by longest-processing-time-first (pangenie_amd/dist.py), i.e.
BASELINE.json configs[3] as written (strong scaling: the total work is fixed); rank 0 collects all
posteriors with one batched group of point-to-point sends per step.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from pangenie_amd import hmm  # noqa: E402
from pangenie_amd.panel import (algorithmic_bytes, default_table_args, synthetic_panel,  # noqa: E402
                                synthetic_sample_counts)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable

# BASELINE.json configs -> shapes (variants, haplotypes, k-mers/variant, multiallelic)
WORKLOADS = {
    "contig_h16": dict(V=50_000, H=16, K=20, multi=0.0, cfg="configs[1]: 1 contig, 50k variants, 16 haplotypes, ~20 k-mers/var"),
    "chr22_h64": dict(V=200_000, H=64, K=20, multi=0.0, cfg="configs[2]: chr22-scale, 200k variants, 64 haplotypes"),
    "chr22_h128": dict(V=60_000, H=128, K=20, multi=0.2, cfg="configs[4] per-GPU slice: 128 haplotypes, 20% multiallelic"),
    "genome24_small": dict(V=40_000, H=64, K=20, multi=0.0, chains=24, cfg="configs[3] shape at 1/5 length, equal contigs: 24 contigs, 64 haplotypes, one GPU"),
    # BASELINE.json configs[3]: whole genome, 24 contigs with human-like length proportions, 5M variants,
    # 64 haplotypes.  164 GB of column slots: fits ONE 288 GB MI355X, so it is the single-GPU workload.
    "genome24_h64": dict(V=5_000_000, H=64, K=20, multi=0.0, chains=24, genome=True,
                         cfg="configs[3]: whole genome, 24 contigs (human chromosome length proportions), 5M variants, 64 haplotypes"),
    # BASELINE.json configs[4]: the HPRC-style panel — 5M variants, 128 haplotypes, 20 % multiallelic — is an 8-GPU
    # workload (656 GB of columns); on ONE GPU the bench runs the share rank 0 gets under `--gpus 8` (LPT over 8 ranks:
    # ~625k variants, ~82 GB of columns), with --gpus N > 1 the whole panel sharded over the N ranks.
    "hprc_h128": dict(V=5_000_000, H=128, K=20, multi=0.2, chains=24, genome=True, share_of=8,
                      cfg="configs[4]: HPRC-style panel, 24 contigs, 5M variants, 128 haplotypes, 20% multiallelic"),
}
# the sampler measurement: contigs x variants x panel paths, 15 passes (the reference's default panel size)
SAMPLER = {"contigs": 8, "V": 40_000, "H": 215, "size": 15}
# the phasing (Viterbi) measurement: contigs x variants x selected paths (the reference's callers pass 30, src/commands.cpp:939)
VITERBI = {"contigs": 8, "V": 100_000, "H": 30}
# the cohort measurement: samples x contigs over one shared index
COHORT = dict(samples=64, contigs=8, V=16_000, H=64, K=20)  # 512 chains, 194 GB of the 288 (columns as compact triangles)
COHORTS_MORE = {
    "cohort_h16": dict(samples=512, contigs=8, V=8_000, H=16, K=20, distinct=16),              # 4096 chains, 32.8 M variants
    "cohort_h128": dict(samples=16, contigs=8, V=3_000, H=128, K=20, multi=0.2, distinct=16),  # 128 chains, 50 GB of columns
    # THE DEFAULT PRODUCTION SHAPE of every panel with more than 100 haplotypes: 15 sampled paths + the reference path = 16
    # paths (src/commands.cpp:799-803, src/haplotypesampler.cpp:43) whose bubbles keep every allele those paths carry
    # (src/multiallelicuniquekmers.cpp:195-232): a fifth of the objects with 3-5 alleles ...
    "cohort_h16m": dict(samples=512, contigs=8, V=8_000, H=16, K=20, multi=0.2, distinct=16),
    # ... and 2 % of them bubbles of 6-12 alleles, of which the 16 paths carry up to nine (wide columns)
    "cohort_h16w": dict(samples=512, contigs=8, V=8_000, H=16, K=20, multi=0.2, wide=0.02, distinct=16),
    # an unsampled <= 100-haplotype panel (src/commands.cpp:799) with its multiallelic bubbles: 32 samples = 256 chains, triangle
    # columns (75 GB; phase 2 on k_sweep_leanx2 since round 6) ...
    "cohort_h64m": dict(samples=32, contigs=8, V=16_000, H=64, K=20, multi=0.2, distinct=16),
    # ... and with 2 % bubbles of 6-12 alleles: wide columns at 64 paths have no fused kernel — the job runs CHUNKED, full columns
    # (134 GB) — the line is here so that the gap is a measured number (VERDICT r5 item 3 asked for a fused one)
    "cohort_h64w": dict(samples=32, contigs=8, V=16_000, H=64, K=20, multi=0.2, wide=0.02, distinct=16),
    # a user-chosen panel size (`-x 16` + the reference path = 17 ... 31 paths; NOT the default, which is 15 + 1 = 16): pads to 32
    "cohort_h17": dict(samples=128, contigs=8, V=8_000, H=17, K=20, multi=0.2, distinct=16),    # 1024 chains, 8.2 M variants, 67 GB of columns
}
# The same production shape as it arrives when every sample brings its OWN sampled panel (HaplotypeSampler picks the 15 paths per
# sample: src/haplotypesampler.cpp:110-294): chains = samples x contigs, each with an index of its own (pg_job_new, no shared
# index) — the four half-chains of a wave then differ in which columns are multiallelic / wide (the wave-uniform branches of
# k_sweep_small16x are taken more often than in a cohort over one index).  64 distinct panels, each used by 64 chains.
PANEL_JOB = dict(chains=4096, distinct=64, V=8_000, H=16, K=20, multi=0.2, wide=0.02)
# GRCh38 chromosome lengths (Mb) 1..22, X, Y: proportions of the 24 synthetic contigs
CONTIG_MB = [248, 242, 198, 190, 181, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57]


def genome_contig_sizes(total_variants: int):
    tot = float(sum(CONTIG_MB))
    return [int(round(total_variants * mb / tot)) for mb in CONTIG_MB]


def cpu_baseline(batches, H, sample_variants):
    """Reported baseline only: the CPU oracle (our long-double port of the reference path) on a
    bounded sample of the same workload.  The reference runs one thread per contig x subset
    (src/commands.cpp:949-953), so a multi-contig workload is timed with one thread per contig
    (up to the host's cores), each on the first `sample_variants` variants of its own contig.
    Plain Python threads: the oracle is a C library without global state and ctypes drops the GIL
    for the duration of the call, so the threads run on separate cores.  The port is a CONSERVATIVE
    baseline: per thread it runs about 3x faster than the compiled reference measured in the survey
    (flat arrays instead of shared_ptr / virtual calls / std::map)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle as orc  # checker / baseline leg only
    workers = max(1, min(len(batches), os.cpu_count() or 1))
    subs = [b.slice(0, min(sample_variants, b.n_variants)) for b in batches[:workers]]
    table = orc.OracleTable(*default_table_args())
    params = orc.make_params(1.26, False, 1e-5)

    def one(sub):
        t0 = time.perf_counter()
        orc.genotype_contig(sub, table, params)
        return time.perf_counter() - t0

    t0 = time.perf_counter()
    if workers == 1:
        times = [one(subs[0])]
    else:
        with ThreadPoolExecutor(workers) as pool:
            times = list(pool.map(one, subs))
    wall = time.perf_counter() - t0
    n = sum(sb.n_variants for sb in subs)
    return {"value": n / wall, "unit": "variants/s", "cores": workers, "kind": "port",
            "sample": f"first {subs[0].n_variants} variants of each of {workers} synthetic contig(s) (H={H}), one thread per contig "
                      f"(the reference's own parallelism), oracle/pg_oracle.c long double; {wall:.1f} s wall; "
                      f"{os.cpu_count()} host cores available, {workers} used; per-thread rate "
                      f"{subs[0].n_variants / times[0]:.0f} variants/s (the compiled reference itself: ~1450/s per thread at H=64, BASELINE.md)"}


def _sweep_phase(name: str):
    """Phase (1, 2, 3) of a sweep kernel from its demangled name, None for other kernels:
    k_sweep<HP, R, VBUF, KEEPW, PHASE>, k_sweep_lean[_tri]<PHASE, R>, k_sweep_leanx<PHASE, HP>, k_sweep_small16[x]<PHASE>,
    k_sweep_generic<PHASE>."""
    import re
    if name.startswith("void k_sweep_lean2<") or name.startswith("k_sweep_leanx2("):  # phase 2 of triangle chains (biallelic / with multiallelic objects)
        return 2
    if name.startswith("k_sweep_leanx_tri(") or name.startswith("k_sweep_leanx_triw(") or name.startswith("k_sweep_tri1("):      # their phase 1 (not templates: no "void", no <PHASE>)
        return 1
    m = re.match(r"void k_sweep_lean<(\d), ", name) or re.match(r"void k_sweep_lean_tri<(\d), ", name) or \
        re.match(r"void k_sweep_leanx<(\d), ", name) or re.match(r"void k_sweep_small16x?<(\d)>", name) or \
        re.match(r"void k_sweep_generic<(\d)>", name) or re.match(r"void k_sweep<\d+, \d+, \d+, \w+, (\d)>", name)
    return int(m.group(1)) if m else None


def profiled_traffic(workload, kernel_phase):
    """HBM bytes per pass of one sweep phase from the committed rocprofv3 PMC summary
    (profiles/rNN_<workload>_summary.json, made by tools/summarize_profile.py from separate
    --pmc FETCH_SIZE / WRITE_SIZE passes).  Phase 1 is one launch of the phase-1 sweep kernel(s); phase 2 is
    one launch of the fused phase-2 sweep, or all store-only chunk launches (phase 3) plus their k_post
    launches (chunked mode), summed and divided by the number of passes the profile ran.
    None if no profile of this workload is committed."""
    cands = sorted((ROOT / "profiles").glob(f"r*_{workload}_summary.json"))
    if not cands:
        return None, None
    data = json.loads(cands[-1].read_text())
    ks = data["kernels"]
    passes = max([v.get("pmc_launches", 0) for n, v in ks.items() if _sweep_phase(n) == 1] or [0])
    if passes == 0:
        return None, None
    total = 0.0
    for name, v in ks.items():
        ph = _sweep_phase(name)
        mine = (ph == kernel_phase) or (kernel_phase == 2 and (ph == 3 or name.startswith("k_post(")))
        if mine:
            total += v.get("hbm_read_bytes_total", 0.0) + v.get("hbm_write_bytes_total", 0.0)
    return (total / passes if total > 0 else None), cands[-1].name


# Seams of the device platform: what a test replaces to drive main()'s N > 1 control flow on CPU ranks (gloo) with the job
# and the gather stubbed (tests/test_bench_world2.py) — the product run never touches them.
BACKEND = "nccl"


def _cuda_ok():
    import torch
    return torch.cuda.is_available()


def _cuda_set(local_rank):
    import torch
    torch.cuda.set_device(local_rank)


def _sync():
    import torch
    torch.cuda.synchronize()


def _make_device(local_rank):
    import torch
    return torch.device("cuda", local_rank)


def _device_view(ptr, n, typestr, dev):
    """zero-copy torch view of a device range owned by the job"""
    import torch
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=dev)


def _make_abi_gather(rank, world, local_rank):
    from pangenie_amd.dist import AbiGather
    return AbiGather(rank, world, local_rank)


def _release_cache():
    hmm._lib.load_hip().pg_hmm_release_cache()


class _DevArray:
    """Zero-copy torch view of a device range owned by the job (CUDA array interface)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def job_info_of(job):
    mode, chunk_cols = job.sweep_mode()
    return {"mode": mode, "chunk_cols": chunk_cols, "tri_chains": job.triangle_chains(), "n_chains": job.n_chains,
            "device_bytes": job.device_bytes(), "upload_bytes": job.upload_bytes(), "index_ms": job.index_ms(), "plan": job.plan()}


def step_table(kms, ms_per_step, p1_bytes, col_read_bytes, in_bytes, out_bytes):
    """The WHOLE step against the roofline (roofline.achieved = sum_v B_v / wall): every kernel class of the
    step with its hipEvent time, its share of the step and the algorithmic bytes SURVEY's B(H, K, A) charges to it — the
    forward columns written once (phase 1) and read once (phase 2), the per-variant inputs 4K + 2H + 3A + 16 (the emission
    kernels) and outputs 8G + 8 (the bins; in the chunked mode k_post forms them inside the phase-2 class).  `step_frac` =
    all of them over the step's wall time: never above the best kernel's `frac`."""
    alg = {"k_prep": in_bytes, "k_compact": 0.0, "k_records": 0.0, "k_sweep_phase1": p1_bytes, "k_sweep_phase2": col_read_bytes, "k_bins": out_bytes}
    total_ms = sum(kms.values()) or 1e-9
    rows = []
    for k in ("k_prep", "k_compact", "k_records", "k_sweep_phase1", "k_sweep_phase2", "k_bins"):
        ms = kms.get(k, 0.0)
        gbs = alg[k] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        rows.append({"kernel": k, "ms": ms, "share": ms / total_ms, "algorithmic_bytes": alg[k], "GBs": gbs, "frac": gbs / HBM_PEAK_GBS})
    bytes_step = p1_bytes + col_read_bytes + in_bytes + out_bytes
    step_gbs = bytes_step / (ms_per_step * 1e-3) / 1e9 if ms_per_step > 0 else 0.0
    return {"step_achieved": step_gbs, "step_frac": step_gbs / HBM_PEAK_GBS, "step_algorithmic_bytes": bytes_step,
            "step_ms": ms_per_step, "kernel_table": rows}


def roofline_of(results, batches, kms, H, workload_name, info, ms_per_step=None):
    """Roofline of the dominant sweep launch: algorithmic bytes per launch / hipEvent time — and of the whole
    step (step_table).  results: the fetched ContigResult of every chain (kept flags); info: job_info_of(job);
    ms_per_step: wall time of a step (default: the sum of the kernel classes)."""
    ncol, bytes_total, out_bytes = 0, 0, 0
    for r, bt in zip(results, batches):
        kp = r.kept
        ncol += int(kp.sum())
        bytes_total += algorithmic_bytes(bt, kp)
        A = np.diff(bt.allele_off.astype(np.int64))
        out_bytes += int((8 * (A * (A + 1) // 2) + 8).sum())
    # sweep phase 1 writes every kept column once (8*H^2 B), phase 2 reads it once; the
    # per-variant inputs/outputs (4K+2H+3A+16+8G+8 B) are charged to phase 2.
    p1_bytes = 8.0 * H * H * ncol
    p2_bytes = bytes_total - p1_bytes
    mode, chunk_cols = info["mode"], info["chunk_cols"]
    if mode == "chunked":
        # phase 2 is ~2*n_chunks short launches (store-only chunks + k_post); the dominant single
        # kernel launch — the one rocprofv3 --stats lists once per pass — is the phase-1 sweep
        dom = "k_sweep_phase1"
    else:
        dom = max(("k_sweep_phase1", "k_sweep_phase2"), key=lambda k: kms.get(k, 0.0))
    dom_bytes = p1_bytes if dom == "k_sweep_phase1" else p2_bytes
    dom_ms = kms.get(dom, 0.0)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    sweep_ms = kms.get("k_sweep_phase1", 0.0) + kms.get("k_sweep_phase2", 0.0)
    traffic, traffic_src = (None, None)
    if workload_name:
        traffic, traffic_src = profiled_traffic(workload_name, 1 if dom == "k_sweep_phase1" else 2)
    extra = {}
    if info["tri_chains"] == info["n_chains"] and H == 64:
        # Every chain keeps its (symmetric) columns as upper triangles: 1152 16-byte units per column instead of 2048
        # (DESIGN.md 4/6).  The algorithmic bytes of THIS formulation are what `achieved` is taken on — the figure of
        # the full formulation (a whole column written once and read once) would put the rate above the
        # HBM peak; it is reported beside it.
        full_bytes = dom_bytes
        dom_bytes = dom_bytes - 8.0 * H * H * ncol + 1152 * 16.0 * ncol
        bytes_total = bytes_total - 2 * 8.0 * H * H * ncol + 2 * 1152 * 16.0 * ncol
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        extra = {"triangle_storage": True, "full_formulation_bytes_per_launch": full_bytes,
                 "full_formulation_GBs": full_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0,
                 "note": "columns are symmetric and stored as upper triangles (18 KB of 32 KB per column): algorithmic bytes and "
                         "PMC traffic are those of the triangle formulation"}
    col_bytes = (1152 * 16.0 if extra else 8.0 * H * H) * ncol        # one column pass in this formulation's storage
    in_bytes = float(bytes_total) - 2.0 * col_bytes - out_bytes        # 4K + 2H + 3A + 16 per variant
    whole = step_table(kms, ms_per_step if ms_per_step else sum(kms.values()), col_bytes, col_bytes, in_bytes, float(out_bytes))
    return {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms, **extra, **whole,
            "sweep_GBs": (bytes_total / (sweep_ms * 1e-3) / 1e9) if sweep_ms > 0 else 0.0,
            "phase2_ms": kms.get("k_sweep_phase2", 0.0),
            "phase2_traffic": (profiled_traffic(workload_name, 2)[0] if workload_name else None)}, ncol, (mode, chunk_cols)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="genome24_h64", choices=sorted(WORKLOADS))
    ap.add_argument("--variants", type=int, default=0, help="override the variant count (debug)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="variants in the CPU-baseline sample (0 = auto ~10-20 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cohort", action="store_true", help="skip the cohort sub-measurement")
    ap.add_argument("--cohort-only", action="store_true", help="profiling: only the cohort measurement")
    ap.add_argument("--cohort-key", default="cohort", choices=["cohort", "panels_h16"] + sorted(COHORTS_MORE), help="with --cohort-only: which cohort (profiling)")
    ap.add_argument("--no-sampler", action="store_true", help="skip the HaplotypeSampler sub-measurement")
    ap.add_argument("--no-viterbi", action="store_true", help="skip the Viterbi phasing sub-measurement")
    ap.add_argument("--no-dropin", action="store_true", help="skip the threaded one-shot (drop-in) sub-measurement")
    ap.add_argument("--cohort-samples", type=int, default=COHORT["samples"])
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not _cuda_ok():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    _cuda_set(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(BACKEND, rank=rank, world_size=world)
    dev = _make_device(local_rank)

    def fence():
        _sync()
        if world > 1:
            dist.barrier()
        _sync()

    def max_over_ranks(x):
        if world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    table = hmm.ProbabilityTable(*default_table_args())
    params = hmm.make_params(1.26, False, 1e-5)  # what run_genotyping passes (reference src/commands.cpp:160)
    out = {"metric": "genotyped variants/sec (whole node) at H haplotypes; HBM GB/s vs roofline", "unit": "variants/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
           "vs_baseline": None, "dtype": "f64", "data": "synthetic"}

    # ------------------------------------------------------------------ main workload
    w = WORKLOADS[args.workload]
    V = args.variants or w["V"]
    H, K = w["H"], w["K"]
    n_chains = int(w.get("chains", 1))
    sizes = genome_contig_sizes(V) if w.get("genome") else [V] * n_chains
    V_total = sum(sizes)
    if not args.cohort_only:
        from pangenie_amd.dist import assign_chains, gather_posteriors
        # chains are sharded over the ranks (every rank computes the same plan); chain i is the same
        # synthetic contig whatever the number of GPUs
        plan = assign_chains([float(s) * H * H for s in sizes], world)
        if world == 1 and w.get("share_of"):  # one GPU's share of a workload that needs several: rank 0's chains of the share_of-rank plan
            plan = [assign_chains([float(s) * H * H for s in sizes], int(w["share_of"]))[0]]
            V_total = sum(sizes[i] for i in plan[0])
        mine = plan[rank]
        batches = [synthetic_panel(sizes[i], H, K, seed=12345 + 1000 * i, multiallelic_frac=w["multi"]) for i in mine]
        job = hmm.Job(batches, table, params, device=local_rank) if mine else None
        hs = job.host_seconds() if job else {"alloc_s": 0.0, "upload_s": 0.0}
        n_lik = [0] * n_chains
        local = {}
        if job:
            d_lik, d_exp, n_tot = job.packed_results()
            for k, i in enumerate(mine):
                n_lik[i] = job.device_results(k)[1]
        if world > 1:
            t = torch.tensor(n_lik, dtype=torch.int64, device=dev)
            dist.all_reduce(t)
            n_lik = [int(x) for x in t]
            # zero-copy views of the job's packed result ranges, cut per chain for the exchange
            if job:
                lik_t = _device_view(d_lik, n_tot, "<f8", dev)
                exp_t = _device_view(d_exp, n_tot, "<i4", dev)
                off = 0
                for i in mine:
                    local[i] = (lik_t[off:off + n_lik[i]], exp_t[off:off + n_lik[i]])
                    off += n_lik[i]

        # the ONE exchange of a sharded run: every rank's packed posteriors to rank 0.  Through the C ABI
        # (pg_comm_init + pg_hmm_gather: grouped RCCL sends, the path a host without torch takes); if that cannot be
        # set up on this node the torch.distributed form of the same exchange runs instead and the line says so.
        per_rank = [int(sum(n_lik[i] for i in chains)) for chains in plan]
        abi, gather_kind = None, "none (one GPU)"
        if world > 1 or os.environ.get("PG_BENCH_FORCE_GATHER"):
            try:
                abi = _make_abi_gather(rank, world, local_rank)
                abi.gather(job, per_rank)
                ok = 1.0
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] pg_hmm_gather unavailable ({e}); using the torch.distributed exchange", file=sys.stderr)
                abi, ok = None, 0.0
            if world > 1:  # every rank must take the same path
                tt = torch.tensor([ok], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MIN)
                if float(tt.item()) < 1.0:
                    if abi:
                        abi.close()
                    abi = None
            gather_kind = "pg_hmm_gather (C ABI: grouped ncclSend / ncclRecv)" if abi else "torch.distributed batch_isend_irecv"

        rank_ms = {"run": 0.0, "gather": 0.0}   # this rank's own wall time per step (run = its chains; gather = the ONE exchange)

        def step(timed=False):
            t_a = time.perf_counter()
            if job:
                job.run()
            t_b = time.perf_counter()
            if abi:
                abi.gather(job, per_rank)
            elif world > 1:
                gather_posteriors(local, n_lik, plan, dst=0, unpack=False, device=dev)
            if timed:
                if world > 1:
                    _sync()
                rank_ms["run"] += (t_b - t_a) * 1e3
                rank_ms["gather"] += (time.perf_counter() - t_b) * 1e3

        for _ in range(args.warmup):
            step()
        kms = {}
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(timed=True)
            if job:
                for k, v in job.kernel_ms().items():
                    kms[k] = kms.get(k, 0.0) + v
        fence()
        dt = max_over_ranks(time.perf_counter() - t0)
        kms = {k: v / args.steps for k, v in kms.items()}
        # per rank: its chains, its variants, its own run time and the time it spent in the exchange (the scaling curve's anatomy)
        mine_row = {"rank": rank, "chains": len(mine), "variants": int(sum(sizes[i] for i in mine)), "longest_chain": int(max([sizes[i] for i in mine] or [0])),
                    "run_ms_per_step": rank_ms["run"] / max(args.steps, 1), "gather_ms_per_step": rank_ms["gather"] / max(args.steps, 1)}
        rows = [mine_row]
        if world > 1:
            keys = ["chains", "variants", "longest_chain", "run_ms_per_step", "gather_ms_per_step"]
            mine_t = torch.tensor([float(mine_row[k]) for k in keys], dtype=torch.float64, device=dev)
            got = [torch.zeros_like(mine_t) for _ in range(world)]
            dist.all_gather(got, mine_t)   # (five numbers per rank, the same collective path as the timing reductions)
            rows = []
            for r, t_r in enumerate(got):
                vals = t_r.tolist()
                rows.append({"rank": r, "chains": int(vals[0]), "variants": int(vals[1]), "longest_chain": int(vals[2]),
                             "run_ms_per_step": vals[3], "gather_ms_per_step": vals[4]})

        # end to end: host buffers -> H2D of every input -> run -> D2H of every result, into the resident arena and into
        # result buffers the host already holds (allocated — and touched — once, outside the timed region)
        results = [hmm.ContigResult(b) for b in batches] if job else []
        job_info = job_info_of(job) if job else None
        if job:
            job.fetch_all(results)
        e2e_rounds, dt_e2e = 2, 0.0
        hs2 = {"upload_s": 0.0, "run_s": 0.0, "fetch_s": 0.0}
        for _ in range(e2e_rounds):
            fence()
            t0 = time.perf_counter()
            if job:
                job.upload_run()   # (pg_job_upload + pg_job_run in one call: the short chains' inputs cross PCIe behind the long chains' phase 1)
                job.fetch_all(results)
            if abi:
                abi.gather(job, per_rank)
            elif world > 1:
                gather_posteriors(local, n_lik, plan, dst=0, unpack=False, device=dev)
            fence()
            dt_e2e += max_over_ranks(time.perf_counter() - t0) / e2e_rounds
            if job:
                for k, v in job.host_seconds().items():
                    if k in hs2:
                        hs2[k] += v / e2e_rounds

        # the drop-in as the reference drives it (src/commands.cpp:949-978): one one-shot call per contig, all at once
        # from worker threads, host buffers in, host buffers out; the library merges the calls into one device job
        dropin = None
        if job and world == 1 and not args.no_dropin:
            job.close()
            job = None
            into = results
            for _ in range(2):  # warm-up: the arena pool fills (a device allocation of this size costs seconds: first-touch mapping)
                hmm.genotype_contigs_threaded(batches, table, params, device=local_rank, into=into)
            st0 = hmm.coalesce_stats()
            d_rounds, round_ms = 3, []
            fence()
            t0 = time.perf_counter()
            for _ in range(d_rounds):
                t1 = time.perf_counter()
                got = hmm.genotype_contigs_threaded(batches, table, params, device=local_rank, into=into)
                round_ms.append((time.perf_counter() - t1) * 1e3)
                bad = [g for g in got if isinstance(g, Exception)]
                if bad:
                    raise bad[0]
            fence()
            dt_d = (time.perf_counter() - t0) / d_rounds
            st1 = hmm.coalesce_stats()
            dropin = {"workload": f"{args.workload} as {len(batches)} concurrent one-shot pg_hmm_genotype_contig calls from {len(batches)} host threads "
                                  "(the reference's thread-pool pattern), H2D and D2H inside every call",
                      "value": V_total / dt_d, "unit": "variants/s", "ms_per_round": dt_d * 1e3, "rounds": d_rounds, "round_ms": round_ms,
                      "device_jobs_per_round": (st1["merged_jobs"] - st0["merged_jobs"]) / d_rounds,
                      "calls_per_round": (st1["calls"] - st0["calls"]) / d_rounds,
                      "note": "steady state: device arenas come from the library's pool (first round excluded)"}

        if rank == 0:
            roof, ncol, (mode, chunk_cols) = roofline_of(results, batches, kms, H, args.workload if (V == w["V"] and world == 1) else None, job_info,
                                                         ms_per_step=dt / args.steps * 1e3)
            out.update({
                "value": V_total * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                # BASELINE configs[3] as written: the total work is fixed, the chains are sharded over the ranks (one GPU: nothing to scale)
                "scaling": "strong" if world > 1 else "single",
                # the call-inclusive rate SURVEY.md §8(d) names (host buffers in, host results out, H2D and D2H inside): `value` stays
                # the resident rate the bench contract prescribes (inputs in HBM when the timed region starts)
                "value_call_inclusive": V_total / dt_e2e,
                "value_end_to_end": V_total / dt_e2e,
                "end_to_end": {"ms": dt_e2e * 1e3, "h2d_ms": hs2["upload_s"] * 1e3, "run_ms": hs2["run_s"] * 1e3, "d2h_ms": hs2["fetch_s"] * 1e3,
                               "h2d_bytes": sum(job_info["upload_bytes"].values()),
                               "note": "rank 0's share; arena resident (device allocation at job creation: alloc_s)"},
                "config": {"workload": f"{args.workload}: {w['cfg']}; {V_total} variants x {H} haplotypes x {K} k-mers/variant in {len(mine) if w.get('share_of') and world == 1 else n_chains} chain(s) "
                                       f"(longest {max(sizes)}), seeds 12345+1000*chain; " +
                                       (f"rank 0's share of the {w['share_of']}-GPU plan (LPT)" if w.get("share_of") and world == 1 else f"sharded over {world} GPU(s) by LPT"),
                           "variants": V_total, "haplotypes": H, "kmers_per_variant": K, "chains": n_chains,
                           "chains_on_rank0": len(mine), "kept_columns_rank0": ncol, "workgroups_per_chain": 2,
                           "parallelism": f"contig-sharded x{world}", "sweep_mode": "%s (chunk_cols=%d)" % (mode, chunk_cols),
                           "gather": gather_kind},
                "per_rank": rows, "index_pass_ms": job_info.get("index_ms"), "plan": job_info.get("plan"),
                "roofline": roof, "kernel_ms": kms, "device_bytes": job_info["device_bytes"],
                "alloc_s": hs["alloc_s"], "upload_s": hs["upload_s"],
            })
            if dropin:
                out["dropin_threads"] = dropin
            if not args.no_cpu_baseline and world == 1:  # reported baseline: rank 0 at N=1 only
                # ~10-20 s of single-thread CPU work: the port runs ~6k variants/s at H=64, ~60k at H=16
                auto = {16: 600_000, 64: 60_000, 128: 12_000}.get(H, 20_000)
                out["cpu_baseline"] = cpu_baseline(batches, H, args.cpu_sample or auto)
        if job:
            job.close()
        if abi:   # the main workload's communicator and rank 0's gathered block: not needed by the cohort lines
            abi.close()
            abi = None
        del batches
        _release_cache()

    # ------------------------------------------------------------------ cohort sub-measurements
    def cohort_measure(c, S, key, profile_name):
        """many (sample x contig) chains over ONE shared index (pg_cohort_new): the regime in which the sweeps are
        bound by the memory system.  c: dict(contigs, V, H, K, multi, distinct)."""
        NC, Hc = c["contigs"], c["H"]
        index = [synthetic_panel(c["V"], Hc, c["K"], seed=777 + i, multiallelic_frac=c.get("multi", 0.0), wide_frac=c.get("wide", 0.0)) for i in range(NC)]
        distinct = min(S, c.get("distinct", S))   # count sets formed; samples beyond reuse them in turn
        pool = []
        for s in range(distinct):  # every rank genotypes its own samples (weak scaling)
            kcs, covs = zip(*[synthetic_sample_counts(ix, seed=100_000 * (rank + 1) + 100 * s + i) for i, ix in enumerate(index)])
            pool.append((list(kcs), list(covs)))
        samples = [pool[s % distinct] for s in range(S)]
        cjob = hmm.Job.cohort(index, samples, table, params, device=local_rank)
        # (three warm-up steps: the first steps on a freshly allocated 100+ GB arena — right after the previous cohort's was
        #  freed — have been seen to run their store phase 60 % slower than every later one: `step_ms` shows each timed step)
        csteps, cwarm = max(2, min(args.steps, 5)), 3
        for _ in range(cwarm):
            cjob.run()
        ckms, step_ms = {}, []
        fence()
        t0 = time.perf_counter()
        for _ in range(csteps):
            t1 = time.perf_counter()
            cjob.run()
            step_ms.append((time.perf_counter() - t1) * 1e3)
            for k, v in cjob.kernel_ms().items():
                ckms[k] = ckms.get(k, 0.0) + v
        fence()
        cdt = max_over_ranks(time.perf_counter() - t0)
        ckms = {k: v / csteps for k, v in ckms.items()}
        fence()
        t0 = time.perf_counter()
        cjob.upload()  # the next batch of samples, one after the other: counts only, the index stays resident
        cjob.run()
        fence()
        cdt_serial = max_over_ranks(time.perf_counter() - t0)
        # ... and as a pipeline (pg_job_upload_begin / _end): batch n + 1 is packed into pinned staging and copied into the
        # job's second set of per-sample arrays WHILE batch n is genotyped; steady state over `csteps` batches
        cjob.upload_begin()
        cjob.run()
        cjob.upload_end()
        fence()
        t0 = time.perf_counter()
        wait_s = 0.0
        for _ in range(csteps):
            cjob.upload_begin()
            cjob.run()
            cjob.upload_end()
            wait_s += cjob.host_seconds()["upload_s"]
        fence()
        cdt_up = max_over_ranks(time.perf_counter() - t0) / csteps
        cjob.run()   # (pg_job_upload_end invalidated the last results: the ones fetched below)
        res = None
        if rank == 0:
            cb = cjob.batches
            croof, cncol, (cmode, _) = roofline_of(cjob.fetch_all(), cb, ckms, Hc, profile_name, job_info_of(cjob), ms_per_step=cdt / csteps * 1e3)
            cv = S * NC * c["V"]
            ub = cjob.upload_bytes()
            res = {
                "workload": f"{S} samples x {NC} contigs of {c['V']} variants, {Hc} haplotypes, {c['K']} k-mers/variant" +
                            (f", {int(100 * c['multi'])} % multiallelic" if c.get("multi") else "") +
                            (f", {int(100 * c['wide'])} % of the objects with 6-12 alleles (wide columns)" if c.get("wide") else "") + f" per GPU: "
                            f"{S * NC} chains over ONE shared index (pg_cohort_new)" +
                            (f"; {distinct} distinct count sets, reused in turn" if distinct < S else ""),
                "value": cv * world * csteps / cdt, "unit": "variants/s", "scaling": "weak", "steps": csteps, "warmup": cwarm, "ms_per_step": cdt / csteps * 1e3,
                "step_ms": [round(x, 2) for x in step_ms],
                "chains_per_gpu": S * NC, "sweep_mode": cmode, "kept_columns": cncol,
                "value_with_sample_upload": cv * world / cdt_up,   # every step with the NEXT batch's counts uploaded beside it
                "sample_upload": {"pipelined_ms_per_step": cdt_up * 1e3, "waited_for_upload_ms_per_step": wait_s / csteps * 1e3,
                                  "serial_upload_then_run_ms": cdt_serial * 1e3, "value_serial": cv * world / cdt_serial,
                                  "note": "pipelined = pg_job_upload_begin(next batch); pg_job_run(this batch); pg_job_upload_end: pinned staging, "
                                          "<= chains/64 H2D copies, second set of per-sample arrays on the device"},
                "h2d_bytes_per_sample_variant": ub["samples"] / float(cv),
                # what the index alone decides (column list, path alleles, transition constants) is formed once per uploaded index,
                # outside the step: its hipEvent time, and the kernels the library chose for this job
                "index_pass_ms": cjob.index_ms(), "plan": cjob.plan(),
                "roofline": croof, "kernel_ms": ckms, "device_bytes": cjob.device_bytes(),
            }
        cjob.close()
        _release_cache()
        return res

    def panel_job_measure(c):
        """chains with an index of their own each (what per-sample haplotype sampling produces), one resident job"""
        Hc = c["H"]
        distinct = [synthetic_panel(c["V"], Hc, c["K"], seed=9000 + 17 * i + 1000 * rank, multiallelic_frac=c["multi"], wide_frac=c["wide"]) for i in range(c["distinct"])]
        # consecutive chains (the four rows of a wave) get different panels
        pj_batches = [distinct[(i * 7 + i // c["distinct"]) % c["distinct"]] for i in range(c["chains"])]
        pjob = hmm.Job(pj_batches, table, params, device=local_rank)
        csteps = max(2, min(args.steps, 5))
        for _ in range(3):
            pjob.run()
        pk = {}
        fence()
        t0 = time.perf_counter()
        for _ in range(csteps):
            pjob.run()
            for k, v in pjob.kernel_ms().items():
                pk[k] = pk.get(k, 0.0) + v
        fence()
        pdt = max_over_ranks(time.perf_counter() - t0)
        pk = {k: v / csteps for k, v in pk.items()}
        res = None
        if rank == 0:
            proof, pncol, (pmode, _) = roofline_of(pjob.fetch_all(), pj_batches, pk, Hc, "panels_h16" if world == 1 else None, job_info_of(pjob),
                                                   ms_per_step=pdt / csteps * 1e3)
            pv = c["chains"] * c["V"]
            res = {"workload": f"{c['chains']} chains of {c['V']} variants, {Hc} paths (15 sampled + the reference path), {int(100 * c['multi'])} % multiallelic, "
                               f"{int(100 * c['wide'])} % of the objects with 6-12 alleles — every chain with an index of its OWN ({c['distinct']} distinct panels, "
                               "neighbouring chains differ): pg_job_new, no shared index",
                   "value": pv * world * csteps / pdt, "unit": "variants/s", "scaling": "weak", "steps": csteps, "ms_per_step": pdt / csteps * 1e3,
                   "chains_per_gpu": c["chains"], "sweep_mode": pmode, "kept_columns": pncol, "roofline": proof, "kernel_ms": pk, "device_bytes": pjob.device_bytes(),
                   # every chain of this job has an index of its OWN (a sampled panel per sample): the index pass — once per uploaded
                   # index — recurs with every batch of samples here, so the rate with it inside the step is the one that counts
                   "index_pass_ms": pjob.index_ms(), "value_incl_index_pass": pv * world / (pdt / csteps + 1e-3 * pjob.index_ms()), "plan": pjob.plan()}
        pjob.close()
        _release_cache()
        return res

    def cohort_strong_measure(key, like):
        """The cohort as a STRONG-scaling line (where north_star's >= 6x 1 -> 8 GPUs can land): a FIXED set
        of samples — COHORTS_MORE[key]'s, whatever N — sharded by sample over the ranks, every rank one resident cohort job
        over the shared index, ONE gather of the packed posteriors to rank 0 per step (pg_hmm_gather; torch.distributed if
        the C-ABI communicator cannot be made).  At N = 1 it IS the weak line `key` (`like`: no second run)."""
        c = COHORTS_MORE[key]
        S_total, NC, Hc = c["samples"], c["contigs"], c["H"]
        if world == 1:
            if like is None or rank != 0:
                return None
            return {"workload": like["workload"] + f"; the same {S_total} samples at every N, sharded by sample", "value": like["value"], "unit": "variants/s",
                    "scaling": "strong", "steps": like["steps"], "ms_per_step": like["ms_per_step"], "n_gpus": 1, "gather": "none (one GPU)",
                    "per_rank": [{"rank": 0, "samples": S_total, "chains": S_total * NC, "run_ms_per_step": like["ms_per_step"], "gather_ms_per_step": 0.0}],
                    "note": f"N = 1: the `{key}` measurement of this line (same job)"}
        S_mine = S_total // world + (1 if rank < S_total % world else 0)
        S_first = rank * (S_total // world) + min(rank, S_total % world)   # this rank's samples: global ids S_first .. S_first + S_mine - 1
        index = [synthetic_panel(c["V"], Hc, c["K"], seed=777 + i, multiallelic_frac=c.get("multi", 0.0), wide_frac=c.get("wide", 0.0)) for i in range(NC)]
        # the count set of a sample hangs on its GLOBAL id alone (set gid % distinct, seeded without the rank): the same samples
        # at every N, whichever rank holds them
        distinct = min(S_total, c.get("distinct", S_total))
        pool = {}
        for gid in range(S_first, S_first + S_mine):
            k = gid % distinct
            if k not in pool:
                kcs, covs = zip(*[synthetic_sample_counts(ix, seed=100_000 + 100 * k + i) for i, ix in enumerate(index)])
                pool[k] = (list(kcs), list(covs))
        cjob, build_err = None, None
        try:
            cjob = hmm.Job.cohort(index, [pool[gid % distinct] for gid in range(S_first, S_first + S_mine)], table, params, device=local_rank) if S_mine else None
        except Exception as e:  # noqa: BLE001 — agreed on below: no rank walks into a collective the others never reach
            build_err = e
        t_ok = torch.tensor([0.0 if build_err else 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        if float(t_ok.item()) < 1.0:
            if cjob:
                cjob.close()
            raise RuntimeError(f"cohort_strong: the job could not be built on every rank (this rank: {build_err!r})")
        n_mine = cjob.packed_results()[2] if cjob else 0
        tt = torch.zeros(world, dtype=torch.int64, device=dev)
        tt[rank] = n_mine
        dist.all_reduce(tt)
        per_rank_lik = [int(x) for x in tt]
        try:
            abi = _make_abi_gather(rank, world, local_rank)
            abi.gather(cjob, per_rank_lik)
            ok = 1.0
        except Exception as e:  # noqa: BLE001
            print(f"[rank {rank}] pg_hmm_gather unavailable ({e}); torch.distributed exchange", file=sys.stderr)
            abi, ok = None, 0.0
        t_ok = torch.tensor([ok], dtype=torch.float64, device=dev)
        dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        if float(t_ok.item()) < 1.0:
            if abi:
                abi.close()
            abi = None
        lik_t = exp_t = recv = None
        if abi is None:   # the same exchange through torch: every rank's two packed ranges to rank 0
            if cjob:
                d_lik, d_exp, n_tot = cjob.packed_results()
                lik_t = _device_view(d_lik, n_tot, "<f8", dev)
                exp_t = _device_view(d_exp, n_tot, "<i4", dev)
            if rank == 0:
                recv = {r: (torch.empty(per_rank_lik[r], dtype=torch.float64, device=dev), torch.empty(per_rank_lik[r], dtype=torch.int32, device=dev))
                        for r in range(1, world) if per_rank_lik[r]}

        def exchange():
            if abi:
                abi.gather(cjob, per_rank_lik)
                return
            ops = []
            if rank == 0:
                for r, (a, b) in recv.items():
                    ops += [dist.P2POp(dist.irecv, a, r), dist.P2POp(dist.irecv, b, r)]
            elif n_mine:
                ops = [dist.P2POp(dist.isend, lik_t, 0), dist.P2POp(dist.isend, exp_t, 0)]
            if ops:
                for wk in dist.batch_isend_irecv(ops):
                    wk.wait()

        csteps = max(2, min(args.steps, 3))
        if cjob:
            cjob.run()
        exchange()
        ms = {"run": 0.0, "gather": 0.0}
        fence()
        t0 = time.perf_counter()
        for _ in range(csteps):
            t_a = time.perf_counter()
            if cjob:
                cjob.run()
            t_b = time.perf_counter()
            exchange()
            _sync()
            ms["run"] += (t_b - t_a) * 1e3
            ms["gather"] += (time.perf_counter() - t_b) * 1e3
        fence()
        cdt = max_over_ranks(time.perf_counter() - t0)
        mine_t = torch.tensor([float(S_mine), ms["run"] / csteps, ms["gather"] / csteps], dtype=torch.float64, device=dev)
        got = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(got, mine_t)
        res = None
        if rank == 0:
            rows = [{"rank": r, "samples": int(g[0].item()), "chains": int(g[0].item()) * NC, "run_ms_per_step": float(g[1].item()), "gather_ms_per_step": float(g[2].item())}
                    for r, g in enumerate(got)]
            res = {"workload": f"{S_total} samples x {NC} contigs of {c['V']} variants, {Hc} haplotypes — the SAME {S_total} samples at every N, sharded by sample "
                               f"({S_total * NC} chains in all), one gather of the packed posteriors to rank 0 per step",
                   "value": S_total * NC * c["V"] * csteps / cdt, "unit": "variants/s", "scaling": "strong", "steps": csteps, "ms_per_step": cdt / csteps * 1e3,
                   "n_gpus": world, "gather": "pg_hmm_gather (C ABI: grouped ncclSend / ncclRecv)" if abi else "torch.distributed batch_isend_irecv",
                   "gathered_bytes_per_step": 12 * int(sum(per_rank_lik[1:])), "per_rank": rows}
        if abi:
            abi.close()
        if cjob:
            cjob.close()
        _release_cache()
        return res

    if not args.no_cohort:
        if args.cohort_only and args.cohort_key == "panels_h16":
            r = panel_job_measure(PANEL_JOB)
            if rank == 0:
                out["panels_h16"] = r
                out.update({"value": r["value"], "ms_per_step": r["ms_per_step"], "scaling": "weak", "config": {"workload": "panels_h16 only: " + r["workload"]}, "roofline": r["roofline"]})
        elif args.cohort_only and args.cohort_key != "cohort":   # profiling: one of the other cohorts alone
            spec = COHORTS_MORE[args.cohort_key]
            r = cohort_measure(spec, spec["samples"], args.cohort_key, args.cohort_key if world == 1 else None)
            if rank == 0:
                out[args.cohort_key] = r
                out.update({"value": r["value"], "ms_per_step": r["ms_per_step"], "scaling": "weak",
                            "config": {"workload": args.cohort_key + " only: " + r["workload"]}, "roofline": r["roofline"]})
        else:
            r = cohort_measure(COHORT, args.cohort_samples, "cohort", "cohort_h64" if world == 1 and args.cohort_samples == COHORT["samples"] else None)
            if rank == 0:
                out["cohort"] = r
                if args.cohort_only:
                    out.update({"value": r["value"], "ms_per_step": r["ms_per_step"], "scaling": "weak",
                                "config": {"workload": "cohort_h64 only: " + r["workload"]}, "roofline": r["roofline"]})
            if not args.cohort_only:
                # the other panel widths of BASELINE.json in the same regime: 16 haplotypes (configs[1]; k_sweep_small16 in
                # phase 1, class sums in phase 2) and 128 haplotypes with 20 % multiallelic objects (configs[4]; the general kernel)
                for key, spec in COHORTS_MORE.items():
                    try:
                        r = cohort_measure(spec, spec["samples"], key, key if world == 1 else None)
                    except Exception as e:  # noqa: BLE001 — on one rank a sub-measurement must not take the main line with it
                        if world > 1:       # (ranks inside collectives: nothing to catch up with)
                            raise
                        print(f"cohort {key} failed: {e!r}", file=sys.stderr)
                        r = {"error": repr(e)}
                    if rank == 0:
                        out[key] = r
                r = panel_job_measure(PANEL_JOB)
                if rank == 0:
                    out["panels_h16"] = r
                # the strong-scaling cohort: the default production shape, a fixed set of samples sharded over the ranks
                try:
                    r = cohort_strong_measure("cohort_h16m", out.get("cohort_h16m") if rank == 0 else None)
                except Exception as e:  # noqa: BLE001 — a sub-measurement must not take the main line with it
                    print(f"[rank {rank}] cohort_strong failed: {e!r}", file=sys.stderr)
                    r = {"error": repr(e)}
                if rank == 0 and r:
                    out["cohort_strong"] = r

    # ------------------------------------------------------------------ HaplotypeSampler sub-measurement
    if not args.no_sampler and not args.cohort_only:
        from pangenie_amd import sampler as smp
        sp = SAMPLER
        panels = [synthetic_panel(sp["V"], sp["H"], 20, seed=4242 + 100 * rank + i, multiallelic_frac=0.2) for i in range(sp["contigs"])]
        for b in panels:
            b.kmer_count[::3] = 1  # spread the fractions of present k-mers (the emission costs) over their range
        smp.sample_contigs(panels[:1], 2, device=local_rank)  # warm-up: module load
        fence()
        t0 = time.perf_counter()
        sampled, _ = smp.sample_contigs(panels, sp["size"], device=local_rank)  # H2D + 15 passes + D2H
        fence()
        sdt = max_over_ranks(time.perf_counter() - t0)
        sms, skern = smp.last_ms()
        if rank == 0:
            cells = sp["contigs"] * sp["V"] * sp["H"] * sp["size"]
            sres = {"workload": f"{sp['contigs']} contigs x {sp['V']} variants, {sp['H']} panel paths, {sp['size']} Viterbi passes (pg_sampler_run_batch), per GPU",
                    "value": cells * world / (1e-3 * sum(sms)), "unit": "cells/s (paths x variants x passes, kernel time)", "scaling": "weak",
                    "value_end_to_end": cells * world / sdt, "ms_expand": sms[0], "ms_forward": sms[1], "ms_backtrack": sms[2],
                    "ns_per_column_pass": 1e6 * sms[1] / (sp["V"] * sp["size"]), "kernel_waves": skern,
                    "bound": "latency: one workgroup per contig and pass, dependent chain per column (DESIGN.md 4b)"}
            if not args.no_cpu_baseline:
                from oracle import pyoracle as orc  # checker / CPU baseline only
                sub = panels[0].slice(0, min(20_000, sp["V"]))
                t0 = time.perf_counter()
                want, _ = orc.sampler_run(sub, sp["size"])
                cdt_s = time.perf_counter() - t0
                sres["cpu_baseline"] = {"value": sub.n_variants * sp["H"] * sp["size"] / cdt_s, "unit": "cells/s", "cores": 1, "kind": "port",
                                        "sample": f"first {sub.n_variants} variants of contig 0, {sp['size']} passes", "ns_per_column_pass": 1e9 * cdt_s / (sub.n_variants * sp["size"])}
                # the first pass of a prefix equals the prefix of... nothing in general (Viterbi looks ahead), so the
                # check is a separate full comparison on a small contig
                small = panels[0].slice(0, 3000)
                got, _ = smp.sample_contigs([small], sp["size"], device=local_rank)
                ref, _ = orc.sampler_run(small, sp["size"])
                sres["matches_oracle"] = bool((got[0] == ref).all())
            out["sampler"] = sres

    # ------------------------------------------------------------------ Viterbi phasing sub-measurement (run_phasing)
    if not args.no_viterbi and not args.cohort_only:
        vp = VITERBI
        vpanels = [synthetic_panel(vp["V"], vp["H"], 20, seed=777 + 100 * rank + i) for i in range(vp["contigs"])]
        vprm = hmm.make_params(1.26, False, 1e-5, run_genotyping=False, run_phasing=True)
        vjob = hmm.Job(vpanels, table, vprm, device=local_rank)
        vjob.run()  # warm-up: module load, the transition constants of the kept columns (host, long double, once per index)
        fence()
        t0 = time.perf_counter()
        vjob.run()
        fence()
        vdt = max_over_ranks(time.perf_counter() - t0)
        vms = vjob.viterbi_ms()
        if rank == 0:
            vcols = [int(vjob.fetch(i).n_columns) for i in range(vp["contigs"])]
            vres = {"workload": f"{vp['contigs']} contigs x {vp['V']} variants, {vp['H']} selected paths, phasing only (run_phasing), per GPU",
                    "value": sum(vcols) * world / (1e-3 * vms), "unit": "columns/s (Viterbi kernels)", "scaling": "weak",
                    "value_end_to_end": sum(vcols) * world / vdt, "viterbi_ms": vms, "ns_per_column_of_a_chain": 1e6 * vms / max(vcols),
                    "bound": "latency: one workgroup per chain, dependent chain per column (DESIGN.md 4c)", "device_bytes": vjob.device_bytes()}
            if not args.no_cpu_baseline:
                from oracle import pyoracle as orc  # checker / CPU baseline only
                otab = orc.OracleTable(*default_table_args())
                oprm = orc.make_params(1.26, False, 1e-5, run_genotyping=False, run_phasing=True)
                sub = vpanels[0].slice(0, 300)
                t0 = time.perf_counter()
                orc.viterbi_contig(sub, otab, oprm, form=0)  # the reference's own O(H^4) loop
                cdt_v = time.perf_counter() - t0
                small = vpanels[0].slice(0, 5000)
                t0 = time.perf_counter()
                ref = orc.viterbi_contig(small, otab, oprm, form=1)  # the same maxima in O(H^2)
                cdt_f = time.perf_counter() - t0
                vres["cpu_baseline"] = {"value": sub.n_variants / cdt_v, "unit": "columns/s", "cores": 1, "kind": "port",
                                        "sample": f"first {sub.n_variants} variants of contig 0, the reference's O(H^4) scan (oracle form 0)",
                                        "o_h2_form_columns_per_s": small.n_variants / cdt_f}
                got = hmm.genotype_contig(small, table, vprm, device=local_rank)
                vres["matches_oracle"] = bool((got.haplotype_1 == ref.hap1).all() and (got.haplotype_2 == ref.hap2).all())
            out["viterbi"] = vres
        vjob.close()

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
