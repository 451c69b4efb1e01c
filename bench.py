#!/usr/bin/env python3
"""bench.py — throughput of the genotyping hot path (emissions + forward-backward HMM).

A "step" = one full pass of the device path (k_prep -> k_compact -> k_records -> k_sweep phase 1 ->
k_sweep phase 2 -> k_bins, plus the RCCL gather of the posteriors when N > 1) over one synthetic
contig batch that is already resident in HBM.  metric = genotyped variants/sec (whole job).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload chr22_h64|contig_h16|...]

N > 1 is launched by the driver with torch.distributed.run (one rank per GPU); contigs are
independent chains, so every rank genotypes its own contig (weak scaling, no data-path
collective) and rank 0 gathers the packed posteriors with one RCCL gather per step.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from pangenie_amd import hmm  # noqa: E402
from pangenie_amd.panel import algorithmic_bytes, default_table_args, synthetic_panel  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable

# BASELINE.json configs -> single-GPU shapes (variants, haplotypes, k-mers/variant, multiallelic)
WORKLOADS = {
    "contig_h16": dict(V=50_000, H=16, K=20, multi=0.0, cfg="configs[1]: 1 contig, 50k variants, 16 haplotypes, ~20 k-mers/var"),
    "chr22_h64": dict(V=200_000, H=64, K=20, multi=0.0, cfg="configs[2]: chr22-scale, 200k variants, 64 haplotypes"),
    "chr22_h128": dict(V=60_000, H=128, K=20, multi=0.2, cfg="configs[4] per-GPU slice: 128 haplotypes, 20% multiallelic"),
    # many independent chains on one GPU (sample x contig shards): the regime
    # in which the sweep is HBM-bound instead of per-column-latency-bound
    "cohort_h64": dict(V=16_000, H=64, K=20, multi=0.0, chains=256, cfg="256 chains (sample x contig shards) of 16k variants, 64 haplotypes"),
    "genome24_small": dict(V=40_000, H=64, K=20, multi=0.0, chains=24, cfg="configs[3] shape at 1/5 length, equal contigs: 24 contigs, 64 haplotypes, one GPU"),
    # BASELINE.json configs[3]: whole genome, 24 contigs with human-like length proportions, 5M variants,
    # 64 haplotypes.  164 GB of column slots: fits ONE 288 GB MI355X, so it is the single-GPU workload.
    "genome24_h64": dict(V=5_000_000, H=64, K=20, multi=0.0, chains=24, genome=True,
                         cfg="configs[3]: whole genome, 24 contigs (human chromosome length proportions), 5M variants, 64 haplotypes"),
}
# GRCh38 chromosome lengths (Mb) 1..22, X, Y: proportions of the 24 synthetic contigs
CONTIG_MB = [248, 242, 198, 190, 181, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57]


def cpu_baseline(batches, H, sample_variants):
    """Reported baseline only: the CPU oracle (our long-double port of the reference path) on a
    bounded sample of the same workload.  The reference runs one thread per contig x subset
    (src/commands.cpp:949-953), so a multi-contig workload is timed with one thread per contig
    (up to the host's cores), each on the first `sample_variants` variants of its own contig.
    Plain Python threads: the oracle is a C library without global state and ctypes drops the GIL
    for the duration of the call, so the threads run on separate cores."""
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
                      f"{subs[0].n_variants / times[0]:.0f} variants/s"}


def profiled_traffic(workload, kernel_phase):
    """HBM bytes per pass of one sweep phase from the committed rocprofv3 PMC summary
    (profiles/rNN_<workload>_summary.json, made by tools/summarize_profile.py from separate
    --pmc FETCH_SIZE / WRITE_SIZE passes).  Phase 1 is one launch of k_sweep<..., 1>; phase 2 is one
    launch of k_sweep<..., 2> (fused mode) or all chunk launches k_sweep<..., 3> plus their k_post
    launches (chunked mode), summed and divided by the number of passes the profile ran.
    None if no profile of this workload is committed."""
    cands = sorted((ROOT / "profiles").glob(f"r*_{workload}_summary.json"))
    if not cands:
        return None, None
    data = json.loads(cands[-1].read_text())
    ks = data["kernels"]
    passes = max([v.get("pmc_launches", 0) for n, v in ks.items() if n.startswith("void k_sweep<") and ", 1>(" in n] or [0])
    if passes == 0:
        return None, None
    total = 0.0
    for name, v in ks.items():
        sweep = name.startswith("void k_sweep<")
        mine = (sweep and f", {kernel_phase}>(" in name) or \
               (kernel_phase == 2 and ((sweep and ", 3>(" in name) or name.startswith("k_post(")))
        if mine:
            total += v.get("hbm_read_bytes_total", 0.0) + v.get("hbm_write_bytes_total", 0.0)
    return (total / passes if total > 0 else None), cands[-1].name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="genome24_h64", choices=sorted(WORKLOADS))
    ap.add_argument("--variants", type=int, default=0, help="override the variant count (debug)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="variants in the CPU-baseline sample (0 = auto ~10-20 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    w = WORKLOADS[args.workload]
    V = args.variants or w["V"]
    H, K = w["H"], w["K"]
    n_chains = int(w.get("chains", 1))
    if w.get("genome"):
        tot = float(sum(CONTIG_MB))
        sizes = [int(round(V * mb / tot)) for mb in CONTIG_MB]
    else:
        sizes = [V] * n_chains
    batches = [synthetic_panel(sizes[i], H, K, seed=12345 + rank + 1000 * i, multiallelic_frac=w["multi"]) for i in range(n_chains)]
    V_total = sum(sizes)
    batch = batches[0]
    table = hmm.ProbabilityTable(*default_table_args())
    params = hmm.make_params(1.26, False, 1e-5)  # what run_genotyping passes (reference src/commands.cpp:160)
    t_up = time.perf_counter()
    job = hmm.Job(batches, table, params, device=local_rank)
    upload_s = time.perf_counter() - t_up

    # gather plumbing: posteriors stay on the device; every rank owns its own n_chains chains (weak
    # scaling), rank 0 collects the packed (lik, lik_exp) of ALL chains of every rank with ONE RCCL
    # gather per step and keeps them in HBM (like the single-GPU run, the timed region ends with the
    # posteriors resident on a device, not on the host)
    hip = C.CDLL("libamdhip64.so")
    dev_res = [job.device_results(i) for i in range(n_chains)]
    plan = [[r * n_chains + i for i in range(n_chains)] for r in range(world)]
    all_lik, all_var, local_t = [], [], {}
    if world > 1:
        from pangenie_amd.dist import gather_posteriors
        mine = torch.tensor([[d[1], d[3]] for d in dev_res], dtype=torch.int64, device="cuda")
        allsz = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allsz, mine)
        for t in allsz:  # chain ids are rank-major, same order as `plan`
            all_lik += [int(x) for x in t[:, 0]]
            all_var += [int(x) for x in t[:, 1]]
        for i, (d_lik, n_lik, d_exp, n_var) in enumerate(dev_res):
            local_t[rank * n_chains + i] = (torch.empty(n_lik, dtype=torch.float64, device="cuda"),
                                            torch.empty(n_var, dtype=torch.int32, device="cuda"))

    def step():
        job.run()
        if world > 1:
            for i, (d_lik, n_lik, d_exp, n_var) in enumerate(dev_res):
                lt, et = local_t[rank * n_chains + i]
                hip.hipMemcpy(C.c_void_p(lt.data_ptr()), C.c_void_p(d_lik), C.c_size_t(n_lik * 8), 3)
                hip.hipMemcpy(C.c_void_p(et.data_ptr()), C.c_void_p(d_exp), C.c_size_t(n_var * 4), 3)
            gather_posteriors(local_t, all_lik, all_var, plan, dst=0, unpack=False)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    kms = {}
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k, v in job.kernel_ms().items():
            kms[k] = kms.get(k, 0.0) + v
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    res = job.fetch(0)
    kms = {k: v / args.steps for k, v in kms.items()}

    if rank == 0:
        total_variants = V_total * world * args.steps
        value = total_variants / dt
        # roofline of the dominant kernel (HBM-bound class), algorithmic bytes per launch
        kept = res.kept
        ncol = int(kept.sum())
        bytes_total = algorithmic_bytes(batch, kept)
        if n_chains > 1:  # all chains of the job run in the same launch
            ncol, bytes_total = 0, 0
            for i, bt in enumerate(batches):
                kp = job.fetch(i).kept
                ncol += int(kp.sum())
                bytes_total += algorithmic_bytes(bt, kp)
        # sweep phase 1 writes every kept column once (8*H^2 B), phase 2 reads it once; the
        # per-variant inputs/outputs (4K+2H+3A+16+8G+8 B) are charged to phase 2.
        p1_bytes = 8.0 * H * H * ncol
        p2_bytes = bytes_total - p1_bytes
        mode, chunk_cols = job.sweep_mode()
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
        if V == w["V"]:  # the committed profile is of this very command (same seeds, same chains)
            traffic, traffic_src = profiled_traffic(args.workload, 1 if dom == "k_sweep_phase1" else 2)
        out = {
            "metric": "genotyped variants/sec (whole node) at H haplotypes; HBM GB/s vs roofline",
            "value": value, "unit": "variants/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {w['cfg']}; {V_total} variants x {H} haplotypes x {K} k-mers/variant "
                                   f"per GPU in {n_chains} chain(s) (longest {max(sizes)}), seed 12345+rank",
                       "variants_per_gpu": V_total, "haplotypes": H, "kmers_per_variant": K,
                       "kept_columns": ncol, "chains_per_gpu": n_chains, "workgroups_per_chain": 2, "parallelism": f"contig-sharded x{world}",
                       "sweep_mode": "%s (chunk_cols=%d)" % (mode, chunk_cols)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms,
                         "sweep_GBs": (bytes_total / (sweep_ms * 1e-3) / 1e9) if sweep_ms > 0 else 0.0,
                         "phase2_ms": kms.get("k_sweep_phase2", 0.0),
                         "phase2_traffic": (profiled_traffic(args.workload, 2)[0] if V == w["V"] else None)},
            "kernel_ms": kms,
            "device_bytes": job.device_bytes(), "upload_s": upload_s,
        }
        if not args.no_cpu_baseline and world == 1:  # reported baseline: rank 0 at N=1 only
            # ~10-20 s of single-thread CPU work: the port runs ~6k variants/s at H=64, ~60k at H=16
            auto = {16: 600_000, 64: 60_000, 128: 12_000}.get(H, 20_000)
            out["cpu_baseline"] = cpu_baseline(batches, H, args.cpu_sample or auto)
        print(json.dumps(out))
    job.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
