"""Multi-GPU sharding of the genotyping path (one process per GPU, torch.distributed).

Chains (contig x path-subset) are independent — the reference already runs them as
independent thread-pool jobs (src/commands.cpp:955-978) — so they shard across ranks with
NO data-path collective.  The only exchange is the final collection of the packed posteriors
on rank 0: ONE gather (RCCL over xGMI on GPUs; gloo in the CPU tests).  Payload per variant is
8*G + 4 bytes (28 B biallelic), i.e. tens of MB per peer for a whole genome, each peer on its
own direct xGMI link to rank 0.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np


def assign_chains(weights: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of chains to ranks.  weight ~ C * H^2
    (columns x states).  Deterministic: every rank computes the same plan."""
    order = sorted(range(len(weights)), key=lambda i: (-float(weights[i]), i))
    load = [0.0] * world
    plan: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        plan[r].append(i)
        load[r] += float(weights[i])
    for p in plan:
        p.sort()
    return plan


def pack_sizes(n_lik: Sequence[int], n_var: Sequence[int], plan: List[List[int]]) -> Tuple[List[int], int]:
    """Per-rank packed length (f64 words: all lik of its chains, then all lik_exp) and the max."""
    per_rank = [int(sum(n_lik[i] + n_var[i] for i in chains)) for chains in plan]
    return per_rank, max(per_rank) if per_rank else 0


def gather_posteriors(local: Dict[int, Tuple["torch.Tensor", "torch.Tensor"]], n_lik: Sequence[int],
                      n_var: Sequence[int], plan: List[List[int]], dst: int = 0, unpack: bool = True):
    """One gather of every rank's packed posteriors to `dst`.

    unpack=False leaves the gathered packed buffers on dst's device (list of world tensors) — what a
    timed loop wants: like a single-GPU run it ends with the posteriors resident in HBM.

    local: {chain id: (lik f64 tensor [n_lik], lik_exp i32 tensor [n_var])} for this rank's chains,
    on the device the process group works with (cuda for nccl = RCCL, cpu for gloo).
    Returns on dst: {chain id: (lik float64 ndarray, lik_exp int32 ndarray)} for ALL chains; None elsewhere.
    """
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    per_rank, width = pack_sizes(n_lik, n_var, plan)
    mine = plan[rank]
    dev = next(iter(local.values()))[0].device if local else torch.device("cpu")
    buf = torch.zeros(max(width, 1), dtype=torch.float64, device=dev)
    off = 0
    for i in mine:  # lik blocks, then exponent blocks (int32 is exact in float64)
        buf[off:off + n_lik[i]] = local[i][0]
        off += n_lik[i]
    for i in mine:
        buf[off:off + n_var[i]] = local[i][1].to(torch.float64)
        off += n_var[i]
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    if not unpack:
        return out
    res = {}
    for r in range(world):
        flat = out[r].cpu().numpy()
        off = 0
        for i in plan[r]:
            res[i] = [flat[off:off + n_lik[i]].copy(), None]
            off += n_lik[i]
        for i in plan[r]:
            res[i][1] = np.rint(flat[off:off + n_var[i]]).astype(np.int32)
            off += n_var[i]
    return {i: (v[0], v[1]) for i, v in res.items()}
