"""Multi-GPU sharding of the genotyping path (one process per GPU, torch.distributed).

Chains (contig x path-subset) are independent — the reference already runs them as
independent thread-pool jobs (src/commands.cpp:955-978) — so they shard across ranks with
NO data-path collective.  The only exchange is the final collection of the packed posteriors
on rank 0: ONE batched group of point-to-point sends (RCCL over xGMI on GPUs; gloo in the CPU
tests).  Payload per genotype bin is 8 + 4 bytes (36 B per biallelic variant), i.e. tens of MB
per peer for a whole genome, each peer on its own direct xGMI link to rank 0.  The same exchange
exists behind the C ABI for hosts without torch (include/pangenie_hmm.h: pg_hmm_gather).
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


def pack_sizes(n_lik: Sequence[int], plan: List[List[int]]) -> Tuple[List[int], int]:
    """Genotype bins per rank (lik f64 + lik_exp i32 each) and the maximum over ranks."""
    per_rank = [int(sum(n_lik[i] for i in chains)) for chains in plan]
    return per_rank, max(per_rank) if per_rank else 0


def gather_posteriors(local: Dict[int, Tuple["torch.Tensor", "torch.Tensor"]], n_lik: Sequence[int],
                      plan: List[List[int]], dst: int = 0, unpack: bool = True, device=None):
    """ONE exchange of every rank's posteriors to `dst`: per peer one send of its lik (f64) and one of
    its lik_exp (i32), exactly as long as that rank's data (no widening, no padding to the longest
    rank), batched into a single group of point-to-point operations — on GPUs each peer uses its own
    xGMI link to dst.

    local: {chain id: (lik f64 tensor [n_lik], lik_exp i32 tensor [n_lik])} for this rank's chains.
    device: where the buffers of the collective live (cuda for nccl = RCCL, cpu for gloo); needed
            explicitly because a rank may own no chain at all.
    unpack=False leaves the gathered buffers on dst's device ({rank: (lik, lik_exp)}) — what a timed
    loop wants: like a single-GPU run it ends with the posteriors resident in HBM.
    Returns on dst: {chain id: (lik float64 ndarray, lik_exp int32 ndarray)} for ALL chains; None elsewhere.
    """
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    per_rank, _ = pack_sizes(n_lik, plan)
    if device is None:
        device = next(iter(local.values()))[0].device if local else torch.device("cpu")
    mine = plan[rank]
    if mine:
        lik = torch.cat([local[i][0].reshape(-1).to(torch.float64) for i in mine]) if len(mine) > 1 else local[mine[0]][0].reshape(-1)
        ex = torch.cat([local[i][1].reshape(-1).to(torch.int32) for i in mine]) if len(mine) > 1 else local[mine[0]][1].reshape(-1)
    else:
        lik = torch.empty(0, dtype=torch.float64, device=device)
        ex = torch.empty(0, dtype=torch.int32, device=device)
    got = None
    if rank == dst:
        got = {r: (torch.empty(per_rank[r], dtype=torch.float64, device=device),
                   torch.empty(per_rank[r], dtype=torch.int32, device=device)) for r in range(world) if r != dst}
        ops = []
        for r, (gl, ge) in got.items():
            if per_rank[r]:
                ops += [dist.P2POp(dist.irecv, gl, r), dist.P2POp(dist.irecv, ge, r)]
        got[dst] = (lik, ex)
    else:
        ops = [dist.P2POp(dist.isend, lik, dst), dist.P2POp(dist.isend, ex, dst)] if per_rank[rank] else []
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if rank != dst:
        return None
    if not unpack:
        return got
    res = {}
    for r in range(world):
        fl, fe = got[r][0].cpu().numpy(), got[r][1].cpu().numpy()
        off = 0
        for i in plan[r]:
            res[i] = (fl[off:off + n_lik[i]].copy(), fe[off:off + n_lik[i]].copy())
            off += n_lik[i]
    return res


def gather_haplotypes(local: Dict[int, Tuple["np.ndarray", "np.ndarray"]], n_variants: Sequence[int],
                      plan: List[List[int]], dst: int = 0, device=None):
    """The phasing results (run_phasing: `haplotype_1` / `haplotype_2` of every chain, u16 per variant) of a sharded
    run on `dst` — the same single group of point-to-point sends as gather_posteriors, one i32 per variant
    (haplotype_1 | haplotype_2 << 16: a dtype every backend carries).

    local: {chain id: (haplotype_1 uint16 [V], haplotype_2 uint16 [V])} for this rank's chains (numpy or tensors).
    Returns on dst: {chain id: (haplotype_1 uint16 ndarray, haplotype_2 uint16 ndarray)} for ALL chains; None elsewhere."""
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    per_rank = [int(sum(n_variants[i] for i in chains)) for chains in plan]
    if device is None:
        device = torch.device("cpu")

    def packed(i):
        h1 = np.asarray(local[i][0]).astype(np.int64).reshape(-1)
        h2 = np.asarray(local[i][1]).astype(np.int64).reshape(-1)
        assert h1.size == n_variants[i] and h2.size == n_variants[i]
        return torch.from_numpy(((h1 & 0xFFFF) | ((h2 & 0xFFFF) << 16)).astype(np.uint32).view(np.int32))
    mine = plan[rank]
    buf = torch.cat([packed(i) for i in mine]).to(device) if mine else torch.empty(0, dtype=torch.int32, device=device)
    got = None
    if rank == dst:
        got = {r: torch.empty(per_rank[r], dtype=torch.int32, device=device) for r in range(world) if r != dst}
        ops = [dist.P2POp(dist.irecv, g, r) for r, g in got.items() if per_rank[r]]
        got[dst] = buf
    else:
        ops = [dist.P2POp(dist.isend, buf, dst)] if per_rank[rank] else []
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if rank != dst:
        return None
    res = {}
    for r in range(world):
        flat = got[r].cpu().numpy().view(np.uint32)
        off = 0
        for i in plan[r]:
            seg = flat[off:off + n_variants[i]]
            res[i] = ((seg & 0xFFFF).astype(np.uint16), (seg >> 16).astype(np.uint16))
            off += n_variants[i]
    return res


class AbiGather:
    """The exchange behind the C ABI (include/pangenie_hmm.h: pg_comm_init + pg_hmm_gather — grouped RCCL point-to-point
    sends, what a host without torch uses): rank 0 makes the RCCL id, the existing process group hands it round, every
    rank opens its own communicator.  `world == 1`: a single-rank communicator (with PG_GATHER_LOOPBACK set the block still
    travels through ncclSend / ncclRecv).

        g = AbiGather(rank, world, device_index)      # collective: every rank
        g.gather(job, n_lik_per_rank)                 # collective; rank 0 ends with g.lik_all / g.exp_all (torch, on device)
    """

    def __init__(self, rank: int, world: int, device: int):
        import ctypes as C
        import torch
        from . import _lib
        self._C, self._lib, self.rank, self.world, self.device = C, _lib.load_hip(), rank, world, device
        ident = torch.zeros(128, dtype=torch.uint8)
        err = C.create_string_buffer(512)
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            rc = self._lib.pg_comm_unique_id(buf, err, 512)
            if rc:
                raise RuntimeError(f"pg_comm_unique_id: {err.value.decode()} ({rc})")
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        if world > 1:
            import torch.distributed as dist
            dev_ident = ident.to(torch.device("cuda", device))
            dist.broadcast(dev_ident, src=0)
            ident = dev_ident.cpu()
        buf = (C.c_uint8 * 128)(*[int(x) for x in ident.tolist()])
        h = C.c_void_p()
        rc = self._lib.pg_comm_init(buf, world, rank, device, C.byref(h), err, 512)
        if rc:
            raise RuntimeError(f"pg_comm_init: {err.value.decode()} ({rc})")
        self.h = h
        self.lik_all = self.exp_all = None

    def gather(self, job, n_lik_per_rank: Sequence[int], root: int = 0) -> None:
        import torch
        C = self._C
        total = int(sum(n_lik_per_rank))
        if self.rank == root and (self.lik_all is None or self.lik_all.numel() != total):
            dev = torch.device("cuda", self.device)
            self.lik_all = torch.empty(max(total, 1), dtype=torch.float64, device=dev)
            self.exp_all = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        plan = (C.c_uint64 * self.world)(*[int(x) for x in n_lik_per_rank])
        err = C.create_string_buffer(512)
        rc = self._lib.pg_hmm_gather(self.h, job.h if job is not None else None, root, plan,
                                     C.c_void_p(self.lik_all.data_ptr()) if self.rank == root else None,
                                     C.c_void_p(self.exp_all.data_ptr()) if self.rank == root else None, err, 512)
        if rc:
            raise RuntimeError(f"pg_hmm_gather: {err.value.decode()} ({rc})")

    def close(self) -> None:
        if getattr(self, "h", None):
            self._lib.pg_comm_destroy(self.h)
            self.h = None
