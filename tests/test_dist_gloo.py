"""N>1 path on CPU: chain sharding plan + the single gather of packed posteriors, run as two
gloo processes (the same code runs over nccl = RCCL on GPUs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pangenie_amd.dist import assign_chains, gather_haplotypes, gather_posteriors, pack_sizes


def test_lpt_plan_is_balanced_and_deterministic():
    w = [248, 242, 198, 190, 181, 170, 159, 145, 138, 133, 135, 133, 114, 107, 102, 90, 83, 80, 58, 64, 46, 50, 156, 57]
    plan = assign_chains(w, 8)
    assert sorted(i for p in plan for i in p) == list(range(24))
    loads = [sum(w[i] for i in p) for p in plan]
    assert max(loads) <= 1.15 * (sum(w) / 8)
    assert plan == assign_chains(w, 8)
    assert assign_chains([5, 1], 4) == [[0], [1], [], []]
    assert pack_sizes([3, 6], [[0], [1]]) == ([3, 6], 6)


def _fake(i, n_lik):
    rng = np.random.default_rng(100 + i)
    return rng.random(n_lik[i]), rng.integers(-17000, 5, size=n_lik[i]).astype(np.int32)


def _fake_haps(i, n_lik):
    rng = np.random.default_rng(500 + i)
    return rng.integers(0, 65536, size=n_lik[i]).astype(np.uint16), rng.integers(0, 65536, size=n_lik[i]).astype(np.uint16)


def _worker(rank, world, port, n_lik, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = assign_chains([a * 1.0 for a in n_lik], world)
    local = {}
    for i in plan[rank]:
        lik, ex = _fake(i, n_lik)
        local[i] = (torch.from_numpy(lik), torch.from_numpy(ex))
    got = gather_posteriors(local, n_lik, plan, dst=0, device=torch.device("cpu"))
    # the phasing results of the same plan (n_lik doubles as the variant counts of the fake chains)
    haps = gather_haplotypes({i: _fake_haps(i, n_lik) for i in plan[rank]}, n_lik, plan, dst=0, device=torch.device("cpu"))
    ok = True
    if rank == 0:
        ok = sorted(got) == list(range(len(n_lik))) and sorted(haps) == list(range(len(n_lik)))
        for i in range(len(n_lik)):
            lik, ex = _fake(i, n_lik)
            ok = ok and np.array_equal(got[i][0], lik) and np.array_equal(got[i][1], ex)
            h1, h2 = _fake_haps(i, n_lik)
            ok = ok and np.array_equal(haps[i][0], h1) and np.array_equal(haps[i][1], h2) and haps[i][0].dtype == np.uint16
    else:
        ok = got is None and haps is None
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, n_lik):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_lik, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    return results


def test_gather_two_ranks_gloo():
    assert _run(2, [150, 21, 99, 40, 3]) == {0: True, 1: True}


def test_gather_with_an_empty_rank_gloo():
    # more ranks than chains: rank 1 owns nothing and must still take part with a buffer on the
    # collective's device
    assert assign_chains([7.0], 2) == [[0], []]
    assert _run(2, [7]) == {0: True, 1: True}
