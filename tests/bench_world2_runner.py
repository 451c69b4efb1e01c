"""One rank of tests/test_bench_world2.py: bench.py's main() on a CPU rank (gloo) with the device seams, the job and the gather
replaced — the plan, the shard-by-sample of cohort_strong, the per_rank all_gather, the gather sizes and rank 0's JSON line
are bench.py's own code (every `world > 1` branch had never executed anywhere)."""
import importlib.util
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

from pangenie_amd import hmm  # noqa: E402


class _Res:
    def __init__(self, b):
        self.kept = np.ones(b.n_variants, dtype=np.uint8)
        self.n_columns = b.n_variants


class MockJob:
    """What bench.py asks of hmm.Job, without a device: packed results are host arrays whose 'device pointers' are their indices
    in a registry (_device_view below turns them back into tensors)."""
    registry = {}

    def __init__(self, batches, table, params, device=0, _cohort=None):
        self.index = list(batches)
        if _cohort is None:
            self.batches = self.index
        else:
            self.batches = [b.with_counts(kc, cov) for (kcs, covs) in _cohort for b, kc, cov in zip(self.index, kcs, covs)]
        self._n = [int(b.geno_off[-1]) for b in self.batches]
        self._lik = torch.arange(sum(self._n), dtype=torch.float64) + 1000.0 * int(os.environ.get("RANK", "0"))
        self._exp = torch.arange(sum(self._n), dtype=torch.int32)
        self._key = len(MockJob.registry) + 1
        MockJob.registry[self._key] = (self._lik, self._exp)
        self.runs = 0

    @classmethod
    def cohort(cls, index, samples, table, params=None, device=0):
        return cls(index, table, params, device, _cohort=list(samples))

    n_chains = property(lambda self: len(self.batches))

    def packed_results(self):
        return self._key, -self._key, sum(self._n)

    def device_results(self, k):
        return 0, self._n[k]

    def run(self, stream=0):
        self.runs += 1

    def kernel_ms(self):
        return {"k_prep": 0.1, "k_compact": 0.0, "k_records": 0.0, "k_sweep_phase1": 1.0, "k_sweep_phase2": 1.2, "k_bins": 0.1}

    def host_seconds(self):
        return {"alloc_s": 0.0, "upload_s": 0.0, "run_s": 0.0, "fetch_s": 0.0}

    def upload(self, samples=None):
        pass

    upload_begin = upload

    def upload_run(self):
        self.run()

    def upload_end(self):
        pass

    def upload_bytes(self):
        return {"index": 1, "samples": 1}

    def fetch_all(self, into=None):
        return [_Res(b) for b in self.batches]

    def fetch(self, i):
        return _Res(self.batches[i])

    def sweep_mode(self):
        return ("chunked", 4096) if len(self.batches) < 64 else ("fused", 0)

    def triangle_chains(self):
        return 0

    def device_bytes(self):
        return 1

    def index_ms(self):
        return 0.5

    def plan(self):
        return "mock"

    def close(self):
        pass


class MockGather:
    """pg_hmm_gather's contract (rank r contributes per_rank[r] bins, rank 0 receives them all) over gloo"""
    calls = 0

    def __init__(self, rank, world, local_rank):
        self.rank, self.world = rank, world

    def gather(self, job, per_rank):
        import torch.distributed as dist
        mine = int(job.packed_results()[2]) if job else 0
        assert mine == per_rank[self.rank], (mine, per_rank)
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        dist.all_gather(sizes, torch.tensor([mine], dtype=torch.int64))
        assert [int(x) for x in sizes] == list(per_rank)
        MockGather.calls += 1

    def close(self):
        pass


def _device_view(ptr, n, typestr, dev):
    lik, exp = MockJob.registry[abs(ptr)]
    return lik if ptr > 0 else exp


def main():
    bench.BACKEND = "gloo"
    bench._cuda_ok = lambda: True
    bench._cuda_set = lambda r: None
    bench._sync = lambda: None
    bench._make_device = lambda r: torch.device("cpu")
    bench._device_view = _device_view
    bench._release_cache = lambda: None
    if os.environ.get("PG_MOCK_GATHER", "abi") == "abi":
        bench._make_abi_gather = MockGather
    else:   # the torch.distributed fall-back of the same exchange
        def refuse(*a):
            raise RuntimeError("no communicator (test)")
        bench._make_abi_gather = refuse
    hmm.Job = MockJob
    bench.hmm.Job = MockJob
    # the shapes of the bench's lines, shrunk to what a CPU rank generates in a second
    bench.WORKLOADS["genome24_small"]["V"] = 600
    bench.COHORT.update(samples=3, contigs=2, V=200)
    for k, v in list(bench.COHORTS_MORE.items()):
        if k != "cohort_h16m":
            del bench.COHORTS_MORE[k]
    bench.COHORTS_MORE["cohort_h16m"].update(samples=5, contigs=2, V=150, distinct=3)
    bench.PANEL_JOB.update(chains=8, distinct=2, V=100)
    sys.argv = ["bench.py", "--gpus", os.environ["WORLD_SIZE"], "--steps", "2", "--warmup", "1", "--workload", "genome24_small",
                "--no-sampler", "--no-viterbi", "--no-cpu-baseline", "--cohort-samples", "3"]
    bench.main()
    if os.environ.get("PG_MOCK_GATHER", "abi") == "abi":
        assert MockGather.calls > 0


if __name__ == "__main__":
    main()
