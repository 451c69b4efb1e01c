"""bench.py's N > 1 control flow on two CPU ranks (gloo), the job and the gather stubbed at bench.py's platform seams
(tests/bench_world2_runner.py): the LPT plan, the per_rank all_gather, the gather sizes, cohort_strong's shard-by-sample and
rank 0's JSON line — none of which had ever executed before an 8-GPU lease.  No scaling number here."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("gather", ["abi", "torch"])
def test_bench_main_on_two_gloo_ranks(gather):
    env = dict(os.environ, PG_MOCK_GATHER=gather, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "tests" / "bench_world2_runner.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]   # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["unit"] == "variants/s"
    assert len(d["per_rank"]) == 2 and {row["rank"] for row in d["per_rank"]} == {0, 1}
    assert sum(row["chains"] for row in d["per_rank"]) == 24 and sum(row["variants"] for row in d["per_rank"]) == d["config"]["variants"]
    assert ("pg_hmm_gather" in d["config"]["gather"]) == (gather == "abi")
    assert d["roofline"]["step_frac"] <= max(row["frac"] for row in d["roofline"]["kernel_table"]) + 1e-12
    cs = d["cohort_strong"]
    assert "error" not in cs, cs
    assert cs["n_gpus"] == 2 and cs["scaling"] == "strong" and len(cs["per_rank"]) == 2
    assert sorted(row["samples"] for row in cs["per_rank"]) == [2, 3]   # five samples sharded by sample over two ranks
    assert ("pg_hmm_gather" in cs["gather"]) == (gather == "abi")
    assert d["cohort_h16m"]["scaling"] == "weak" and d["cohort"]["chains_per_gpu"] == 6
