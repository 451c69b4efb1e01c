"""Timing of the device Viterbi (run_phasing): resident jobs, kernels only (hipEvents) and wall."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

from pangenie_amd import hmm
from pangenie_amd.panel import default_table_args, synthetic_panel


def main():
    t = hmm.ProbabilityTable(*default_table_args())
    prm = hmm.make_params(1.26, False, 1e-5, run_genotyping=False, run_phasing=True)
    for name, shapes in (("chr22_h30", [(200_000, 30)]), ("chr22_h16", [(200_000, 16)]), ("chr22_h64", [(200_000, 64)]),
                         ("8contigs_h30", [(100_000, 30)] * 8)):
        batches = [synthetic_panel(V, H, 20, seed=900 + i) for i, (V, H) in enumerate(shapes)]
        job = hmm.Job(batches, t, prm)
        job.run()
        ms, wall = [], []
        for _ in range(3):
            t0 = time.perf_counter()
            job.run()
            wall.append(time.perf_counter() - t0)
            ms.append(job.viterbi_ms())
        cols = sum(int(job.fetch(i).n_columns) for i in range(len(batches)))
        longest = max(int(job.fetch(i).n_columns) for i in range(len(batches)))
        out = {"workload": name, "chains": len(batches), "columns": cols, "viterbi_ms": min(ms), "run_wall_ms": min(wall) * 1e3,
               "ns_per_column_of_longest_chain": min(ms) * 1e6 / max(longest, 1), "device_GB": job.device_bytes() / 1e9}
        print(json.dumps(out))
        job.close()


if __name__ == "__main__":
    main()
