#!/usr/bin/env python3
"""The CPU twin of tools/pipeline_check.sh: the same host pieces, the HMM by the long double oracle instead of the device
(test infrastructure: this is how the scale run's VCF is checked without a GPU).
    pipeline_cpu_check.py <index prefix> <reads> <out.vcf>"""
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from pangenie_amd import cereal_io                               # noqa: E402
from pangenie_amd.build import build_host, HOST_TEST             # noqa: E402
from pangenie_amd.genotyping_result import results_from_flat     # noqa: E402
from pangenie_amd.panel import flatten                           # noqa: E402
from oracle import pyoracle as orc                               # noqa: E402


def main(prefix, reads, out):
    build_host()
    t = time.time()
    r = subprocess.run([str(HOST_TEST), "counts", prefix, reads, "8"], capture_output=True, text=True, check=True)
    peak = int(r.stdout.strip().split("=")[1])
    print(f"peak {peak}; counts {time.time() - t:.1f} s")
    return_peak = peak
    counted = cereal_io.load(prefix + "_counted_UniqueKmersMap.cereal")
    res = cereal_io.Results()
    t = time.time()
    for chrom, objects in counted.unique_kmers.items():
        batch = flatten(objects)
        if batch.n_paths > 100:   # the reference's default: 15 sampled haplotypes (+ the reference path) — oracle sampler
            import numpy as np
            sampled, _ = orc.sampler_run(batch, 15, 1.26, np.longdouble("0.01"), 5)
            if counted.add_reference:
                sampled = np.vstack([sampled, np.zeros((1, batch.n_variants), np.uint32)])
            batch = batch.update_paths(sampled)
        ref = orc.genotype_contig(batch, orc.OracleTable(peak // 4, peak * 4, 2 * peak, 0.01), orc.make_params(1.26, False, 1e-5))
        results = results_from_flat(batch, ref.lik, ref.kept, ref.allele_present, ref.n_kmers, ref.coverage)
        for g in results:
            g.normalize()
        res.result[chrom] = results
        res.runtimes[chrom] = 0.0
    print(f"oracle HMM {time.time() - t:.1f} s")
    archive = out + ".results.cereal"
    Path(archive).write_bytes(cereal_io.dumps_results(res))
    subprocess.run([str(HOST_TEST), "vcf", prefix, archive, out], check=True)
    return return_peak


if __name__ == "__main__":
    main(*sys.argv[1:4])
