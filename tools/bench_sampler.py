"""HaplotypeSampler timing on the device: ms per pass, ns per column, against the CPU oracle on a
bounded sample.  usage: python tools/bench_sampler.py [--variants V] [--paths H] [--size S] [--contigs G]
(TEST/MEASUREMENT TOOL: the oracle is only the timed CPU baseline and the checker here.)"""
import argparse, json, os, sys, time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pangenie_amd import sampler as smp  # noqa: E402
from pangenie_amd.panel import synthetic_panel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=200_000)
    ap.add_argument("--paths", type=int, default=215)
    ap.add_argument("--size", type=int, default=15)
    ap.add_argument("--contigs", type=int, default=1)
    ap.add_argument("--cpu-variants", type=int, default=20_000)
    ap.add_argument("--check", action="store_true", help="compare the full result with the oracle")
    a = ap.parse_args()
    batches = [synthetic_panel(a.variants, a.paths, 20, seed=11 + g, multiallelic_frac=0.2) for g in range(a.contigs)]
    for b in batches:
        b.kmer_count[::3] = 1  # spread the 'present' fractions
    t0 = time.perf_counter()
    sampled, best = smp.sample_contigs(batches, a.size)
    wall = time.perf_counter() - t0
    ms, kern = smp.last_ms()
    cols = a.variants * a.contigs
    out = {"variants": a.variants, "paths": a.paths, "size": a.size, "contigs": a.contigs, "kernel": kern,
           "ms_expand": ms[0], "ms_forward": ms[1], "ms_backtrack": ms[2], "wall_s": wall,
           "ns_per_column_pass": 1e6 * (ms[1] + ms[2] + ms[0]) / (a.variants * a.size),
           "ns_per_column_pass_forward": 1e6 * ms[1] / (a.variants * a.size),
           "cells_per_s": cols * a.paths * a.size / (1e-3 * sum(ms)),
           "switches_pass0": int((np.diff(sampled[0][0].astype(np.int64)) != 0).sum())}
    from oracle import pyoracle as orc
    sub = batches[0].slice(0, min(a.cpu_variants, a.variants))
    t0 = time.perf_counter()
    want, wbest = orc.sampler_run(sub, a.size)
    cpu = time.perf_counter() - t0
    out["cpu_oracle_ns_per_column_pass"] = 1e9 * cpu / (sub.n_variants * a.size)
    out["cpu_oracle_variants"] = sub.n_variants
    if a.check:
        w, wb = orc.sampler_run(batches[0], a.size)
        out["matches_oracle"] = bool(np.array_equal(w, sampled[0]) and np.array_equal(wb, best[0]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
