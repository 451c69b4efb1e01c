#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header -x -k "leanx or multi_contig or config4_shape or wide_columns or panels_vs_oracle or generic_kernel_cross or many_alleles" 2>&1 | tail -3
python bench.py --workload genome24_small --steps 2 --warmup 1 --no-cpu-baseline --no-sampler --no-viterbi --no-dropin 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
for k in ('cohort_h128',):
    if k in d: print(k, round(d[k]['value'] / 1e6, 2), 'M/s', round(d[k]['ms_per_step'], 2), {a: round(b, 2) for a, b in d[k]['kernel_ms'].items()}, 'frac', round(d[k]['roofline']['frac'], 3))
"
PG_LEANX=0 python tools/bench_leanx.py 20000 2>&1 | grep "H 128" | head -4
