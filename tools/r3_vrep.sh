#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header -x -k "class_sums or panels or known_answers or many_alleles or wide or cohort or small16 or no_columns or fixture_shape or transition or multi_contig or lean_kernel" 2>&1 | tail -3
python bench.py --workload genome24_small --steps 2 --warmup 1 --no-cpu-baseline --no-sampler --no-viterbi --no-dropin 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('main', round(d['value'] / 1e6, 2), {a: round(b, 2) for a, b in d['kernel_ms'].items()})
for k in ('cohort', 'cohort_h16', 'cohort_h128'):
    if k in d: print(k, round(d[k]['value'] / 1e6, 2), 'M/s', round(d[k]['ms_per_step'], 2), {a: round(b, 2) for a, b in d[k]['kernel_ms'].items()}, 'frac', round(d[k]['roofline']['frac'], 3))
"
