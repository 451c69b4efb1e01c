#!/bin/bash
# lean-step variants on the GPU box: quick parity of the lean kernels, then per-column times of every built variant
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header -x -k "lean or triangle or multi_contig or chunk_boundaries or unregularized or cohort" 2>&1 | tail -3
python tools/exp_lean.py run default $(ls tools/_build/ | grep PG_LEAN_EXP | sed 's/libpangenie_hmm_PG_LEAN_EXP/PG_LEAN_EXP=/;s/.so//')
python bench.py --steps 3 --warmup 1 --cohort-only --no-cpu-baseline --no-sampler 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
for k in ('cohort', 'cohort_h16', 'cohort_h128'):
    if k in d: print(k, round(d[k]['value'] / 1e6, 2), 'M/s', d[k]['ms_per_step'], {a: round(b, 2) for a, b in d[k]['kernel_ms'].items()}, 'frac', round(d[k]['roofline']['frac'], 3))
"
