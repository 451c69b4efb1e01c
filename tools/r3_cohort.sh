#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header -x -k "triangle or cohort" 2>&1 | tail -3
timeout 600 python bench.py --cohort-only --no-cpu-baseline --no-sampler --no-viterbi --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['cohort']; print('cohort', round(c['value'] / 1e6, 1), 'M/s', round(c['ms_per_step'], 2), 'ms', {k: round(v, 2) for k, v in c['kernel_ms'].items()}, c['roofline']['frac'])"
