#!/bin/bash
# k_prep iteration: all parity tests (one process), then the two benches whose k_prep share matters.
mkdir -p gpurun_out
{
echo "== parity"; timeout 1500 python -m pytest tests -q -m gpu -x -k "not config3 and not config4 and not full_size" 2>&1 | tail -6
echo "== cohort"; timeout 600 python bench.py --cohort-only --no-sampler --no-cpu-baseline --steps 3 --warmup 1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['cohort']['value'], d['cohort']['ms_per_step'], d['cohort']['kernel_ms'])"
echo "== genome24"; timeout 600 python bench.py --no-cohort --no-sampler --no-cpu-baseline --steps 3 --warmup 1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms'])"
} > gpurun_out/prep.log 2>&1
tail -20 gpurun_out/prep.log
