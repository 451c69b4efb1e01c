#!/bin/bash
mkdir -p gpurun_out
{
echo "== triangle tests"; timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "triangle or cohort or fused or lean" 2>&1 | tail -6
echo "== cohort"; timeout 600 python bench.py --cohort-only --no-sampler --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['cohort']['value'], d['cohort']['ms_per_step'], d['cohort']['kernel_ms'])"
} > gpurun_out/tri.log 2>&1
tail -20 gpurun_out/tri.log
