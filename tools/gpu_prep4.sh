#!/bin/bash
mkdir -p gpurun_out
{
echo "== prep tests"; timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "prep_four or known_answers or combine or triangle or cohort" 2>&1 | tail -6
echo "== soak"; timeout 300 python tools/soak_parity.py 120 77 2>&1 | grep -v amdgpu.ids | tail -2
echo "== cohort"; timeout 900 python bench.py --cohort-only --no-sampler --no-viterbi --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['cohort']['value'], d['cohort']['ms_per_step'], d['cohort']['kernel_ms'])"
echo "== genome24"; timeout 900 python bench.py --no-cohort --no-sampler --no-viterbi --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_ms'))"
} > gpurun_out/prep4.log 2>&1
tail -30 gpurun_out/prep4.log
