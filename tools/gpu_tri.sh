#!/bin/bash
mkdir -p gpurun_out
{
echo "== triangle tests"; timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "triangle or cohort or fused or lean" 2>&1 | tail -4
echo "== soak tri"; SOAK_TRI=1 timeout 300 python tools/soak_parity.py 80 123 2>&1 | grep -v amdgpu.ids | tail -3
for S in 32 64; do
echo "== cohort $S samples"; timeout 900 python bench.py --cohort-only --cohort-samples $S --no-sampler --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['cohort']['value'], d['cohort']['ms_per_step'], d['cohort']['device_bytes']/1e9, d['cohort']['kernel_ms'])"
done
} > gpurun_out/tri.log 2>&1
tail -20 gpurun_out/tri.log
