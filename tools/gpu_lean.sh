#!/bin/bash
# lean-sweep iteration on the GPU box: parity tests that exercise the lean kernels, then the lone-chain benches.
mkdir -p gpurun_out
{
echo "== parity (lean shapes)"; timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "lean or regime or zero or chunk or deep or resume or full_size or fused or cohort or determin or triangle or pipelined" 2>&1 | tail -5
for W in chr22_h64 genome24_h64; do
echo "== $W"; timeout 600 python bench.py --workload $W --no-cohort --no-sampler --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms'])"
done
} > gpurun_out/lean.log 2>&1
tail -30 gpurun_out/lean.log
