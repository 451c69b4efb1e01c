#!/bin/bash
mkdir -p gpurun_out
{
echo "== triangle tests"; timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "triangle or cohort or fused or lean" 2>&1 | tail -4
echo "== cohort tri (ring 4)"; timeout 600 python bench.py --cohort-only --no-sampler --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['cohort']['value'], d['cohort']['ms_per_step'], d['cohort']['kernel_ms'])"
if [ -f tools/_build/libpangenie_hmm_ring2.so ]; then
echo "== cohort tri (ring 2: two workgroups per CU in phase 2)"; PANGENIE_HMM_LIB=$PWD/tools/_build/libpangenie_hmm_ring2.so timeout 600 python bench.py --cohort-only --no-sampler --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['cohort']['value'], d['cohort']['ms_per_step'], d['cohort']['kernel_ms'])"
echo "== ring 2 parity"; PANGENIE_HMM_LIB=$PWD/tools/_build/libpangenie_hmm_ring2.so timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "triangle or cohort or fused" 2>&1 | tail -4
fi
} > gpurun_out/tri.log 2>&1
tail -20 gpurun_out/tri.log
