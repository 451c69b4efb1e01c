#!/bin/bash
# One GPU call for the sampler row: parity tests on both kernels, then timings.
# usage (from the repo root, here): /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_sampler.sh'
mkdir -p gpurun_out
{
echo "== tests (fast kernel where it applies)"; timeout 400 python -m pytest tests/test_sampler.py -q -m gpu -x 2>&1 | tail -15
echo "== tests (general kernel forced)"; PG_SAMPLER_KERNEL=general timeout 400 python -m pytest tests/test_sampler.py -q -m gpu -x 2>&1 | tail -15
echo "== bench"
timeout 300 python tools/bench_sampler.py --variants 200000 --paths 215 --size 15 --check
timeout 300 python tools/bench_sampler.py --variants 200000 --paths 64 --size 15
timeout 300 python tools/bench_sampler.py --variants 100000 --paths 1000 --size 15
timeout 300 python tools/bench_sampler.py --variants 100000 --paths 215 --size 15 --contigs 24
PG_SAMPLER_KERNEL=general timeout 300 python tools/bench_sampler.py --variants 50000 --paths 215 --size 15
} > gpurun_out/sampler.log 2>&1
tail -60 gpurun_out/sampler.log
