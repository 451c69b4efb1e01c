#!/bin/bash
# Viterbi phasing on the GPU: parity tests, the C++ mirror, a timing line
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_viterbi_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -15
timeout 300 ./tests/cpp/test_host.bin gpu tests/golden 2>&1 | grep -v "^ok" | tail -8
timeout 600 python tools/bench_viterbi.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/viterbi_bench.txt | tail -12
