set -x
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r06_gpu_tests.txt
cat gpurun_out/r06_gpu_tests.txt
