set -x
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r06_gpu_tests.txt
cat gpurun_out/r06_gpu_tests.txt
timeout 1500 python bench.py --steps 10 --warmup 3 > gpurun_out/r06_bench_full.json 2> gpurun_out/r06_bench_full.err
tail -3 gpurun_out/r06_bench_full.err
