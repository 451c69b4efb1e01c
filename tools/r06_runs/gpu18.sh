set -x
timeout 600 python tools/exp_small_crossover.py > gpurun_out/r06_small_crossover.txt 2>&1
timeout 600 python tools/exp_small_crossover.py --multi 0.2 >> gpurun_out/r06_small_crossover.txt 2>&1
cat gpurun_out/r06_small_crossover.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r06_gpu_tests18.txt
cat gpurun_out/r06_gpu_tests18.txt
