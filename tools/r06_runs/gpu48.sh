set -x
timeout 900 python -m pytest tests/test_viterbi_gpu.py -x -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/r06_tests48.txt
cat gpurun_out/r06_tests48.txt
timeout 1500 python tools/exp_viterbi_timeline.py > gpurun_out/r06_viterbi_timeline2.txt 2>&1
cat gpurun_out/r06_viterbi_timeline2.txt
