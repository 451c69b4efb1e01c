set -x
timeout 600 python -m pytest tests/test_viterbi_gpu.py -x -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/r06_tests47.txt
cat gpurun_out/r06_tests47.txt
timeout 600 python -m pytest tests/test_split_gpu.py -x -q -k "leanx2" 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 >> gpurun_out/r06_tests47.txt
cat gpurun_out/r06_tests47.txt
timeout 1500 python tools/exp_viterbi_timeline.py > gpurun_out/r06_viterbi_timeline.txt 2>&1
cat gpurun_out/r06_viterbi_timeline.txt
