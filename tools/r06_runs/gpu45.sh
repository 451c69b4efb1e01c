set -x
timeout 900 python tools/exp_leanx2_multi.py > gpurun_out/r06_leanx2_multi.txt 2>&1
cat gpurun_out/r06_leanx2_multi.txt
