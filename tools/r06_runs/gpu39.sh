set -x
SOAK_PERSIST=1 timeout 900 python tools/soak_parity.py 400 6406 2>&1 | tail -2 > gpurun_out/r06_soak_persist.txt
cat gpurun_out/r06_soak_persist.txt
