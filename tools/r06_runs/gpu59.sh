set -x
SOAK_WIDEF=1 timeout 1200 python tools/soak_parity.py 1500 9951 2>&1 | tail -4 > gpurun_out/r06_soak_widef.txt
cat gpurun_out/r06_soak_widef.txt
