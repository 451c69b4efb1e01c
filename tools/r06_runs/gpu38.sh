set -x
rm -f gpurun_out/r06_soak.txt
timeout 900 python tools/soak_parity.py 900 606 2>&1 | tail -1 >> gpurun_out/r06_soak.txt
SOAK_X=1 timeout 900 python tools/soak_parity.py 2500 6106 2>&1 | tail -1 >> gpurun_out/r06_soak.txt
SOAK_TRI=1 timeout 900 python tools/soak_parity.py 600 6206 2>&1 | tail -1 >> gpurun_out/r06_soak.txt
PG_KERNELS=persist PG_SWEEP_MODE=chunked PG_CHUNK_COLS=64 SOAK_TRI=1 timeout 900 python tools/soak_parity.py 300 6306 2>&1 | tail -1 >> gpurun_out/r06_soak.txt
cat gpurun_out/r06_soak.txt
