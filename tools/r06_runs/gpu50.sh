set -x
SOAK_LX2=1 timeout 900 python tools/soak_parity.py 300 9901 2>&1 | tail -3 > gpurun_out/r06_soak_lx2.txt
cat gpurun_out/r06_soak_lx2.txt
timeout 1200 python tools/soak_parity.py 600 9902 2>&1 | tail -3 > gpurun_out/r06_soak_final.txt
cat gpurun_out/r06_soak_final.txt
SOAK_X=1 timeout 600 python tools/soak_parity.py 200 9903 2>&1 | tail -3 > gpurun_out/r06_soak_x_final.txt
cat gpurun_out/r06_soak_x_final.txt
SOAK_TRI=1 timeout 600 python tools/soak_parity.py 150 9904 2>&1 | tail -3 > gpurun_out/r06_soak_tri_final.txt
cat gpurun_out/r06_soak_tri_final.txt
