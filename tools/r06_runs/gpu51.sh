set -x
SOAK_LX2=1 timeout 900 python tools/soak_parity.py 5000 9911 2>&1 | tail -3 > gpurun_out/r06_soak_lx2_big.txt
cat gpurun_out/r06_soak_lx2_big.txt
bash tools/profile.sh viterbi r06c
bash tools/profile.sh cohort_h64w r06c
