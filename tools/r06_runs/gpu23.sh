set -x
run() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cohort --no-sampler --no-viterbi --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'e2e', d['end_to_end'], 'dropin', d['dropin_threads']['ms_per_round'])
" >> gpurun_out/r06_upload23.txt; }
rm -f gpurun_out/r06_upload23.txt
run t4
PG_UPLOAD_THREADS=8 run t8
PG_UPLOAD_THREADS=12 run t12
PG_UPLOAD_THREADS=2 run t2
cat gpurun_out/r06_upload23.txt
