set -x
run() { timeout 600 python bench.py --steps 5 --warmup 2 --no-cohort --no-sampler --no-viterbi --no-dropin --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms', round(d['ms_per_step'],2), {a:round(b,2) for a,b in d['kernel_ms'].items() if 'sweep' in a}, d.get('device_bytes'))
" >> gpurun_out/r06_chunk35.txt; }
rm -f gpurun_out/r06_chunk35.txt
for rep in 1 2; do
run k4096_$rep
PG_CHUNK_COLS=8192 run k8192_$rep
PG_CHUNK_COLS=12288 run k12288_$rep
PG_CHUNK_COLS=6144 run k6144_$rep
done
cat gpurun_out/r06_chunk35.txt
