set -x
run() { timeout 600 python bench.py --steps 5 --warmup 2 --no-cohort --no-sampler --no-viterbi --no-dropin --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms', round(d['ms_per_step'],2), {a:round(b,2) for a,b in d['kernel_ms'].items() if 'sweep' in a})
" >> gpurun_out/r06_prio34.txt; }
rm -f gpurun_out/r06_prio34.txt
for i in 1 2 3 4 5; do run prio_$i; done
for i in 1 2 3 4 5; do PG_STREAM_PRIO=0 run noprio_$i; done
cat gpurun_out/r06_prio34.txt
