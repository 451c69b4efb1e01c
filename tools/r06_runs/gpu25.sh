set -x
run() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cohort --no-sampler --no-viterbi --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'resident ms', round(d['ms_per_step'],2), 'e2e', round(d['end_to_end']['ms'],2), 'dropin', d['dropin_threads']['round_ms'], d['dropin_threads']['value']/1e6)
" >> gpurun_out/r06_pipe25.txt; }
rm -f gpurun_out/r06_pipe25.txt
PG_STREAM_NB=1 run pipelined_nb
PG_STREAM_NB=1 PG_NO_PIPELINE=1 run sequential_nb
cat gpurun_out/r06_pipe25.txt
