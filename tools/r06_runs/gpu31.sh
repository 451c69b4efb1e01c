set -x
run() { timeout 600 python bench.py --steps 5 --warmup 2 --no-cohort --no-sampler --no-viterbi --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'resident ms', round(d['ms_per_step'],2), {a:round(b,2) for a,b in d['kernel_ms'].items()}, 'e2e', round(d['end_to_end']['ms'],1), round(d['end_to_end']['run_ms'],1), 'dropin', [round(x,1) for x in d['dropin_threads']['round_ms']])
" >> gpurun_out/r06_var31.txt; }
rm -f gpurun_out/r06_var31.txt
run a
run b
run c
cat gpurun_out/r06_var31.txt
