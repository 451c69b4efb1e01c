set -x
timeout 1200 python -m pytest tests/test_dropin_threads_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/r06_tests29.txt
cat gpurun_out/r06_tests29.txt
run() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cohort --no-sampler --no-viterbi --no-cpu-baseline 2>gpurun_out/r06_pipe29_$1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'resident ms', round(d['ms_per_step'],2), 'e2e', d['end_to_end'], 'dropin', d['dropin_threads']['round_ms'], d['dropin_threads']['value']/1e6)
" >> gpurun_out/r06_pipe29.txt; }
rm -f gpurun_out/r06_pipe29.txt
run pipelined
PG_NO_PIPELINE=1 run sequential
cat gpurun_out/r06_pipe29.txt
