set -x
timeout 1200 python -m pytest tests/test_dropin_threads_gpu.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_tests24.txt
cat gpurun_out/r06_tests24.txt
run() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cohort --no-sampler --no-viterbi --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'resident ms', round(d['ms_per_step'],2), 'e2e', round(d['end_to_end']['ms'],2), 'dropin', d['dropin_threads']['round_ms'], d['dropin_threads']['value']/1e6)
" >> gpurun_out/r06_pipe24.txt; }
rm -f gpurun_out/r06_pipe24.txt
run pipelined
PG_NO_PIPELINE=1 run sequential
run pipelined_again
cat gpurun_out/r06_pipe24.txt
