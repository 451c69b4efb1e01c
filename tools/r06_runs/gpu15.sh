set -x
timeout 900 python -m pytest tests/test_persist_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r06_tests15.txt
cat gpurun_out/r06_tests15.txt
run() { timeout 600 python bench.py --steps 5 --warmup 2 --no-cohort --no-sampler --no-viterbi --no-dropin --no-cpu-baseline 2>gpurun_out/r06_p15_$1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', '%.2f M/s'%(d['value']/1e6), 'ms/step %.2f'%d['ms_per_step'], {a:round(b,2) for a,b in d['kernel_ms'].items()})
" >> gpurun_out/r06_persist15.txt; }
rm -f gpurun_out/r06_persist15.txt
run default_4096
PG_CHUNK_COLS=2048 run k2048
PG_CHUNK_COLS=3072 run k3072
PG_CHUNK_COLS=6144 run k6144
PG_CHUNK_COLS=1024 run k1024
cat gpurun_out/r06_persist15.txt
