set -x
timeout 900 python -m pytest tests/test_persist_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r06_tests14.txt
cat gpurun_out/r06_tests14.txt
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_dropin_threads_gpu.py -x -q -k "lean or chunk or config3 or full_size or multi_contig or threads" 2>&1 | tail -8 >> gpurun_out/r06_tests14.txt
tail -8 gpurun_out/r06_tests14.txt
run() { timeout 600 python bench.py --steps 5 --warmup 2 --no-cohort --no-sampler --no-viterbi --no-dropin --no-cpu-baseline 2>gpurun_out/r06_p14_$1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', '%.2f M/s'%(d['value']/1e6), 'ms/step %.2f'%d['ms_per_step'], {a:round(b,2) for a,b in d['kernel_ms'].items()})
" >> gpurun_out/r06_persist14.txt; }
rm -f gpurun_out/r06_persist14.txt
run persist_pb8
PG_POST_BLOCKS=6 run persist_pb6

PG_CHUNK_COLS=1024 run persist_pb8_k1024

cat gpurun_out/r06_persist14.txt
