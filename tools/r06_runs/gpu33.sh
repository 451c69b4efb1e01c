set -x
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_small16x_gpu.py -x -q -k "chunk or lean or config or full_size or multi_contig or wide or generic or deep" 2>&1 | tail -5 > gpurun_out/r06_tests33.txt
cat gpurun_out/r06_tests33.txt
run() { timeout 600 python bench.py --steps 5 --warmup 2 --no-cohort --no-sampler --no-viterbi --no-dropin --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms', round(d['ms_per_step'],2), {a:round(b,2) for a,b in d['kernel_ms'].items() if 'sweep' in a})
" >> gpurun_out/r06_post33.txt; }
rm -f gpurun_out/r06_post33.txt
run shared
PG_POST_FIXED=1 run fixed
run shared_again
PG_CHUNK_COLS=2048 run shared_k2048
PG_CHUNK_COLS=8192 run shared_k8192
cat gpurun_out/r06_post33.txt
