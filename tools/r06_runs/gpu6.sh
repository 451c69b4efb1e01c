set -x
timeout 1500 python -m pytest tests/test_split_gpu.py tests/test_small16x_gpu.py tests/test_parity_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r06_tests6.txt
cat gpurun_out/r06_tests6.txt
timeout 600 python bench.py --steps 5 --warmup 2 --cohort-only --cohort-key panels_h16 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['panels_h16']
print('%.1f M/s'%(r['value']/1e6), 'ms/step %.2f'%r['ms_per_step'], {a:round(b,2) for a,b in r['kernel_ms'].items()}, 'index pass', r['index_pass_ms'], 'incl', r['value_incl_index_pass']/1e6)
print(r['plan'])
" > gpurun_out/r06_panels6.txt
cat gpurun_out/r06_panels6.txt
