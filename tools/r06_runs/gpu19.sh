set -x
timeout 1500 python -m pytest tests/test_small16x_gpu.py tests/test_split_gpu.py -x -q 2>&1 | tail -6 > gpurun_out/r06_tests19.txt
cat gpurun_out/r06_tests19.txt
for key in cohort_h16m cohort_h16w panels_h16; do
timeout 600 python bench.py --steps 5 --warmup 2 --cohort-only --cohort-key $key --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['$key']
print('$key', '%.1f M/s'%(r['value']/1e6), 'ms/step %.2f'%r['ms_per_step'], {a:round(b,2) for a,b in r['kernel_ms'].items()})
" >> gpurun_out/r06_x19.txt
done
cat gpurun_out/r06_x19.txt
