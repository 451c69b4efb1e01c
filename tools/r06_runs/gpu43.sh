set -x
timeout 1500 python -m pytest tests/test_split_gpu.py tests/test_parity_gpu.py -x -q -k "triangle or multi_contig or deep_bins" 2>&1 | tail -12 > gpurun_out/r06_tests43.txt
cat gpurun_out/r06_tests43.txt
run() { timeout 600 python bench.py --steps 5 --warmup 2 --cohort-only --cohort-key cohort_h64m --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['cohort_h64m']
print('$1', '%.1f M/s'%(r['value']/1e6), 'ms/step %.2f'%r['ms_per_step'], {a:round(b,2) for a,b in r['kernel_ms'].items()})
" >> gpurun_out/r06_h64m_43.txt; }
rm -f gpurun_out/r06_h64m_43.txt
run leanx2
PG_KERNELS=noleanx2 run ring
run leanx2_again
cat gpurun_out/r06_h64m_43.txt
