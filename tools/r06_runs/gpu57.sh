set -x
timeout 900 python -m pytest tests/test_widef_gpu.py tests/test_split_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r06_tests57.txt
cat gpurun_out/r06_tests57.txt
grep -q "failed\|error" gpurun_out/r06_tests57.txt && exit 0
timeout 600 python bench.py --steps 3 --warmup 1 --cohort-only --cohort-key cohort_h64w --no-cpu-baseline > gpurun_out/r06_h64w_57.json 2> gpurun_out/r06_h64w_57.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_h64w_57.json').read().strip().splitlines()[-1]); r=d['cohort_h64w']
print(r['value'], r['ms_per_step'], r['sweep_mode']); print(r['plan'])
print({a:round(b,2) for a,b in r['kernel_ms'].items()})
PY
timeout 600 python bench.py --steps 3 --warmup 1 --cohort-only --cohort-key cohort_h64m --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['cohort_h64m']
print('h64m', r['value'], r['ms_per_step'], {a:round(b,2) for a,b in r['kernel_ms'].items()})"
