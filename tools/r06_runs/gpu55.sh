set -x
timeout 900 python -m pytest tests/test_widef_gpu.py tests/test_split_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r06_tests55.txt
cat gpurun_out/r06_tests55.txt
grep -q "failed\|error" gpurun_out/r06_tests55.txt && exit 0
timeout 600 python bench.py --steps 3 --warmup 1 --cohort-only --cohort-key cohort_h64w --no-cpu-baseline > gpurun_out/r06_h64w_55.json 2> gpurun_out/r06_h64w_55.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_h64w_55.json').read().strip().splitlines()[-1]); r=d['cohort_h64w']
print(r['value'], r['ms_per_step'], r['sweep_mode']); print(r['plan'])
print({a:round(b,2) for a,b in r['kernel_ms'].items()})
PY
