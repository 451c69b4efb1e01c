set -x
timeout 900 python -m pytest tests/test_widef_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r06_tests52.txt
cat gpurun_out/r06_tests52.txt
grep -q "failed\|error" gpurun_out/r06_tests52.txt && exit 0
timeout 600 python bench.py --steps 3 --warmup 1 --cohort-only --cohort-key cohort_h64w --no-cpu-baseline > gpurun_out/r06_h64w_52.json 2> gpurun_out/r06_h64w_52.err
cut -c1-1200 gpurun_out/r06_h64w_52.json; tail -3 gpurun_out/r06_h64w_52.err
PG_KERNELS=nowidef timeout 600 python bench.py --steps 3 --warmup 1 --cohort-only --cohort-key cohort_h64w --no-cpu-baseline 2>/dev/null | cut -c1-600
