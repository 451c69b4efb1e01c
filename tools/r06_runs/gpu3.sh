set -x
timeout 900 python -m pytest tests/test_split_gpu.py -x -q -k triangle 2>&1 | tail -15 > gpurun_out/r06_tri_tests.txt
cat gpurun_out/r06_tri_tests.txt
timeout 600 python bench.py --steps 5 --warmup 2 --cohort-only --cohort-key cohort_h64m --no-cpu-baseline > gpurun_out/r06_b_cohort_h64m.json 2> gpurun_out/r06_b_cohort_h64m.err
tail -2 gpurun_out/r06_b_cohort_h64m.err
