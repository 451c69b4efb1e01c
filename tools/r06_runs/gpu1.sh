set -x
timeout 900 python -m pytest tests/test_split_gpu.py tests/test_small16x_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/r06_split_tests.txt
cat gpurun_out/r06_split_tests.txt
for k in cohort_h16 cohort_h16m cohort_h16w; do
  timeout 600 python bench.py --steps 5 --warmup 2 --cohort-only --cohort-key $k --no-cpu-baseline > gpurun_out/r06_a_$k.json 2> gpurun_out/r06_a_$k.err
  tail -2 gpurun_out/r06_a_$k.err
done
