set -x
SOAK_LX2=1 timeout 900 python tools/soak_parity.py 250 9901 2>&1 | tail -3 > gpurun_out/r06_soak_lx2.txt
cat gpurun_out/r06_soak_lx2.txt
timeout 600 python bench.py --steps 3 --warmup 1 --cohort-only --cohort-key cohort_h64w --no-cpu-baseline > gpurun_out/r06_h64w.json 2> gpurun_out/r06_h64w.err
cut -c1-1500 gpurun_out/r06_h64w.json; tail -3 gpurun_out/r06_h64w.err
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/r06_pytest_final5.txt
cat gpurun_out/r06_pytest_final5.txt
timeout 1200 python bench.py > gpurun_out/r06_bench_final5.json 2> gpurun_out/r06_bench_final5.err
cut -c1-400 gpurun_out/r06_bench_final5.json; tail -3 gpurun_out/r06_bench_final5.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
