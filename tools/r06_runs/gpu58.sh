set -x
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/r06_pytest_final8.txt
cat gpurun_out/r06_pytest_final8.txt
timeout 900 python tools/soak_parity.py 600 9941 2>&1 | tail -3 > gpurun_out/r06_soak_final8.txt
cat gpurun_out/r06_soak_final8.txt
timeout 1200 python bench.py > gpurun_out/r06_bench_final8.json 2> gpurun_out/r06_bench_final8.err
cut -c1-400 gpurun_out/r06_bench_final8.json; tail -3 gpurun_out/r06_bench_final8.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python tools/plan_table.py > gpurun_out/r06_plan_table.txt 2>&1
bash tools/profile.sh cohort_h64w r06c
