set -x
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06_pytest_final4.txt
cat gpurun_out/r06_pytest_final4.txt
timeout 900 python bench.py > gpurun_out/r06_bench_final4.json 2> gpurun_out/r06_bench_final4.err
cut -c1-600 gpurun_out/r06_bench_final4.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python tools/plan_table.py > gpurun_out/r06_plan_table.txt 2>&1
tail -30 gpurun_out/r06_plan_table.txt
bash tools/profile.sh cohort_h64m r06c
