set -x
timeout 900 python tools/plan_table.py > gpurun_out/r06_plan_table.txt 2> gpurun_out/r06_plan_table.err
tail -3 gpurun_out/r06_plan_table.err
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r06_gpu_tests21.txt
cat gpurun_out/r06_gpu_tests21.txt
timeout 1500 python bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
tail -2 gpurun_out/r06_bench_final.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_smoke.txt 2>&1; tail -2 gpurun_out/r06_smoke.txt
bash tools/profile.sh genome24_h64 r06b > gpurun_out/r06b_profile.log 2>&1
tail -30 gpurun_out/r06b_profile.log
