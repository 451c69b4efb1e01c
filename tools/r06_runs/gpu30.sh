set -x
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r06_gpu_tests30.txt
cat gpurun_out/r06_gpu_tests30.txt
timeout 1500 python bench.py > gpurun_out/r06_bench_final2.json 2> gpurun_out/r06_bench_final2.err
tail -2 gpurun_out/r06_bench_final2.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_smoke2.txt 2>&1; tail -2 gpurun_out/r06_smoke2.txt
