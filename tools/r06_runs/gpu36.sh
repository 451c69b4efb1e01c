set -x
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r06_gpu_tests36.txt
cat gpurun_out/r06_gpu_tests36.txt
timeout 1500 python bench.py > gpurun_out/r06_bench_final3.json 2> gpurun_out/r06_bench_final3.err
tail -2 gpurun_out/r06_bench_final3.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_smoke3.txt 2>&1; tail -2 gpurun_out/r06_smoke3.txt
bash tools/profile.sh genome24_h64 r06c > gpurun_out/r06c_profile.log 2>&1
tail -25 gpurun_out/r06c_profile.log
