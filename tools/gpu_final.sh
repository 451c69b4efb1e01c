mkdir -p gpurun_out/r02f
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -5
timeout 900 python bench.py > gpurun_out/r02f/bench_default.log 2>&1; tail -c 600 gpurun_out/r02f/bench_default.log
bash tools/profile_workload.sh cohort_h64 r02 > gpurun_out/r02_cohort_h64.log 2>&1
python tools/summarize_profile.py gpurun_out/r02_cohort_h64 gpurun_out/profiles/r02_cohort_h64 cohort_h64 > /dev/null 2>&1
cp gpurun_out/r02_cohort_h64/kt/kt_kernel_stats.csv gpurun_out/profiles/r02_cohort_h64_kernel_stats.csv
cat gpurun_out/profiles/r02_cohort_h64_summary.txt
