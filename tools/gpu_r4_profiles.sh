#!/bin/bash
# rocprofv3 captures of round 4 (kernel stats + FETCH_SIZE / WRITE_SIZE passes; SQ counters for the 17-path cohort) and the
# default bench line.  output: gpurun_out/r04_<workload>/, gpurun_out/r04_bench_default.json
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for w in genome24_h64 cohort_h64 cohort_h16 cohort_h128; do bash tools/profile_workload.sh $w r04 > gpurun_out/r04_$w.log 2>&1; done
SQ=1 bash tools/profile_workload.sh cohort_h17 r04 > gpurun_out/r04_cohort_h17.log 2>&1
python bench.py > gpurun_out/r04_bench.log 2>&1; grep '^{' gpurun_out/r04_bench.log | tail -1 > gpurun_out/r04_bench_default.json
for w in chr22_h64 contig_h16 chr22_h128 hprc_h128; do python bench.py --workload $w --no-cohort --no-sampler --no-viterbi --no-dropin 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r04_bench_$w.json; done
ls gpurun_out | grep r04_ | head -30
