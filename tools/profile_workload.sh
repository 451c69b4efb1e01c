#!/bin/bash
# rocprofv3 capture for one bench workload (run on the GPU box through gpurun):
#   pass 1: --kernel-trace --stats           -> per-kernel average duration
#   pass 2: --pmc FETCH_SIZE  (own pass)     -> HBM read  KiB per launch
#   pass 3: --pmc WRITE_SIZE  (own pass)     -> HBM write KiB per launch
# usage: tools/profile_workload.sh <workload> <round-tag>      output: gpurun_out/<tag>_<workload>/
set -u
W=${1:-genome24_h64}; TAG=${2:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
CMD="python bench.py --steps 3 --warmup 1 --workload $W --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- $CMD > $OUT/kt.log 2>&1
tail -1 $OUT/kt.log | cut -c1-400
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc --output-format csv -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc --output-format csv -- $CMD > $OUT/pmc_write.log 2>&1
$CMD > $OUT/bench.json 2> $OUT/bench.err
# keep the merge-back small: kernel trace rows are not needed, only stats + counters
rm -f $OUT/*/*kernel_trace.csv $OUT/*/*agent_info.csv
ls -la $OUT $OUT/kt | head -30
