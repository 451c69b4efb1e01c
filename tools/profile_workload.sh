#!/bin/bash
# rocprofv3 capture for one bench workload (run on the GPU box through gpurun):
#   pass 1: --kernel-trace --stats           -> per-kernel average duration
#   pass 2: --pmc FETCH_SIZE  (own pass)     -> HBM read  KiB per launch
#   pass 3: --pmc WRITE_SIZE  (own pass)     -> HBM write KiB per launch
#   pass 4 (optional, SQ=1): four --pmc passes of SQ counters (issue / wait picture of the sweep)
# usage: tools/profile_workload.sh <workload|cohort_h64|cohort_h16|cohort_h128> <round-tag>      output: gpurun_out/<tag>_<workload>/
set -u
W=${1:-genome24_h64}; TAG=${2:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
if [ "$W" = "cohort_h64" ]; then
  CMD="python bench.py --steps 3 --warmup 1 --cohort-only --no-cpu-baseline --no-sampler"
elif [ "$W" = "cohort_h16" ] || [ "$W" = "cohort_h128" ] || [ "$W" = "cohort_h17" ]; then
  CMD="python bench.py --steps 3 --warmup 1 --cohort-only --cohort-key $W --no-cpu-baseline --no-sampler"
else
  CMD="python bench.py --steps 3 --warmup 1 --workload $W --no-cpu-baseline --no-cohort --no-sampler --no-viterbi --no-dropin"
fi
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- $CMD > $OUT/kt.log 2>&1
tail -1 $OUT/kt.log | cut -c1-400
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc --output-format csv -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc --output-format csv -- $CMD > $OUT/pmc_write.log 2>&1
if [ "${SQ:-0}" = "1" ]; then
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc_sq$i -o pmc --output-format csv -- $CMD > $OUT/pmc_sq$i.log 2>&1
  done
fi
# keep the merge-back small: kernel trace rows are not needed, only stats + counters
rm -f $OUT/*/*kernel_trace.csv $OUT/*/*agent_info.csv
ls $OUT $OUT/kt | head -30
