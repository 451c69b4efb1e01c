#!/bin/bash
# THE profiling recipe (run on the GPU box through gpurun): rocprofv3 captures of ONE bench workload and their summary.
#   pass 1: --kernel-trace --stats           -> per-kernel average duration
#   pass 2: --pmc FETCH_SIZE  (own pass)     -> HBM read  KiB per launch (doubled: the gfx950 correction, MI355X_MICROARCH.md)
#   pass 3: --pmc WRITE_SIZE  (own pass)     -> HBM write KiB per launch
#   SQ=1  : four more --pmc passes of SQ counters (issue / wait picture of the sweeps)
#   UNITS=1: three more passes — TA busy / stalls, VMEM FIFO stalls (which unit bounds a sweep).  SLOW: the TA_* counters
#            serialise the dispatches — ten minutes per workload on the cohorts; budget for it
# usage: [SQ=1] [UNITS=1] tools/profile.sh <workload> <tag>
#   <workload> = a main workload of bench.py (genome24_h64, chr22_h64, ...), `cohort_h64`, any key of bench.py's COHORTS_MORE
#                (cohort_h16, cohort_h16m, cohort_h16w, cohort_h64m, cohort_h128, cohort_h17), `sampler` or `viterbi`
# output: gpurun_out/<tag>_<workload>/ (raw, trimmed) and gpurun_out/profiles/<tag>_<workload>_{summary.txt,summary.json,kernel_stats.csv}
#         — copy the latter into profiles/ (tracked).  Every summary under profiles/ of round 5 on was made by this script.
set -u
W=${1:?workload}; TAG=${2:?tag}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_$W
mkdir -p $OUT $R/gpurun_out/profiles
cd /tmp && export TMPDIR=/tmp
cd $R
# (the bench's end-to-end rounds run pg_job_upload_run, which splits phase 1 into two launches — the long chains' and the others' —:
#  the per-kernel AVERAGES of a profile would then mix whole and partial launches; profiled runs take the sequential path)
export PG_NO_PIPELINE=1
case $W in
  cohort_h64) CMD="python bench.py --steps 3 --warmup 1 --cohort-only --no-cpu-baseline --no-sampler" ;;
  cohort_*|panels_h16)   CMD="python bench.py --steps 3 --warmup 1 --cohort-only --cohort-key $W --no-cpu-baseline --no-sampler" ;;
  sampler)    CMD="python tools/bench_sampler.py --variants 40000 --paths 215 --size 15 --contigs 8 --cpu-variants 2000" ;;
  viterbi)    CMD="python tools/bench_viterbi.py" ;;
  *)          CMD="python bench.py --steps 3 --warmup 1 --workload $W --no-cpu-baseline --no-cohort --no-sampler --no-viterbi --no-dropin" ;;
esac
pass() { local name=$1; shift; timeout 900 rocprofv3 "$@" -d $OUT/$name -o ${name%%_*} --output-format csv -- $CMD > $OUT/$name.log 2>&1; }
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- $CMD > $OUT/kt.log 2>&1
grep '^{' $OUT/kt.log | tail -1 | cut -c1-300
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc --output-format csv -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc --output-format csv -- $CMD > $OUT/pmc_write.log 2>&1
i=0
if [ "${SQ:-0}" = "1" ]; then
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1)); timeout 600 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc_sq$i -o pmc --output-format csv -- $CMD > $OUT/pmc_sq$i.log 2>&1
  done
fi
if [ "${UNITS:-0}" = "1" ]; then
  for set in "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "GRBM_GUI_ACTIVE TA_FLAT_WAVEFRONTS_sum SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
    i=$((i+1)); timeout 600 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc_sq$i -o pmc --output-format csv -- $CMD > $OUT/pmc_sq$i.log 2>&1
  done
fi
# keep the merge-back small: kernel trace rows are not needed, only stats + counters
rm -f $OUT/*/*kernel_trace.csv $OUT/*/*agent_info.csv
python tools/summarize_profile.py $OUT $R/gpurun_out/profiles/${TAG}_$W $W > /dev/null 2>&1
# ... and the raw counter rows are in the summary now (a TA_* pass writes one row per instance and dispatch: hundreds of MB,
# and gpurun refuses to copy back more than 64 MiB — round 5 lost a 22-minute capture that way)
rm -rf $OUT/pmc_* $OUT/kt/*_domain_stats.csv
cp $OUT/kt/kt_kernel_stats.csv $R/gpurun_out/profiles/${TAG}_${W}_kernel_stats.csv 2>/dev/null
grep '^{' $OUT/kt.log | tail -1 > $R/gpurun_out/profiles/${TAG}_${W}_bench_under_rocprof.json 2>/dev/null
head -40 $R/gpurun_out/profiles/${TAG}_${W}_summary.txt
