#!/bin/bash
# rocprofv3 captures of the three bench measurements + summaries into profiles/ (GPU box).  usage: bash tools/gpu_profiles.sh <tag>
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/profiles
for W in genome24_h64 cohort_h64 cohort_h16 cohort_h128; do
  bash tools/profile_workload.sh $W $TAG > gpurun_out/${TAG}_$W.log 2>&1
  python tools/summarize_profile.py gpurun_out/${TAG}_$W gpurun_out/profiles/${TAG}_$W $W > /dev/null 2>&1
  cp gpurun_out/${TAG}_$W/kt/kt_kernel_stats.csv gpurun_out/profiles/${TAG}_${W}_kernel_stats.csv 2>/dev/null
done
SQ=1 bash tools/profile_workload.sh chr22_h64 $TAG > gpurun_out/${TAG}_chr22_h64.log 2>&1
python tools/summarize_profile.py gpurun_out/${TAG}_chr22_h64 gpurun_out/profiles/${TAG}_chr22_h64 chr22_h64 > /dev/null 2>&1
cp gpurun_out/${TAG}_chr22_h64/kt/kt_kernel_stats.csv gpurun_out/profiles/${TAG}_chr22_h64_kernel_stats.csv 2>/dev/null
# the 128-path chain with multiallelic objects (k_sweep_leanx), SQ counters included
SQ=1 bash tools/profile_workload.sh chr22_h128 $TAG > gpurun_out/${TAG}_chr22_h128.log 2>&1
python tools/summarize_profile.py gpurun_out/${TAG}_chr22_h128 gpurun_out/profiles/${TAG}_chr22_h128 chr22_h128 > /dev/null 2>&1
cp gpurun_out/${TAG}_chr22_h128/kt/kt_kernel_stats.csv gpurun_out/profiles/${TAG}_chr22_h128_kernel_stats.csv 2>/dev/null
# the sampler: kernel trace + SQ instruction counters of the bench's sampler shape
mkdir -p gpurun_out/${TAG}_sampler
( cd /tmp && export TMPDIR=/tmp && cd $R
  SCMD="python tools/bench_sampler.py --variants 40000 --paths 215 --size 15 --contigs 8 --cpu-variants 2000"
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_sampler/kt -o kt --output-format csv -- $SCMD > gpurun_out/${TAG}_sampler/kt.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d gpurun_out/${TAG}_sampler/pmc_sq1 -o pmc --output-format csv -- $SCMD > gpurun_out/${TAG}_sampler/pmc_sq1.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d gpurun_out/${TAG}_sampler/pmc_sq2 -o pmc --output-format csv -- $SCMD > gpurun_out/${TAG}_sampler/pmc_sq2.log 2>&1
  rm -f gpurun_out/${TAG}_sampler/*/*kernel_trace.csv gpurun_out/${TAG}_sampler/*/*agent_info.csv )
cp gpurun_out/${TAG}_sampler/kt/kt_kernel_stats.csv gpurun_out/profiles/${TAG}_sampler_kernel_stats.csv 2>/dev/null
grep "^{" gpurun_out/${TAG}_sampler/kt.log | tail -1 > gpurun_out/profiles/${TAG}_sampler_bench.json 2>/dev/null
python tools/summarize_profile.py gpurun_out/${TAG}_sampler gpurun_out/profiles/${TAG}_sampler "sampler 8 contigs x 40000 x 215 paths x 15 passes" > /dev/null 2>&1
ls -la gpurun_out/profiles
