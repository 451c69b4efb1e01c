#!/bin/bash
# rocprofv3 captures of the three bench measurements + summaries into profiles/ (GPU box).  usage: bash tools/gpu_profiles.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/profiles
for W in genome24_h64 cohort_h64; do
  bash tools/profile_workload.sh $W $TAG > gpurun_out/${TAG}_$W.log 2>&1
  python tools/summarize_profile.py gpurun_out/${TAG}_$W gpurun_out/profiles/${TAG}_$W $W > /dev/null 2>&1
  cp gpurun_out/${TAG}_$W/kt/kt_kernel_stats.csv gpurun_out/profiles/${TAG}_${W}_kernel_stats.csv 2>/dev/null
done
SQ=1 bash tools/profile_workload.sh chr22_h64 $TAG > gpurun_out/${TAG}_chr22_h64.log 2>&1
python tools/summarize_profile.py gpurun_out/${TAG}_chr22_h64 gpurun_out/profiles/${TAG}_chr22_h64 chr22_h64 > /dev/null 2>&1
cp gpurun_out/${TAG}_chr22_h64/kt/kt_kernel_stats.csv gpurun_out/profiles/${TAG}_chr22_h64_kernel_stats.csv 2>/dev/null
ls -la gpurun_out/profiles
