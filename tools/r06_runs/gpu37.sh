set -x
bash tools/profile.sh genome24_h64 r06c > gpurun_out/r06c_profile.log 2>&1
tail -25 gpurun_out/r06c_profile.log
