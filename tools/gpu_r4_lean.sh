#!/bin/bash
# The lone-chain evidence of round 4 in one GPU-box session: microbenchmarks, the stamped timeline of the plain lean step,
# plain vs pipelined step with ablations, SQ counters of both.  output: gpurun_out/r04lean/
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04lean; mkdir -p $O; cd $R
timeout 120 tools/mb_issue.bin > $O/mb_issue.txt 2>&1
timeout 120 tools/mb_waves.bin > $O/mb_waves.txt 2>&1
timeout 300 python tools/exp_timeline.py > $O/timeline.txt 2>&1
timeout 600 python tools/exp_pipe.py check time variants prof prof@pipe dppsum PG_LEANP_EXP1@pipe PG_LEANP_EXP2@pipe PG_LEANP_EXP3@pipe > $O/pipe.txt 2>&1
SQ=1 bash tools/profile_workload.sh chr22_h64 r04 > $O/prof_plain.log 2>&1
PG_KERNELS=leanpipe SQ=1 bash tools/profile_workload.sh chr22_h64 r04pipe > $O/prof_pipe.log 2>&1
tail -12 $O/pipe.txt
