#!/bin/bash
# extra bench lines of round 3: configs[4] one-GPU share; the C-ABI gather forced on one GPU (loop-back) inside bench.py
cd ${GRAFT_REPO_ROOT:-.}
O=${1:-gpurun_out}
python bench.py --workload hprc_h128 --no-cohort --no-sampler --no-viterbi --steps 2 --warmup 1 > $O/bench_hprc_h128.json 2> $O/bench_hprc_h128.err; tail -c 1500 $O/bench_hprc_h128.json
PG_BENCH_FORCE_GATHER=1 PG_GATHER_LOOPBACK=1 python bench.py --workload genome24_small --no-cohort --no-sampler --no-viterbi --no-dropin --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_forced_gather.json 2> $O/bench_forced_gather.err; grep -o '"gather": "[^"]*"' $O/bench_forced_gather.json; tail -3 $O/bench_forced_gather.err
