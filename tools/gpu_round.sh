#!/bin/bash
# One GPU-box session: parity tests in separate processes (a hung kernel must not take the rest down),
# smoke, bench.  usage (through gpurun): bash tools/gpu_round.sh <tag> [quick]
TAG=${1:-r02a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
T0=$(date +%s)
run() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  local t=$(date +%s)
  timeout $to "$@" > $O/$name.log 2>&1
  echo "[$name] exit $? after $(( $(date +%s) - t )) s (total $(( $(date +%s) - T0 )) s)" | tee -a $O/summary.txt
  tail -3 $O/$name.log | cut -c1-300 | tee -a $O/summary.txt
}
run smoke 600 python -c "import __graft_entry__ as g; g.smoke()"
run pytest_core 1200 python -m pytest tests -m gpu -q -k "not generic and not many_paths and not fixture_shape and not config3 and not config4 and not full_size and not cohort" --no-header -rf
run pytest_generic 900 python -m pytest tests -m gpu -q -k "generic or many_paths or fixture_shape" --no-header -rf
run pytest_cohort 600 python -m pytest tests -m gpu -q -k "cohort" --no-header -rf
if [ "${2:-}" != "quick" ]; then
  run pytest_big 1500 python -m pytest tests -m gpu -q -k "config3 or config4 or full_size" --no-header -rf
  run bench_default 1500 python bench.py
  cp $O/bench_default.log $O/bench_default.json
  run bench_chr22 600 python bench.py --workload chr22_h64 --no-cohort --steps 3 --warmup 1
  run bench_h16 600 python bench.py --workload contig_h16 --no-cohort --steps 3 --warmup 1
  run bench_h128 600 python bench.py --workload chr22_h128 --no-cohort --steps 3 --warmup 1
fi
if [ "${2:-}" != "quick" ]; then
  # kernel trace of the Viterbi timing script (profiles/<tag>_viterbi_kernel_stats.csv)
  ( cd /tmp && export TMPDIR=/tmp && cd $R && mkdir -p gpurun_out/profiles &&
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/viterbi_kt -o kt --output-format csv -- python tools/bench_viterbi.py > $O/viterbi_kt.log 2>&1
    rm -f $O/viterbi_kt/*kernel_trace.csv $O/viterbi_kt/*agent_info.csv
    cp $O/viterbi_kt/kt_kernel_stats.csv gpurun_out/profiles/${TAG}_viterbi_kernel_stats.csv 2>/dev/null
    grep "^{" $O/viterbi_kt.log > gpurun_out/profiles/${TAG}_viterbi_bench.jsonl 2>/dev/null )
fi
echo "total $(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
