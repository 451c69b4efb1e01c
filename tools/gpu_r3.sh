#!/bin/bash
# One GPU-box session of round 3.  usage (through gpurun): bash tools/gpu_r3.sh <tag> <steps...>
# steps: threads | tests | bench | benchall | <any script under tools/ ending in .sh>
TAG=${1:-r03a}; shift
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
  tail -4 $O/$name.log | cut -c1-400 | tee -a $O/summary.txt
}
for step in "$@"; do
  case $step in
    threads) run threads 900 python -m pytest tests/test_dropin_threads_gpu.py tests/test_host_cpp.py -m gpu -q --no-header -rf -x ;;
    tests) run pytest_all 1500 python -m pytest tests -m gpu -q --no-header -rf ;;
    smoke) run smoke 600 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) run bench_default 1500 python bench.py; grep '^{' $O/bench_default.log > $O/bench_default.json ;;
    benchall)
      run bench_chr22 600 python bench.py --workload chr22_h64 --no-cohort --no-sampler --no-viterbi --steps 3 --warmup 1
      run bench_h16 600 python bench.py --workload contig_h16 --no-cohort --no-sampler --no-viterbi --steps 3 --warmup 1
      run bench_h128 600 python bench.py --workload chr22_h128 --no-cohort --no-sampler --no-viterbi --steps 3 --warmup 1 ;;
    profiles) run profiles 2400 bash tools/gpu_profiles.sh r03 ;;
    *.sh) run $(basename $step .sh) 1500 bash tools/$step $O ;;
    *) echo "unknown step $step" | tee -a $O/summary.txt ;;
  esac
done
echo "total $(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
