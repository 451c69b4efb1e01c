#!/bin/bash
# One GPU-box session (through gpurun):   bash tools/gpu_r4.sh <tag> <steps...>      output: gpurun_out/<tag>/
# steps:
#   pipe      tools/exp_pipe.py: pipelined vs plain lean step (agreement, ns per column, variants under tools/_build)
#   timeline  tools/exp_timeline.py: per-segment cycles of the plain lean step (-DPG_LEAN_TIMELINE build)
#   tests     the whole -m gpu suite                 smoke     __graft_entry__.smoke()
#   leantests the lean-kernel parity tests only      bench     python bench.py (default line) -> bench_default.json
TAG=${1:-r04a}; shift
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
  tail -${TAILN:-8} $O/$name.log | cut -c1-600 | tee -a $O/summary.txt
}
for step in "$@"; do
  case $step in
    pipe)      TAILN=30 run pipe 600 python tools/exp_pipe.py check time variants ${PIPE_VARIANTS:-prof PG_LEANP_EXP=1 PG_LEANP_EXP=2 PG_LEANP_EXP=3} ;;
    pipetime)  TAILN=30 run pipetime 600 python tools/exp_pipe.py time variants ${PIPE_VARIANTS:-prof} ;;
    timeline)  TAILN=30 run timeline 300 python tools/exp_timeline.py ;;
    leantests) run leantests 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "lean_kernel or chunk_boundaries or full_size or panels_vs_oracle" --maxfail=5 ;;
    tests)     run tests 1500 python -m pytest tests -q -m gpu --maxfail=8 ;;
    smoke)     run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench)     run bench 900 python bench.py; grep '^{' $O/bench.log | tail -1 > $O/bench_default.json ;;
    *) echo "unknown step $step" ;;
  esac
done
