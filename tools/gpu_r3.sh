#!/bin/bash
# One GPU-box session (through gpurun):   bash tools/gpu_r3.sh <tag> <steps...>      output: gpurun_out/<tag>/
# steps:
#   tests     the whole -m gpu suite                 smoke     __graft_entry__.smoke()
#   threads   the threaded drop-in tests              bench     python bench.py (default line) -> bench_default.json
#   benchall  the other BASELINE workloads            extra     configs[4] one-GPU share + the C-ABI gather forced on one GPU
#   profiles  rocprofv3 captures + summaries (tools/gpu_profiles.sh)
#   lean      parity of the lean kernels + per-column times of the library and of every variant under tools/_build
#   leanx     parity of k_sweep_leanx + tools/bench_leanx.py + tools/exp_leanx.py
#   cohorts   quick parity of the per-variant kernels + the three cohort measurements
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
  tail -6 $O/$name.log | cut -c1-400 | tee -a $O/summary.txt
}
cohort_lines() { python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
for k in ('cohort', 'cohort_h16', 'cohort_h128'):
    if k in d: print(k, round(d[k]['value'] / 1e6, 2), 'M/s', round(d[k]['ms_per_step'], 2), {a: round(b, 2) for a, b in d[k]['kernel_ms'].items()}, 'frac', round(d[k]['roofline']['frac'], 3))
"; }
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
    extra)
      run bench_hprc_h128 900 python bench.py --workload hprc_h128 --no-cohort --no-sampler --no-viterbi --steps 2 --warmup 1
      PG_BENCH_FORCE_GATHER=1 PG_GATHER_LOOPBACK=1 run bench_forced_gather 600 python bench.py --workload genome24_small --no-cohort --no-sampler --no-viterbi --no-dropin --no-cpu-baseline --steps 2 --warmup 1
      grep -o '"gather": "[^"]*"' $O/bench_forced_gather.log | tee -a $O/summary.txt ;;
    profiles) run profiles 2400 bash tools/gpu_profiles.sh r03 ;;
    lean)
      run lean_tests 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header -x -k "lean or triangle or multi_contig or chunk_boundaries or unregularized or cohort"
      run lean_times 600 python tools/exp_lean.py run default $(ls tools/_build/ 2>/dev/null | grep PG_LEAN_EXP | sed 's/libpangenie_hmm_PG_LEAN_EXP/PG_LEAN_EXP=/;s/.so//') ;;
    leanx)
      run leanx_tests 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header -x -k "leanx or config4_shape or panels_vs_oracle or multi_contig_job or chunk_boundaries or generic_kernel_cross or wide_columns or many_alleles"
      run leanx_bench 600 python tools/bench_leanx.py 20000
      run leanx_times 600 python tools/exp_leanx.py run default $(ls tools/_build/ 2>/dev/null | grep PG_LX_EXP | sed 's/libpangenie_hmm_PG_LX_EXP/PG_LX_EXP=/;s/.so//') ;;
    cohorts)
      run cohort_tests 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header -x -k "class_sums or panels or known_answers or many_alleles or wide or cohort or small16 or no_columns or fixture_shape or transition or multi_contig"
      run cohort_bench 900 python bench.py --workload genome24_small --steps 2 --warmup 1 --no-cpu-baseline --no-sampler --no-viterbi --no-dropin
      cohort_lines < $O/cohort_bench.log | tee -a $O/summary.txt ;;
    *) echo "unknown step $step" | tee -a $O/summary.txt ;;
  esac
done
echo "total $(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
