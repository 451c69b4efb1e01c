#!/bin/bash
# Variant libraries side by side on one cohort workload (GPU box, through gpurun): per-kernel-class ms of each.
#   bash tools/ab_libs.sh <cohort key> <name> ...      libraries tools/_build/libpangenie_hmm_<name>.so (built here on the CPU:
#   pangenie_amd.build.build_hip(out=..., defines=[...]) — e.g. the PG_X_EXP ablation masks of pg_experiments.h; "default" = the product library)
K=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/ab_$K; mkdir -p $O
for v in "$@"; do
  if [ "$v" = default ]; then unset PANGENIE_HMM_LIB; else export PANGENIE_HMM_LIB=$R/tools/_build/libpangenie_hmm_$v.so; fi
  python bench.py --steps 3 --warmup 1 --cohort-only --cohort-key $K --no-cpu-baseline --no-sampler > $O/l_$v.log 2>/dev/null
  grep '^{' $O/l_$v.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['$K']
print('%-8s %7.1f M  %6.2f ms  ' % ('$v', r['value'] / 1e6, r['ms_per_step']) + '  '.join('%s %.2f' % (k.replace('k_sweep_', ''), v) for k, v in r['kernel_ms'].items()))"
done
