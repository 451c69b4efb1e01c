#!/bin/bash
# A/B of one cohort workload on the GPU box (through gpurun):
#   bash tools/cohort_ab.sh <cohort_h16|cohort_h17|cohort_h128|cohort> trace            kernel trace (rocprofv3 --stats) of the default kernels
#   bash tools/cohort_ab.sh <key> kernels - fullcols nosmall2 ...                        bench line per PG_KERNELS value ("-" = unset)
#   bash tools/cohort_ab.sh <key> libs <name> ...                                        bench line per variant library tools/_build/libpangenie_hmm_<name>.so
K=$1; M=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/ab_$K; mkdir -p $O
CK="--cohort-only --no-cpu-baseline --no-sampler"; [ "$K" != cohort ] && CK="$CK --cohort-key $K"
line() { grep '^{' $1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2), 'M variants/s', round(d['ms_per_step'],2), 'ms, phase 2', round(r['phase2_ms'],2), 'ms')"; }
case $M in
  trace)
    cd /tmp && export TMPDIR=/tmp; cd $R
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python bench.py --steps 3 --warmup 1 $CK > $O/kt.log 2>&1
    rm -f $O/kt/*kernel_trace.csv $O/kt/*agent_info.csv
    python - <<P
import csv
for r in list(csv.DictReader(open("$O/kt/kt_kernel_stats.csv")))[:10]:
    print(r["Name"][:64].ljust(64), r["Calls"], round(float(r["AverageNs"]) / 1e6, 3), "ms")
P
    ;;
  kernels)
    for v in "$@"; do
      if [ "$v" = "-" ]; then unset PG_KERNELS; else export PG_KERNELS=$v; fi
      python bench.py --steps 5 --warmup 2 $CK > $O/k_$v.log 2>&1; echo "PG_KERNELS=$v:" $(line $O/k_$v.log)
    done ;;
  libs)
    for v in default "$@"; do
      if [ $v = default ]; then unset PANGENIE_HMM_LIB; else export PANGENIE_HMM_LIB=$R/tools/_build/libpangenie_hmm_$v.so; fi
      python bench.py --steps 5 --warmup 2 $CK > $O/l_$v.log 2>&1; echo "$v:" $(line $O/l_$v.log)
    done ;;
esac
