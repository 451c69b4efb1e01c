#!/bin/bash
# Short GPU-box session: selected tests + a few bench lines.  usage: bash tools/gpu_quick.sh <tag> "<pytest -k expr>" [bench workloads...]
TAG=${1:-q}; KEXPR=${2:-lean}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -k "$KEXPR" --no-header -rf > $O/pytest.log 2>&1
echo "[pytest] exit $? ($(( $(date +%s) - T0 )) s)"; tail -4 $O/pytest.log | cut -c1-250
for W in "$@"; do
  timeout 600 python bench.py --workload $W --no-cohort --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_$W.log 2>&1
  python - <<PY
import json
for l in open("$O/bench_$W.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("$W", "value %.4g" % d["value"], "ms %.2f" % d["ms_per_step"], {k: round(v, 2) for k, v in d["kernel_ms"].items()})
PY
done
echo "total $(( $(date +%s) - T0 )) s"
