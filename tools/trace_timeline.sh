#!/bin/bash
# Kernel timeline of one workload (GPU box): rocprofv3 --kernel-trace, the per-dispatch rows kept (start / end ns, stream):
# gaps between the chunk sweeps and k_post of a chunked job, overlap of the two streams.   usage: tools/trace_timeline.sh <workload> <tag>
W=${1:-genome24_h64}; TAG=${2:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${TAG}_trace_$W; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 600 rocprofv3 --kernel-trace -d $OUT/kt -o kt --output-format csv -- python bench.py --steps 1 --warmup 1 --workload $W --no-cpu-baseline --no-cohort --no-sampler --no-viterbi --no-dropin > $OUT/kt.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/kt/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open("$OUT/timeline.tsv", "w") as o:
    o.write("start_us\tend_us\tdur_us\tqueue\tkernel\n")
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        o.write("%.1f\t%.1f\t%.1f\t%s\t%s\n" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", ""), r["Kernel_Name"][:60]))
print(len(rows), "dispatches")
PY
rm -rf $OUT/kt
