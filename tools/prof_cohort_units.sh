#!/bin/bash
# Which unit bounds the two cohort sweeps: TA busy / stalls, VMEM FIFO stalls, SQ issue vs wait (rocprofv3 --pmc, one pass per set).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/cohort_units; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
CMD="python bench.py --steps 2 --warmup 1 --cohort-only --no-cpu-baseline --no-sampler"
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE TA_FLAT_WAVEFRONTS_sum SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o pmc --output-format csv -- $CMD > $OUT/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob("gpurun_out/cohort_units/p*/pmc_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:40]
        if "k_sweep_lean" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
for k in agg:
    print(k)
    for c in sorted(agg[k]): print("   %-36s %.4g per launch" % (c, agg[k][c]/cnt[k][c]))
PY
rm -f $OUT/*/*kernel_trace.csv $OUT/*/*agent_info.csv
