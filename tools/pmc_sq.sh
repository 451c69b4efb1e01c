cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/pmc/sq_counters.txt
cd $R
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmc/$n -o out --output-format csv -- python bench.py --steps 1 --warmup 0 --workload chr22_h64 --no-cpu-baseline > gpurun_out/pmc/$n.log 2>&1
done
find gpurun_out/pmc -name "*counter_collection.csv" | head
