set -x
run() { timeout 600 python bench.py --steps 5 --warmup 2 --cohort-only --cohort-key cohort_h64m --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['cohort_h64m']
print('$1', '%.1f M/s'%(r['value']/1e6), 'ms/step %.2f'%r['ms_per_step'], {a:round(b,2) for a,b in r['kernel_ms'].items()})
" >> gpurun_out/r06_h64m_44.txt; }
rm -f gpurun_out/r06_h64m_44.txt
run ew4
cp pangenie_amd/csrc/libpangenie_hmm.so /tmp/lib_orig.so
for v in "PG_LX2_EW=2" "PG_LX2_EW=8" "PG_LX2_ONEHOT=0" "PG_LX2_TWO=0"; do
python - <<PY
import sys
sys.path.insert(0,'.')
from pangenie_amd import build as b
from pathlib import Path
b.build_hip(force=True, out=Path('/tmp/libv/libpangenie_hmm.so'), defines=["$v"])
PY
cp /tmp/libv/libpangenie_hmm.so pangenie_amd/csrc/libpangenie_hmm.so
run $v
done
cp /tmp/lib_orig.so pangenie_amd/csrc/libpangenie_hmm.so
cat gpurun_out/r06_h64m_44.txt
