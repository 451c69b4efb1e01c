set -x
run() { for key in cohort_h16m panels_h16; do
timeout 600 python bench.py --steps 5 --warmup 2 --cohort-only --cohort-key $key --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['$key']
print('$1 $key', '%.1f M/s'%(r['value']/1e6), 'ms/step %.2f'%r['ms_per_step'], {a:round(b,2) for a,b in r['kernel_ms'].items()})
" >> gpurun_out/r06_x20.txt
done; }
rm -f gpurun_out/r06_x20.txt
run twopass
cp pangenie_amd/csrc/libpangenie_hmm.so /tmp/lib_orig.so
python - <<PY
import sys
sys.path.insert(0,'.')
from pangenie_amd import build as b
from pathlib import Path
b.build_hip(force=True, out=Path('/tmp/libp/libpangenie_hmm.so'), defines=['PG_X_ONEPASS'])
PY
cp /tmp/libp/libpangenie_hmm.so pangenie_amd/csrc/libpangenie_hmm.so
run onepass
cp /tmp/lib_orig.so pangenie_amd/csrc/libpangenie_hmm.so
run twopass_again
cat gpurun_out/r06_x20.txt
