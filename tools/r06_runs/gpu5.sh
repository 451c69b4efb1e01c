set -x
run() { timeout 600 python bench.py --steps 5 --warmup 2 --cohort-only --cohort-key cohort_h64m --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['cohort_h64m']
print('$1', '%.1f M/s'%(r['value']/1e6), 'ms/step %.2f'%r['ms_per_step'], {a:round(b,2) for a,b in r['kernel_ms'].items()})
" >> gpurun_out/r06_h64m_variants.txt; }
rm -f gpurun_out/r06_h64m_variants.txt
run slots7_cls
PG_KERNELS=nocls4 run slots7_nocls
python - <<'PY'
import sys
sys.path.insert(0,'.')
from pangenie_amd import build as b
from pathlib import Path
b.build_hip(force=True, out=Path('/tmp/lib3/libpangenie_hmm.so'), defines=['PG_TRI_SLOTS=3'])
PY
cp pangenie_amd/csrc/libpangenie_hmm.so /tmp/lib_orig.so
cp /tmp/lib3/libpangenie_hmm.so pangenie_amd/csrc/libpangenie_hmm.so
run slots3_cls
PG_KERNELS=nocls4 run slots3_nocls
cp /tmp/lib_orig.so pangenie_amd/csrc/libpangenie_hmm.so
cat gpurun_out/r06_h64m_variants.txt
