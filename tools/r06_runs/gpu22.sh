set -x
run() { timeout 600 python bench.py --steps 5 --warmup 2 --no-cohort --no-sampler --no-viterbi --no-dropin --no-cpu-baseline 2>gpurun_out/r06_p22_$1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', '%.2f M/s'%(d['value']/1e6), 'ms/step %.2f'%d['ms_per_step'], {a:round(b,2) for a,b in d['kernel_ms'].items()})
" >> gpurun_out/r06_nt22.txt; }
rm -f gpurun_out/r06_nt22.txt
run warm
run default
cp pangenie_amd/csrc/libpangenie_hmm.so /tmp/lib_orig.so
for v in "PG_NT_STORES" "PG_NT_POST" "PG_NT_STORES PG_NT_POST"; do
tag=$(echo $v | tr ' ' '+')
python - <<PY
import sys
sys.path.insert(0,'.')
from pangenie_amd import build as b
from pathlib import Path
b.build_hip(force=True, out=Path('/tmp/libv/libpangenie_hmm.so'), defines="$v".split())
PY
cp /tmp/libv/libpangenie_hmm.so pangenie_amd/csrc/libpangenie_hmm.so
run $tag
done
cp /tmp/lib_orig.so pangenie_amd/csrc/libpangenie_hmm.so
run default_again
cat gpurun_out/r06_nt22.txt
