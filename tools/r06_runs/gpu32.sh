set -x
run() { timeout 600 python bench.py --steps 5 --warmup 2 --no-cohort --no-sampler --no-viterbi --no-dropin --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms', round(d['ms_per_step'],2), {a:round(b,2) for a,b in d['kernel_ms'].items() if 'sweep' in a})
" >> gpurun_out/r06_post32.txt; }
rm -f gpurun_out/r06_post32.txt
run warm
run cap8
PG_POST_BLOCKS=7 run cap7
PG_POST_BLOCKS=6 run cap6
PG_POST_BLOCKS=5 run cap5
cp pangenie_amd/csrc/libpangenie_hmm.so /tmp/lib_orig.so
for nb in 2 4; do
python - <<PY
import sys
sys.path.insert(0,'.')
from pangenie_amd import build as b
from pathlib import Path
b.build_hip(force=True, out=Path('/tmp/libv/libpangenie_hmm.so'), defines=['PG_SCRATCH_BUFS=${nb}u'])
PY
cp /tmp/libv/libpangenie_hmm.so pangenie_amd/csrc/libpangenie_hmm.so
run bufs$nb
done
cp /tmp/lib_orig.so pangenie_amd/csrc/libpangenie_hmm.so
cat gpurun_out/r06_post32.txt
