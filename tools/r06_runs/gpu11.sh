set -x
run() { timeout 600 python bench.py --steps 5 --warmup 2 --no-cohort --no-sampler --no-viterbi --no-dropin --no-cpu-baseline 2>gpurun_out/r06_p11_$1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', '%.2f M/s'%(d['value']/1e6), 'ms/step %.2f'%d['ms_per_step'], {a:round(b,2) for a,b in d['kernel_ms'].items()})
" >> gpurun_out/r06_persist11.txt; }
rm -f gpurun_out/r06_persist11.txt
run persist_pb8
PG_POST_BLOCKS=6 run persist_pb6
PG_POST_BLOCKS=4 run persist_pb4
PG_POST_BLOCKS=3 run persist_pb3
cp pangenie_amd/csrc/libpangenie_hmm.so /tmp/lib_orig.so
python - <<PY
import sys
sys.path.insert(0,'.')
from pangenie_amd import build as b
from pathlib import Path
b.build_hip(force=True, out=Path('/tmp/libp/libpangenie_hmm.so'), defines=['PG_POLL_SLEEPS=16'])
PY
cp /tmp/libp/libpangenie_hmm.so pangenie_amd/csrc/libpangenie_hmm.so
run persist_poll16_pb8
PG_POST_BLOCKS=5 run persist_poll16_pb5
cp /tmp/lib_orig.so pangenie_amd/csrc/libpangenie_hmm.so
cat gpurun_out/r06_persist11.txt
