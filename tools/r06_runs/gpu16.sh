set -x
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_persist_gpu.py -x -q -k "lean or chunk or config3 or full_size or multi_contig or persist or triangle" 2>&1 | tail -8 > gpurun_out/r06_tests16.txt
cat gpurun_out/r06_tests16.txt
run() { timeout 600 python bench.py --steps 5 --warmup 2 --no-cohort --no-sampler --no-viterbi --no-dropin --no-cpu-baseline 2>gpurun_out/r06_p16_$1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', '%.2f M/s'%(d['value']/1e6), 'ms/step %.2f'%d['ms_per_step'], {a:round(b,2) for a,b in d['kernel_ms'].items()})
" >> gpurun_out/r06_pretotal16.txt; }
rm -f gpurun_out/r06_pretotal16.txt
run pretotal
cp pangenie_amd/csrc/libpangenie_hmm.so /tmp/lib_orig.so
python - <<PY
import sys
sys.path.insert(0,'.')
from pangenie_amd import build as b
from pathlib import Path
b.build_hip(force=True, out=Path('/tmp/libp/libpangenie_hmm.so'), defines=['PG_LEAN_PRETOTAL=0'])
PY
cp /tmp/libp/libpangenie_hmm.so pangenie_amd/csrc/libpangenie_hmm.so
run round3_step
cp /tmp/lib_orig.so pangenie_amd/csrc/libpangenie_hmm.so
cat gpurun_out/r06_pretotal16.txt
