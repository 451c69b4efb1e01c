set -x
timeout 900 python -m pytest tests/test_persist_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r06_tests10.txt
cat gpurun_out/r06_tests10.txt
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_dropin_threads_gpu.py -x -q -k "lean or chunk or config3 or full_size or multi_contig or threads" 2>&1 | tail -8 >> gpurun_out/r06_tests10.txt
tail -8 gpurun_out/r06_tests10.txt
run() { timeout 600 python bench.py --steps 5 --warmup 2 --no-cohort --no-sampler --no-viterbi --no-dropin --no-cpu-baseline 2>gpurun_out/r06_p10_$1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', '%.2f M/s'%(d['value']/1e6), 'ms/step %.2f'%d['ms_per_step'], {a:round(b,2) for a,b in d['kernel_ms'].items()}, d.get('plan','')[-120:])
" >> gpurun_out/r06_persist10.txt; }
rm -f gpurun_out/r06_persist10.txt
run persist_4096
PG_CHUNK_COLS=1024 run persist_1024
PG_CHUNK_COLS=256 run persist_256
cp pangenie_amd/csrc/libpangenie_hmm.so /tmp/lib_orig.so
for e in 1 3; do
python - <<PY
import sys
sys.path.insert(0,'.')
from pangenie_amd import build as b
from pathlib import Path
b.build_hip(force=True, out=Path('/tmp/libe$e/libpangenie_hmm.so'), defines=['PG_PERSIST_EXP=$e'])
PY
cp /tmp/libe$e/libpangenie_hmm.so pangenie_amd/csrc/libpangenie_hmm.so
run persist_exp$e
done
cp /tmp/lib_orig.so pangenie_amd/csrc/libpangenie_hmm.so
cat gpurun_out/r06_persist10.txt
