set -x
L=pangenie_amd/csrc/libpangenie_hmm.so
cp $L /tmp/lib_new.so
run() { timeout 600 python bench.py --steps 5 --warmup 2 --no-cohort --no-sampler --no-viterbi --no-dropin --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', '%.2f M/s'%(d['value']/1e6), 'ms/step %.2f'%d['ms_per_step'], {a:round(b,2) for a,b in d['kernel_ms'].items()})
" >> gpurun_out/r06_ab54.txt; }
rm -f gpurun_out/r06_ab54.txt
run new
cp tools/r06_runs/_ab/libpangenie_hmm_prewidef.so $L; run old
cp /tmp/lib_new.so $L; run new
cp tools/r06_runs/_ab/libpangenie_hmm_prewidef.so $L; run old
cp /tmp/lib_new.so $L
cat gpurun_out/r06_ab54.txt
