for K in 4096 16384 32768; do
echo "chr22 K=$K"; PG_CHUNK_COLS=$K timeout 300 python bench.py --workload chr22_h64 --no-cohort --no-sampler --no-viterbi --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_ms'))"
done
for K in 4096 8192; do
echo "genome24 K=$K"; PG_CHUNK_COLS=$K timeout 300 python bench.py --no-cohort --no-sampler --no-viterbi --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_ms'))"
done
