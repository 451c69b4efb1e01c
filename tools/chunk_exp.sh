for K in 2048 4096 8192; do
  echo "== PG_CHUNK_COLS=$K"; PG_CHUNK_COLS=$K timeout 600 python bench.py --no-cohort --no-sampler --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms']['k_sweep_phase1'], d['kernel_ms']['k_sweep_phase2'])"
done
