for cfg in "2 4096" "4 4096" "6 4096" "8 4096" "10 4096" "16 4096" "8 8192" "4 8192"; do
  set -- $cfg
  PG_POST_BLOCKS=$1 PG_CHUNK_COLS=$2 python bench.py --workload genome24_h64 --no-cohort --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('post_blocks $1 chunk $2: value %.4g ms %.1f p1 %.1f p2 %.1f' % (d['value'], d['ms_per_step'], d['kernel_ms']['k_sweep_phase1'], d['kernel_ms']['k_sweep_phase2']))
"
done
