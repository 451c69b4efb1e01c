for L in l2d1; do
echo "== $L"; PANGENIE_HMM_LIB=$PWD/tools/_build/libpangenie_hmm_$L.so timeout 600 python bench.py --cohort-only --no-sampler --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['cohort']['value'], d['cohort']['ms_per_step'], d['cohort']['kernel_ms'])"
done
