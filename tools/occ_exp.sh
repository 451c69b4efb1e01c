for L in tri3 tri4; do
echo "== $L"; PG_TRI_ONLY=1 PANGENIE_HMM_LIB=$PWD/tools/_build/libpangenie_hmm_$L.so timeout 600 python bench.py --cohort-only --no-sampler --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['cohort']['value'], d['cohort']['ms_per_step'], d['cohort']['kernel_ms'])"
done
echo "== tri3 parity"; PG_TRI_ONLY=1 PANGENIE_HMM_LIB=$PWD/tools/_build/libpangenie_hmm_tri3.so timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "triangle" 2>&1 | tail -3
