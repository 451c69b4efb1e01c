set -x
for k in 4096 8192 16384; do
  PG_CHUNK_COLS=$k timeout 600 python bench.py --steps 5 --warmup 2 --no-cohort --no-sampler --no-viterbi --no-dropin --no-cpu-baseline > gpurun_out/r06_chunk_$k.json 2> gpurun_out/r06_chunk_$k.err
done
