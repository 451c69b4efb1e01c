set -x
PG_DEBUG_PIPE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cohort --no-sampler --no-viterbi --no-cpu-baseline 2> gpurun_out/r06_pipe26.err | tail -c 300
grep pipe gpurun_out/r06_pipe26.err | head
