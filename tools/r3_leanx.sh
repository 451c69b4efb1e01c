#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header -x -k "leanx or config4_shape or panels_vs_oracle or multi_contig_job or chunk_boundaries or lean_kernel or generic_kernel_cross" 2>&1 | tail -8
timeout 600 python tools/bench_leanx.py 20000
