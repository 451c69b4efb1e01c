#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header -x -k "small16" 2>&1 | tail -5
timeout 600 python tools/bench_small.py 24 512 4096
