#!/bin/bash
# lean-step variants on the GPU box: quick parity of the lean kernels, then per-column times of every built variant
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header -x -k "lean or triangle or multi_contig or chunk_boundaries or unregularized" 2>&1 | tail -3
python tools/exp_lean.py run default PG_LEAN_DPPF=1 PG_LEAN_DPPF=0 PG_LEAN_DEFER=8 PG_LEAN_DEFER=0
