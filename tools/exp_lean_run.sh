#!/bin/bash
# lean-step variants on the GPU box: quick parity of the lean kernels, then per-column times of every built variant
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header -x -k "lean or triangle or multi_contig or chunk_boundaries or unregularized" 2>&1 | tail -3
python tools/exp_lean.py run default $(ls tools/_build/ | sed 's/libpangenie_hmm_PG_LEAN_DEFER/PG_LEAN_DEFER=/;s/.so//')
