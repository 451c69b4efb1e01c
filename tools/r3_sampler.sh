#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_sampler.py -m gpu -q --no-header -x -k "then_job" 2>&1 | tail -15
