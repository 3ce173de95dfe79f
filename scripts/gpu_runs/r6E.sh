#!/bin/bash
# round 6, run E: the whole GPU suite on the tree with the split exchange, the update plans, the per-round at-size overlap test
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6E"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 --durations=15 -rA -s ) > $O/pytest_gpu_full.log 2>&1
grep "at-size\|\[plan\]" $O/pytest_gpu_full.log | cut -c1-230 > $O/observed_deviations.txt
grep -v "^PASSED\|at-size\|^\[" $O/pytest_gpu_full.log | tail -45
