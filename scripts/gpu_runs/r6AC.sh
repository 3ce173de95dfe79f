#!/bin/bash
# round 6, run AC: the whole GPU suite + smoke on the final tree
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6AC"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 2700 python -m pytest tests -m gpu -q --maxfail=10 --durations=8 -rA -s ) > $O/pytest_gpu_full.log 2>&1
grep "at-size\|\[plan\]" $O/pytest_gpu_full.log | cut -c1-230 > $O/observed_deviations.txt
grep -v "^PASSED\|at-size\|^\[" $O/pytest_gpu_full.log | tail -22
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
