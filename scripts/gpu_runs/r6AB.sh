#!/bin/bash
# round 6, run AB: stacked GRU layers with parameter sharing / standardise_returns
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6AB"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 1500 python -m pytest tests/test_gru_stacked.py -m gpu -q --maxfail=6 --durations=5 ) > $O/pytest.log 2>&1
tail -30 $O/pytest.log | cut -c1-300
