#!/bin/bash
# round 6, run S: stacked GRU layers (tests/test_gru_stacked.py) + the one-layer recurrent tests on the changed kernels
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6S"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 1500 python -m pytest tests/test_gru_stacked.py tests/test_gru.py -m gpu -q --maxfail=6 --durations=5 ) > $O/pytest.log 2>&1
tail -40 $O/pytest.log | cut -c1-400
