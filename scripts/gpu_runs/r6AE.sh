#!/bin/bash
# round 6, run AE: a stacked recurrent family next to a feed-forward one + the recurrent / actor-critic tests on the changed entry points
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6AE"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 2400 python -m pytest tests/test_gru_stacked.py tests/test_gru.py tests/test_abi.py tests/test_gpu_ac_update.py tests/test_gpu_layers.py -m gpu -q --maxfail=8 --durations=3 ) > $O/pytest.log 2>&1
tail -30 $O/pytest.log | cut -c1-300
