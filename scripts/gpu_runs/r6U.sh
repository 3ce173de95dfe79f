#!/bin/bash
# round 6, run U: the recurrent + actor-critic tests again after the scratch sizing fix
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6U"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 2400 python -m pytest tests/test_gru_stacked.py tests/test_gru.py tests/test_gpu_layers.py tests/test_gpu_ac_update.py -m gpu -q --maxfail=8 --durations=5 ) > $O/pytest.log 2>&1
tail -25 $O/pytest.log | cut -c1-330
