#!/bin/bash
# round 6, run T: stacked GRU layers in the actor-critic learners + every recurrent / actor-critic test on the changed a2c_core.h
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6T"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 2400 python -m pytest tests/test_gru_stacked.py tests/test_gru.py tests/test_ac_collector.py tests/test_action_masks.py tests/test_gpu_ac_keep.py tests/test_gpu_ac_update.py tests/test_gpu_at_size_vs_oracle.py tests/test_gpu_layers.py -m gpu -q --maxfail=8 --durations=5 ) > $O/pytest.log 2>&1
tail -40 $O/pytest.log | cut -c1-330
