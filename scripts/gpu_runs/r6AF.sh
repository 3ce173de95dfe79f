#!/bin/bash
# round 6, run AF: agents with different observation / action sizes in the actor-critic learners + the actor-critic host tests around them
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6AF"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 2400 python -m pytest tests/test_hetero_agents.py tests/test_gpu_ac_update.py tests/test_gpu_layers.py tests/test_action_masks.py tests/test_gpu_host_api.py -m gpu -q --maxfail=8 --durations=3 ) > $O/pytest.log 2>&1
tail -40 $O/pytest.log | cut -c1-300
