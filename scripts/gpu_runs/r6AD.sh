#!/bin/bash
# round 6, run AD: recurrent families of two depths + every recurrent / actor-critic test on the changed a2c_core.h / gru_ac.hip
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6AD"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 2400 python -m pytest tests/test_gru_stacked.py tests/test_gru.py tests/test_abi.py tests/test_ac_collector.py tests/test_action_masks.py tests/test_gpu_ac_keep.py tests/test_gpu_ac_update.py tests/test_gpu_at_size_vs_oracle.py tests/test_gpu_layers.py tests/test_gpu_two_ranks.py tests/test_bench_launch.py -m gpu -q --maxfail=8 --durations=3 ) > $O/pytest.log 2>&1
tail -25 $O/pytest.log | cut -c1-300
