#!/bin/bash
# round 6, run L: multi-chunk planning, mixed use_rnn on the warehouse shape, plan tests after the 8-byte length reads
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6L"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_update_plan.py tests/test_gru.py -x -q -m gpu -k "plan or one_recurrent_family or feed_forward_critics" -rA -s 2>&1 | grep -v "^PASSED" | tail -25
