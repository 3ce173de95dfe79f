#!/bin/bash
# round 6, run I: actor.use_rnn != critic.use_rnn (csrc/mixed_ac.hip) against the reference's goldens + end to end; the recurrent AC tests around it
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6I"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gru.py -x -q -m gpu -k "one_recurrent_family or feed_forward_critics_end_to_end or recurrent_actor_critic or ia2c_and_ippo" 2>&1 | tail -30
