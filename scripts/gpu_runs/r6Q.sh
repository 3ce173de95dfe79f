#!/bin/bash
# round 6, run Q: the two-rank worker (both exchanges in-library, two processes on one device) eight times in a row - how often does a
# lane run into its peer timeout there, and on which lane / rank
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6Q"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
export MASTER_ADDR=127.0.0.1 MARLHIP_P2P=1 MARLHIP_P2P_TIMEOUT_MS=20000 MARLHIP_P2P_SHARED_DEVICE=1
for k in 1 2 3 4 5 6 7 8; do
  ( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + k)) tests/two_rank_worker.py ) > $O/run_$k.log 2>&1
  echo "run $k: $(grep -c TWO_RANK_OK $O/run_$k.log) ok; $(grep -m1 'a p2p lane' $O/run_$k.log | cut -c1-300)"
done
