#!/bin/bash
# round 6, run W: the two-rank worker 96 times with MARLHIP_TWO_RANK_DIAG=1 - when a lane's wait runs into its bound, where is each rank's
# HOST (faulthandler dumps its stack after 8 s; the per-round enqueue stamps of both ranks)
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6W"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
export MASTER_ADDR=127.0.0.1 MARLHIP_P2P=1 MARLHIP_P2P_TIMEOUT_MS=20000 MARLHIP_P2P_SHARED_DEVICE=1 MARLHIP_TWO_RANK_DIAG=1
bad=0
for k in $(seq 1 96); do
  ( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + k % 50)) tests/two_rank_worker.py ) > $O/run_$k.log 2>&1
  if grep -q TWO_RANK_OK $O/run_$k.log; then rm $O/run_$k.log; else bad=$((bad + 1)); echo "run $k FAILED: $(grep -m1 'AssertionError: rank' $O/run_$k.log | cut -c1-220)"; fi
  if [ $bad -ge 4 ]; then break; fi
done
echo "runs $k, failed $bad"
