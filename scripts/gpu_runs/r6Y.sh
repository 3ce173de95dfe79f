#!/bin/bash
# round 6, run Y: 96 more runs of the two-rank worker on one device with the exchange's grid capped (the default)
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6Y"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
export MASTER_ADDR=127.0.0.1 MARLHIP_P2P=1 MARLHIP_P2P_TIMEOUT_MS=20000 MARLHIP_P2P_SHARED_DEVICE=1 MARLHIP_TWO_RANK_DIAG=1
bad=0
for k in $(seq 1 96); do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + k % 40)) tests/two_rank_worker.py > $O/run.log 2>&1
  if ! grep -q TWO_RANK_OK $O/run.log; then bad=$((bad + 1)); cp $O/run.log $O/fail_$k.log; fi
  if [ $bad -ge 4 ]; then break; fi
done
echo "capped: $k runs, $bad failed"
