#!/bin/bash
# round 6, run R: the two-rank worker, two processes on one device, 16 runs with the critics' stream on half of the compute units (the
# default) and 16 with an unmasked one (MARLHIP_SIDE_SHARE=100): where do the side lane's peer timeouts of run Q come from
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6R"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
export MASTER_ADDR=127.0.0.1 MARLHIP_P2P=1 MARLHIP_P2P_TIMEOUT_MS=20000 MARLHIP_P2P_SHARED_DEVICE=1
for share in 100 50; do
  export MARLHIP_SIDE_SHARE=$share
  for k in $(seq 1 16); do
    ( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + k)) tests/two_rank_worker.py ) > $O/run_${share}_$k.log 2>&1
    echo "share $share run $k: $(grep -c TWO_RANK_OK $O/run_${share}_$k.log) ok; $(grep -m1 'AssertionError: rank' $O/run_${share}_$k.log | cut -c1-200)"
  done
done
