#!/bin/bash
# round 6, run X: is it the spinning workgroups?  The two-rank worker on one device with the exchange's grid capped (32 workgroups, the new
# default) and uncapped (MARLHIP_P2P_MAX_WGS=100000: one workgroup per 1024 floats, as before), alternating blocks of 16 runs on ONE box
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6X"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
python -c "import torch; p = torch.cuda.get_device_properties(0); print('device', p.name, 'compute units', p.multi_processor_count)"
rocm-smi --showcomputepartition 2>/dev/null | grep -i partition | head -3
export MASTER_ADDR=127.0.0.1 MARLHIP_P2P=1 MARLHIP_P2P_TIMEOUT_MS=20000 MARLHIP_P2P_SHARED_DEVICE=1 MARLHIP_TWO_RANK_DIAG=1
for block in capped uncapped capped uncapped; do
  if [ $block = uncapped ]; then export MARLHIP_P2P_MAX_WGS=100000; else unset MARLHIP_P2P_MAX_WGS; fi
  bad=0
  for k in $(seq 1 16); do
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + k)) tests/two_rank_worker.py > $O/run.log 2>&1
    if ! grep -q TWO_RANK_OK $O/run.log; then bad=$((bad + 1)); cp $O/run.log $O/fail_${block}_$k.log; fi
    if [ $bad -ge 4 ]; then break; fi
  done
  echo "$block: $k runs, $bad failed"
done
