#!/bin/bash
# round 6, run Z: the two-rank / data-parallel tests and the bench launch tests on the capped exchange; bench --gpus 2 --algo ia2c line
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6Z"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 2400 python -m pytest tests/test_gpu_two_ranks.py tests/test_bench_launch.py tests/test_parallel_gloo.py -m gpu -q --maxfail=8 --durations=5 -s ) > $O/pytest.log 2>&1
grep -c "second attempt" $O/pytest.log; tail -12 $O/pytest.log | cut -c1-250
MARLHIP_BENCH_BACKEND=gloo MARLHIP_BENCH_ONE_DEVICE=1 MARLHIP_P2P_SHARED_DEVICE=1 MARLHIP_P2P_TIMEOUT_MS=20000 timeout 900 python bench.py --gpus 2 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 1024 --hidden 128 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/ia2c_2ranks.json
python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r6Z/ia2c_2ranks.json")).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["rccl_ranks"].get("exchange"), d["rccl_ranks"].get("side_lane"), d["roofline"].get("critic_backward_overlaps_next_rollout"))
PY
