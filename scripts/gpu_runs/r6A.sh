#!/bin/bash
# round 6, run A: the split exchange beside the deferred critics (two ranks on one GPU), the at-size overlap test, the N = 2 bench record
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6A"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_ac_keep.py tests/test_gpu_two_ranks.py::test_two_ranks_stay_bit_identical_idqn_qmix_a2c tests/test_bench_launch.py -x -q -m gpu -rA 2>&1 | tail -40 > $O/t1.log
tail -15 $O/t1.log
timeout 900 python -m pytest "tests/test_gpu_at_size_vs_oracle.py::test_config4_two_rounds_through_update_async_overlap_exactly_as_bench_drives_them" -x -q -m gpu -rA -s 2>&1 | tail -40 > $O/t2.log
tail -25 $O/t2.log
( MARLHIP_BENCH_BACKEND=gloo MARLHIP_BENCH_ONE_DEVICE=1 MARLHIP_P2P_SHARED_DEVICE=1 MARLHIP_P2P_TIMEOUT_MS=20000 timeout 600 python bench.py --gpus 2 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 1024 --hidden 128 --steps 8 --warmup 2 --no-cpu-baseline > $O/ia2c_2ranks.json 2> $O/ia2c_2ranks.err ); echo "ia2c 2 ranks exit $?"; tail -3 $O/ia2c_2ranks.err
python - <<'PY'
import json, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6A"
try:
    d=json.loads([l for l in open(O+"/ia2c_2ranks.json") if l.startswith("{")][-1]); print(d["value"], d["ms_per_step"], d["rccl_ranks"], d["roofline"].get("critic_backward_overlaps_next_rollout"), d["roofline"].get("whole_round"))
except Exception as e: print("no line", e)
PY
