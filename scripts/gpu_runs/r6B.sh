#!/bin/bash
# round 6, run B: the filled-aware plan (plan invariants, planned vs unplanned, the oracle comparisons of the bench path), the per-round
# at-size overlap test, and the headline / trained-policy rows with and without the plan
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6B"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_update_plan.py -x -q -m gpu -rA -s 2>&1 | tail -30 > $O/t1.log; tail -14 $O/t1.log
timeout 1200 python -m pytest tests/test_gpu_bench_path_vs_oracle.py tests/test_gpu_fused_epilogue.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 > $O/t2.log; tail -6 $O/t2.log
timeout 900 python -m pytest "tests/test_gpu_at_size_vs_oracle.py::test_config4_two_rounds_through_update_async_overlap_exactly_as_bench_drives_them" -x -q -m gpu -rA -s 2>&1 | grep -v "^$" | tail -60 > $O/t3.log; grep "at-size\|passed\|failed\|Error" $O/t3.log | tail -30
B="python $R/bench.py --no-cpu-baseline --no-modes"
: > $O/rows.jsonl
run() { timeout 400 $B "$@" 2>/dev/null | grep '^{' >> $O/rows.jsonl; }
run --steps 20 --warmup 3
MARLHIP_NO_PLAN=1 run --steps 20 --warmup 3
run --steps 20 --warmup 2 --hparams tuned --pretrain-rounds 1500 --eps-fixed 0.05
MARLHIP_NO_PLAN=1 run --steps 20 --warmup 2 --hparams tuned --pretrain-rounds 1500 --eps-fixed 0.05
run --steps 20 --warmup 2 --hparams tuned --pretrain-rounds 6000 --eps-fixed 0.02
python - <<'PY'
import json, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6B"
for l in open(O+"/rows.jsonl"):
    d=json.loads(l); r=d["roofline"]; c=d["config"]; print(round(d["value"]/1e6,2), round(d["ms_per_step"],3), "len", round(c["mean_episode_length"],2), "ret", round(c["mean_episode_return_last_round"],3), "lossgrad us", round(r["avg_launch_us"],1), "frac", round(r["frac"],3))
PY
