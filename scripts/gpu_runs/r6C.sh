#!/bin/bash
# round 6, run C: (i) the reference's stale `filled` tails - trained-policy rows with the reference's replay and with clear_stale, planned and
# not; (ii) the per-round at-size overlap test on two seeds, with the positions of the largest gradient deviations
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6C"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
B="python $R/bench.py --no-cpu-baseline --no-modes"
: > $O/rows.jsonl
run() { timeout 400 $B "$@" 2>/dev/null | grep '^{' >> $O/rows.jsonl; }
run --steps 20 --warmup 2 --hparams tuned --pretrain-rounds 1500 --eps-fixed 0.05
run --steps 20 --warmup 2 --hparams tuned --pretrain-rounds 1500 --eps-fixed 0.05 --clear-stale
MARLHIP_NO_PLAN=1 run --steps 20 --warmup 2 --hparams tuned --pretrain-rounds 1500 --eps-fixed 0.05 --clear-stale
run --steps 20 --warmup 2 --hparams tuned --pretrain-rounds 6000 --eps-fixed 0.02 --clear-stale
run --steps 20 --warmup 2 --hparams tuned --pretrain-rounds 1500 --eps-fixed 0.05 --replay-rounds 2000
python - <<'PY'
import json, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6C"
for l in open(O+"/rows.jsonl"):
    d=json.loads(l); r=d["roofline"]; c=d["config"]; print(round(d["value"]/1e6,2), round(d["ms_per_step"],3), "len", round(c["mean_episode_length"],2), "rows/sampled ep", c["mean_filled_rows_per_sampled_episode"], "clear", c["replay_clear_stale"], "ret", round(c["mean_episode_return_last_round"],3), "lossgrad us", round(r["avg_launch_us"],1))
PY
for seed in 27 21; do
MARLHIP_TEST_SEED=$seed timeout 900 python -m pytest "tests/test_gpu_at_size_vs_oracle.py::test_config4_two_rounds_through_update_async_overlap_exactly_as_bench_drives_them" -x -q -m gpu -rA -s 2>&1 | grep "at-size\|passed\|failed\|Error" | tail -30 > $O/t3_$seed.log; cat $O/t3_$seed.log
done
