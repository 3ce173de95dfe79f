#!/bin/bash
# round 6, run D: plan kernel with the full-batch early-out (headline A/B), plan tests, at-size overlap test with the output-layer bound (seed 27)
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6D"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_update_plan.py tests/test_gpu_fused_epilogue.py -x -q -m gpu -rA -s 2>&1 | grep "plan\]\|passed\|failed\|Error" | tail -12
B="python $R/bench.py --no-cpu-baseline --no-modes"
: > $O/rows.jsonl
run() { timeout 400 $B "$@" 2>/dev/null | grep '^{' >> $O/rows.jsonl; }
run --steps 40 --warmup 3
MARLHIP_NO_PLAN=1 run --steps 40 --warmup 3
run --steps 40 --warmup 3
MARLHIP_NO_PLAN=1 run --steps 40 --warmup 3
python - <<'PY'
import json, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6D"
for l in open(O+"/rows.jsonl"):
    d=json.loads(l); r=d["roofline"]; c=d["config"]; print(round(d["value"]/1e6,2), round(d["ms_per_step"],3), "len", round(c["mean_episode_length"],2), "rows/sampled ep", c["mean_filled_rows_per_sampled_episode"], "lossgrad us", round(r["avg_launch_us"],1), c.get("learns_at_these_hparams"))
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- $B --steps 10 --warmup 2 --no-kernel-timing > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-170
cd $R
MARLHIP_TEST_SEED=27 timeout 900 python -m pytest "tests/test_gpu_at_size_vs_oracle.py::test_config4_two_rounds_through_update_async_overlap_exactly_as_bench_drives_them" -x -q -m gpu -rA -s 2>&1 | grep "at-size\|passed\|failed\|Error" | cut -c1-260 | tail -40
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
