#!/bin/bash
# round 6, run G: tp_bwd with one row block per step and TWO workgroups per compute unit (launch_bounds(256, 2), 256 registers): does a second
# wave per SIMD hide the barrier / LDS-exchange latency of the pass?  H128 rows + kernel stats; the H128 goldens for correctness
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6G"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_path_vs_oracle.py -x -q -m gpu -k "128 or H128 or hidden128" 2>&1 | tail -4
B="python $R/bench.py --no-cpu-baseline --no-modes"
: > $O/rows.jsonl
run() { timeout 400 $B "$@" 2>/dev/null | grep '^{' >> $O/rows.jsonl; }
run --steps 10 --warmup 2 --hidden 128
run --steps 3 --warmup 1 --algo vdn --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192 --hidden 128
run --steps 8 --warmup 2 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --envs 2048 --hidden 128 --time-limit 500
python - <<'PY'
import json, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6G"
for l in open(O+"/rows.jsonl"):
    d=json.loads(l); r=d["roofline"]; c=d["config"]; print(d["metric"][-40:], round(d["value"]/1e6,3), round(d["ms_per_step"],3), "lossgrad us", round(r["avg_launch_us"],1), "frac", round(r["frac"],3))
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_h128 --output-format csv -- $B --steps 6 --warmup 2 --hidden 128 --no-kernel-timing > $O/stats_h128.log 2>&1
f=$(find $O/stats_h128 -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-60,150-260
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
