#!/bin/bash
# round 6, run K: tp_fwd's epilogue on two waves per row block: H128 goldens / at-size + rows + kernel stats; mixed use_rnn with the full shape list
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6K"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_path_vs_oracle.py tests/test_gpu_qmix.py tests/test_gpu_standardise.py tests/test_gpu_sharing.py tests/test_gru.py "tests/test_gpu_at_size_vs_oracle.py::test_config3_vdn_15x15_4p5f_H128_B8192_vs_oracle_port" -x -q -m gpu 2>&1 | tail -4
B="python $R/bench.py --no-cpu-baseline --no-modes"
: > $O/rows.jsonl
run() { timeout 400 $B "$@" 2>/dev/null | grep '^{' >> $O/rows.jsonl; }
run --steps 10 --warmup 2 --hidden 128
run --steps 3 --warmup 1 --algo vdn --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192 --hidden 128
run --steps 3 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128
python - <<'PY'
import json, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6K"
for l in open(O+"/rows.jsonl"):
    d=json.loads(l); r=d["roofline"]; print(d["metric"][-40:], round(d["value"]/1e6,3), round(d["ms_per_step"],3), "lossgrad us", round(r["avg_launch_us"],1), "frac", round(r["frac"],3))
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_h128 --output-format csv -- $B --steps 6 --warmup 2 --hidden 128 --no-kernel-timing > $O/stats_h128.log 2>&1
f=$(find $O/stats_h128 -name "*kernel_stats.csv" | head -1); head -4 $f | cut -c1-60,150-260
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
