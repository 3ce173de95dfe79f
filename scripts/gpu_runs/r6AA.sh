#!/bin/bash
# round 6, run AA: the default bench line and the BASELINE-config rows once more on the final tree (every kernel was recompiled when AgentMap
# gained its depth field; the hashed kernel sources are unchanged)
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6AA"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
( timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default_line.err )
B="python bench.py --no-cpu-baseline --no-modes"
: > $O/rows.jsonl
run() { timeout 400 $B "$@" 2>/dev/null | grep '^{' >> $O/rows.jsonl; }
run --steps 60 --warmup 5
run --steps 30 --warmup 3 --hidden 128
run --steps 6 --warmup 2 --algo vdn --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192 --hidden 128
run --steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128
run --steps 10 --warmup 2 --rnn
run --steps 4 --warmup 1 --cadence reference
python - <<'PY'
import json,os
o=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r6AA")
d=json.loads(open(o+"/bench_default_line.json").read().strip().splitlines()[-1])
print("default", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["cpu_baseline"]["value"])
for l in open(o+"/rows.jsonl"):
    r=json.loads(l); print(r["metric"][-30:], round(r["value"]/1e6,3), r["ms_per_step"], (r.get("roofline") or {}).get("frac"))
PY
