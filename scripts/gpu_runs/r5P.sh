#!/bin/bash
# round 5, run P: config 4's rows over 20 rounds (the critics of the LAST timed update finish inside the timed region, beside no rollout:
# 7.3 ms over K rounds), the default line with the 16-round config-4 mode row
O="${GRAFT_REPO_ROOT:?}/gpurun_out/prof5k"; mkdir -p "$O"; rm -f "$O/matrix_ac.jsonl"; R=$GRAFT_REPO_ROOT; cd $R
B="python $R/bench.py --no-cpu-baseline --no-modes"
: > $O/matrix_ac.jsonl
run() { timeout 400 $B "$@" 2>/dev/null | grep '^{' >> $O/matrix_ac.jsonl; }
RW="--algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048"
run --steps 20 --warmup 2 $RW --hidden 128
MARLHIP_AC_NO_OVERLAP=1 run --steps 20 --warmup 2 $RW --hidden 128
MARLHIP_AC_NO_OVERLAP=1 MARLHIP_AC_NO_KEEP=1 run --steps 20 --warmup 2 $RW --hidden 128
run --steps 20 --warmup 2 $RW --hidden 64
run --steps 200 --warmup 5 --algo ia2c --envs 2048 --hidden 128
wc -l $O/matrix_ac.jsonl
( timeout 600 python $R/bench.py > $O/bench_default_line.json 2> $O/bench_default_line.err ); echo "default line exit code $?"
python - <<'PY'
import json, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof5k"
for l in open(O+"/matrix_ac.jsonl"):
    d=json.loads(l); r=d["roofline"]; print(d["metric"][-30:], d["config"].get("envs_per_gpu"), round(d["value"]/1e6,2), round(d["ms_per_step"],3), r.get("actor_forward_kept"), r.get("critic_backward_overlaps_next_rollout"))
d=json.loads(open(O+"/bench_default_line.json").read())
print(d["value"], d["roofline"]["frac"], d["roofline"]["traffic"])
for k,v in d["modes"].items():
    if "config 4" in k: print(k, v.get("value"), v.get("ms_per_step"))
PY
