# round 3: DQN-family rows on the warehouse with the reference-default 128-128 nets, IDQN on the larger LBF tasks
O=$GRAFT_REPO_ROOT/gpurun_out/r3AB; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
B="python $R/bench.py --no-cpu-baseline --no-modes"
timeout 200 $B --steps 3 --warmup 1 --algo idqn --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/idqn_rware128.json 2>/dev/null
timeout 200 $B --steps 3 --warmup 1 --algo vdn --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/vdn_rware128.json 2>/dev/null
timeout 200 $B --steps 6 --warmup 1 --algo idqn --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 64 > $O/idqn_8p64.json 2>/dev/null
timeout 200 $B --steps 6 --warmup 1 --algo idqn --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192 --hidden 64 > $O/idqn_4p64.json 2>/dev/null
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3AB"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "frac %.3f"%(r.get("frac") or 0), {k[:24]:round(v["avg_us"],1) for k,v in d["kernels"].items()})
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
