# round 3: actor-critic learner at hidden 64 with the hidden layers kept from the forward-rows pass
O=$GRAFT_REPO_ROOT/gpurun_out/r3Y; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu -k "ac_update or ac_collector or a2c or ppo or action_masks or rware or two_ranks or standardise or gru or layers or host_api" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
timeout 200 $B --steps 100 --warmup 5 --algo ia2c > $O/ia2c64.json 2>/dev/null
timeout 200 $B --steps 50 --warmup 5 --algo ippo > $O/ippo64.json 2>/dev/null
timeout 200 $B --steps 50 --warmup 5 --algo mappo > $O/mappo64.json 2>/dev/null
timeout 200 $B --steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64 > $O/ia2c_rware64.json 2>/dev/null
timeout 200 $B --steps 100 --warmup 5 --algo ia2c --hidden 128 > $O/ia2c128.json 2>/dev/null
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3Y"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "frac %.3f"%(r.get("frac") or 0), {k[:24]:round(v["avg_us"],1) for k,v in d["kernels"].items()})
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
