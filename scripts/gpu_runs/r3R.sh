# round 3: the 71-wide warehouse rows at hidden 64 on the LDS-resident learner kernel (two-round fold) instead of the tensor-parallel one
O=$GRAFT_REPO_ROOT/gpurun_out/r3R; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu -k "rware or parity or fused or bench_path or ac_update or qmix or standardise or two_ranks" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
timeout 200 $B --steps 3 --warmup 1 --algo idqn --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64 > $O/idqn_rware.json 2>/dev/null
timeout 200 $B --steps 3 --warmup 1 --algo qmix --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64 > $O/qmix_rware.json 2>/dev/null
timeout 200 $B --steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64 > $O/ia2c_rware64.json 2>/dev/null
timeout 200 $B --steps 20 --warmup 3 > $O/default.json 2>/dev/null
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3R"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "frac %.3f"%(r.get("frac") or 0), r.get("kernel"))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
