# round 3: independent learners on the 71-wide warehouse rows (hidden 64) through the two-pass form with stored hidden layers
O=$GRAFT_REPO_ROOT/gpurun_out/r3AA; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu -k "rware or parity or fused or bench_path or standardise or two_ranks or host_api or checkpoints" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
timeout 200 $B --steps 3 --warmup 1 --algo idqn --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64 > $O/idqn_rware.json 2>/dev/null
timeout 200 $B --steps 3 --warmup 1 --algo idqn --env-name rware:rware-tiny-2ag-v2 --time-limit 500 --envs 2048 --hidden 64 > $O/idqn_rware2.json 2>/dev/null
timeout 200 $B --steps 20 --warmup 3 > $O/default.json 2>/dev/null
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3AA"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "frac %.3f"%(r.get("frac") or 0), {k[:24]:round(v["avg_us"],1) for k,v in d["kernels"].items()})
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
