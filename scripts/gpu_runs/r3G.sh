# round 3: actor-critic learner with the stored second hidden layer (hidden 128) - suite, AC rows, kernel stats of the IA2C rware row
O=$GRAFT_REPO_ROOT/gpurun_out/r3G; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
timeout 200 $B --steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/ia2c_rware.json 2>/dev/null
timeout 200 $B --steps 5 --warmup 1 --algo ia2c --hidden 128 > $O/ia2c_h128.json 2>/dev/null
timeout 200 $B --steps 5 --warmup 1 --algo mappo --hidden 128 > $O/mappo_h128.json 2>/dev/null
timeout 200 $B --steps 5 --warmup 1 --algo ippo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/ippo_rware.json 2>/dev/null
timeout 200 $B --steps 20 --warmup 3 > $O/default.json 2>/dev/null
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3G"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "frac %.3f"%(r.get("frac") or 0), "us %.0f"%(r.get("avg_launch_us") or 0))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_ia2c_rware --output-format csv -- $B --steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/stats_ia2c.log 2>&1
cd $R; python - <<'PY'
import csv,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3G"
for d in ("stats_ia2c_rware",):
    for f in glob.glob(O+"/"+d+"/*/*kernel_stats.csv"):
        print("==",d)
        for r in list(csv.DictReader(open(f)))[:12]: print("%-84s calls %6s avg_us %9.2f pct %5s"%(r["Name"][:84],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
