# round 3: tensor-parallel passes with alternating LDS sets (one barrier fewer per step) - suite, hidden-128 rows, QMIX 2p kernel breakdown
O=$GRAFT_REPO_ROOT/gpurun_out/r3F; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
timeout 200 $B --hidden 128 --steps 20 --warmup 3 > $O/h128.json 2>/dev/null
timeout 200 $B --hidden 128 --steps 6 --warmup 2 --algo vdn --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192 > $O/vdn_h128.json 2>/dev/null
timeout 200 $B --steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128 > $O/qmix8p.json 2>/dev/null
timeout 200 $B --steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/ia2c_rware.json 2>/dev/null
timeout 200 $B --steps 3 --warmup 1 --algo idqn --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64 > $O/idqn_rware.json 2>/dev/null
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3F"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "frac %.3f"%(r.get("frac") or 0), "us %.0f"%(r.get("avg_launch_us") or 0))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_h128 --output-format csv -- $B --steps 10 --warmup 2 --hidden 128 > $O/stats_h128.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_qmix2p --output-format csv -- $B --steps 10 --warmup 2 --algo qmix > $O/stats_qmix2p.log 2>&1
cd $R; python - <<'PY'
import csv,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3F"
for d in ("stats_h128","stats_qmix2p"):
    for f in glob.glob(O+"/"+d+"/*/*kernel_stats.csv"):
        print("==",d)
        for r in list(csv.DictReader(open(f)))[:12]: print("%-84s calls %6s avg_us %9.2f pct %5s"%(r["Name"][:84],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
