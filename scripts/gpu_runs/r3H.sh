# round 3: QMIX mixer as per-net fused kernels (first layers -> mixing network in registers) + one weight-gradient launch
O=$GRAFT_REPO_ROOT/gpurun_out/r3H; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -q -m gpu -k "qmix or standardise or rware or two_ranks or gru or host_api or layers" > $O/tests_qmix.log 2>&1; echo "qmix tests rc=$?"; tail -4 $O/tests_qmix.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
timeout 200 $B --steps 10 --warmup 2 --algo qmix > $O/qmix2p.json 2>/dev/null
timeout 200 $B --steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128 > $O/qmix8p.json 2>/dev/null
timeout 200 $B --steps 6 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192 > $O/qmix4p.json 2>/dev/null
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3H"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "frac %.3f"%(r.get("frac") or 0), "us %.0f"%(r.get("avg_launch_us") or 0))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_qmix2p --output-format csv -- $B --steps 10 --warmup 2 --algo qmix > $O/stats_qmix2p.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_qmix8p --output-format csv -- $B --steps 3 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128 > $O/stats_qmix8p.log 2>&1
cd $R; python - <<'PY'
import csv,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3H"
for d in ("stats_qmix2p","stats_qmix8p"):
    for f in glob.glob(O+"/"+d+"/*/*kernel_stats.csv"):
        print("==",d)
        for r in list(csv.DictReader(open(f)))[:12]: print("%-84s calls %6s avg_us %9.2f pct %5s"%(r["Name"][:84],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
