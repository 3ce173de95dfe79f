O=$GRAFT_REPO_ROOT/gpurun_out/r3Z; mkdir -p $O; R=$GRAFT_REPO_ROOT
B="python $R/bench.py --no-cpu-baseline --no-modes"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_ia2c64 --output-format csv -- $B --steps 50 --warmup 5 --algo ia2c > $O/ia2c64.log 2>&1
cd $R; python - <<'PY'
import csv,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3Z"
for f in glob.glob(O+"/st_ia2c64/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:9]: print("%-100s calls %6s avg_us %9.2f pct %5s"%(r["Name"].replace("marl::","")[:100],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
