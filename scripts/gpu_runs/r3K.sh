# round 3: kernel durations at the reference's own cadence (updates of 32 episodes)
O=$GRAFT_REPO_ROOT/gpurun_out/r3K; mkdir -p $O; R=$GRAFT_REPO_ROOT
B="python $R/bench.py --no-cpu-baseline --no-modes"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_ref --output-format csv -- $B --steps 2 --warmup 1 --cadence reference > $O/ref.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_ref128 --output-format csv -- $B --steps 2 --warmup 1 --cadence reference --hidden 128 > $O/ref128.log 2>&1
cd $R; python - <<'PY'
import csv,glob,os,json
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3K"
for d in sorted(glob.glob(O+"/st_*")):
    n=os.path.basename(d)[3:]
    try:
        l=[x for x in open(O+"/"+n+".log").read().splitlines() if x.startswith("{")][-1]; j=json.loads(l); print("==",n,"%.3f M"%(j["value"]/1e6),"ms %.2f"%j["ms_per_step"])
    except Exception as e: print("==",n,"ERR",e)
    for f in glob.glob(d+"/*/*kernel_stats.csv"):
        for r in list(csv.DictReader(open(f)))[:8]:
            print("   %-80s calls %6s avg_us %9.2f pct %s"%(r["Name"].replace("marl::","")[:80],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
