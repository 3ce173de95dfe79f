# round 3: kernel shares of the actor-critic rows at hidden 64 / 128 and of VDN on 15x15-4p
O=$GRAFT_REPO_ROOT/gpurun_out/r3S; mkdir -p $O; R=$GRAFT_REPO_ROOT
B="python $R/bench.py --no-cpu-baseline --no-modes"
cd /tmp; export TMPDIR=/tmp
run() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_$n --output-format csv -- $B "$@" > $O/$n.log 2>&1; }
run ia2c64 --steps 50 --warmup 5 --algo ia2c
run ia2c128 --steps 50 --warmup 5 --algo ia2c --hidden 128
run ippo128 --steps 20 --warmup 3 --algo ippo --hidden 128
run vdn4p --steps 6 --warmup 2 --algo vdn --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192
cd $R; python - <<'PY'
import csv,glob,os,json
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3S"
for d in sorted(glob.glob(O+"/st_*")):
    n=os.path.basename(d)[3:]
    try:
        l=[x for x in open(O+"/"+n+".log").read().splitlines() if x.startswith("{")][-1]; j=json.loads(l); print("==",n,"%.3f M"%(j["value"]/1e6),"ms %.3f"%j["ms_per_step"], "frac %.3f"%j["roofline"]["frac"], "steps", j["steps"])
    except Exception as e: print("==",n,"ERR",e)
    for f in glob.glob(d+"/*/*kernel_stats.csv"):
        for r in list(csv.DictReader(open(f)))[:10]:
            print("   %-92s calls %6s avg_us %9.2f pct %s"%(r["Name"].replace("marl::","")[:92],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
