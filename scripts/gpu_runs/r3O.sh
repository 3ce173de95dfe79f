# round 3: kernel shares of the GEMM-path rows (centralised critics of 8 agents / the warehouse, 256-wide nets)
O=$GRAFT_REPO_ROOT/gpurun_out/r3O; mkdir -p $O; R=$GRAFT_REPO_ROOT
B="python $R/bench.py --no-cpu-baseline --no-modes"
cd /tmp; export TMPDIR=/tmp
run() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_$n --output-format csv -- $B "$@" > $O/$n.log 2>&1; }
run maa2c8p --steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128
run mapporw --steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128
run idqn256 --steps 4 --warmup 1 --hidden 256
cd $R; python - <<'PY'
import csv,glob,os,json
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3O"
for d in sorted(glob.glob(O+"/st_*")):
    n=os.path.basename(d)[3:]
    try:
        l=[x for x in open(O+"/"+n+".log").read().splitlines() if x.startswith("{")][-1]; j=json.loads(l); print("==",n,"%.3f M"%(j["value"]/1e6),"ms %.2f"%j["ms_per_step"], "frac %.3f"%j["roofline"]["frac"])
    except Exception as e: print("==",n,"ERR",e)
    for f in glob.glob(d+"/*/*kernel_stats.csv"):
        for r in list(csv.DictReader(open(f)))[:9]:
            print("   %-90s calls %6s avg_us %9.2f pct %s"%(r["Name"].replace("marl::","")[:90],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
