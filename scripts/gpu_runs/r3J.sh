# round 3: QMIX mixer with the weight gradients inside the online instance's kernel (single-chunk shapes)
O=$GRAFT_REPO_ROOT/gpurun_out/r3J; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -q -m gpu -k "qmix or standardise or rware or two_ranks or gru or host_api or layers" > $O/tests_qmix.log 2>&1; echo "qmix tests rc=$?"; tail -15 $O/tests_qmix.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
cd /tmp; export TMPDIR=/tmp
run() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_$n --output-format csv -- $B "$@" > $O/$n.log 2>&1; }
run q2p --steps 10 --warmup 2 --algo qmix
run q3p --steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-10x10-3p-3f-v3 --envs 8192
cd $R; python - <<'PY'
import csv,glob,os,json
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3J"
for d in sorted(glob.glob(O+"/st_*")):
    n=os.path.basename(d)[3:]
    try:
        l=[x for x in open(O+"/"+n+".log").read().splitlines() if x.startswith("{")][-1]; j=json.loads(l); print("==",n,"%.3f M"%(j["value"]/1e6), j.get("roofline"))
    except Exception as e: print("==",n,"ERR",e)
    for f in glob.glob(d+"/*/*kernel_stats.csv"):
        for r in list(csv.DictReader(open(f)))[:14]:
            print("   %-70s calls %5s avg_us %9.2f"%(r["Name"].replace("marl::","")[:70],r["Calls"],float(r["AverageNs"])/1e3))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
