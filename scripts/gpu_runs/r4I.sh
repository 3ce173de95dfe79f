# GEMM-path rows (wide centralised critics): kernel breakdown
O=$GRAFT_REPO_ROOT/gpurun_out/r4I; mkdir -p $O; R=$GRAFT_REPO_ROOT
B="python $R/bench.py --no-cpu-baseline --no-modes"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_maa2c8p --output-format csv -- $B --steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128 > $O/maa2c8p.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_mappo_rw --output-format csv -- $B --steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/mappo_rw.log 2>&1
cd $R; python - <<'PY'
import csv,glob,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4I"
for d in ("st_maa2c8p","st_mappo_rw"):
    print("==",d)
    for f in glob.glob(O+"/"+d+"/*/*kernel_stats.csv"):
        for r in list(csv.DictReader(open(f)))[:12]: print("%-120s calls %6s avg_us %9.2f pct %5s"%(r["Name"].replace("marl::","")[:120],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
grep -h '^{' $O/maa2c8p.log $O/mappo_rw.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print(d['metric'],d['value'],d['ms_per_step'],d['roofline']['frac'])"
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
