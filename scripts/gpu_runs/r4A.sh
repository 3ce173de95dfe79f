# round 4, first call: the whole -m gpu suite (incl. the new oracle tests at the driver-reported sizes), smoke, the default line (now the
# reference's hyper-parameters), and BEFORE profiles of BASELINE configs 4 and 5 for this round's kernel work
O=$GRAFT_REPO_ROOT/gpurun_out/r4A; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json;d=json.load(open('$O/bench_default.json'));print(d['value'],d['roofline']['frac'],d['roofline']['traffic'],d['config']['lr'],{k:v['value'] for k,v in d['modes'].items()},d['cpu_baseline']['value'])"
B="python $R/bench.py --no-cpu-baseline --no-modes"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_rw128 --output-format csv -- $B --steps 4 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/rw128.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_rw64 --output-format csv -- $B --steps 4 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64 > $O/rw64.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_qmix8p --output-format csv -- $B --steps 2 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128 > $O/qmix8p.log 2>&1
cd $R; python - <<'PY'
import csv,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4A"
for d in ("st_rw128","st_rw64","st_qmix8p"):
    print("==",d)
    for f in glob.glob(O+"/"+d+"/*/*kernel_stats.csv"):
        for r in list(csv.DictReader(open(f)))[:8]: print("%-110s calls %6s avg_us %9.2f pct %5s"%(r["Name"].replace("marl::","")[:110],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
grep -h '^{' $O/rw128.log $O/rw64.log $O/qmix8p.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print(d['metric'],d['value'],d['ms_per_step'])"
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
