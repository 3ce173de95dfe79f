# round 3: GEMM path with coalescing 16-byte operand loads - its parity tests, then the rows it carries, A/B against the element-wise loads
O=$GRAFT_REPO_ROOT/gpurun_out/r3P; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests -q -m gpu -k "wide or layers or centralised or gemm or standardise" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
for v in vec scalar; do
  if [ $v = scalar ]; then export MARLHIP_WIDE_SCALAR_LOADS=1; else unset MARLHIP_WIDE_SCALAR_LOADS; fi
  timeout 200 $B --steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128 > $O/maa2c8p_$v.json 2>/dev/null
  timeout 200 $B --steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/mapporw_$v.json 2>/dev/null
  timeout 200 $B --steps 4 --warmup 1 --hidden 256 > $O/idqn256_$v.json 2>/dev/null
done
unset MARLHIP_WIDE_SCALAR_LOADS
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3P"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "frac %.3f"%(r.get("frac") or 0))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_mapporw --output-format csv -- $B --steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/st_mapporw.log 2>&1
cd $R; python - <<'PY'
import csv,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3P"
for f in glob.glob(O+"/st_mapporw/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:8]: print("%-84s calls %6s avg_us %9.2f pct %5s"%(r["Name"].replace("marl::","")[:84],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
