# GEMM path: two-slice prefetch + epilogue operands requested together - parity tests, then the rows and their kernel breakdown
O=$GRAFT_REPO_ROOT/gpurun_out/r4L; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_ac_update.py tests/test_gpu_qmix.py tests/test_gpu_standardise.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
B="python bench.py --no-cpu-baseline --no-modes"
for a in "--steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128" "--steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 10 --warmup 2 --hidden 256"; do
  timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print(d['metric'][25:],'->',round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms frac', round(d['roofline']['frac'],3))"
done 2>&1 | tee $O/rows.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_maa2c8p --output-format csv -- $B --steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128 > $O/maa2c8p.log 2>&1
cd $R; python - <<'PY'
import csv,glob,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4L"
for f in glob.glob(O+"/st_maa2c8p/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:8]: print("%-90s calls %6s avg_us %9.2f pct %5s"%(r["Name"].replace("marl::","")[:90],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
