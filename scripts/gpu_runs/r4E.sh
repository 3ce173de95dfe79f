# in-library peer-to-peer gradient exchange: two ranks on one GPU (both exchanges, same final parameters), the torchrun entry points,
# bench.py --gpus 2 on one device, and the forced one-rank dist profile (launches per update)
O=$GRAFT_REPO_ROOT/gpurun_out/r4E; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_two_ranks.py tests/test_bench_launch.py -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.log
cd /tmp; export TMPDIR=/tmp
MARLHIP_BENCH_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_dist --output-format csv -- python $R/bench.py --no-cpu-baseline --no-modes --steps 20 --warmup 3 > $O/dist.log 2>&1
cd $R; grep '^{' $O/dist.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print(d['value'],d['ms_per_step'],d['rccl_ranks'])"
python - <<'PY'
import csv,glob,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4E"
for f in glob.glob(O+"/st_dist/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:12]: print("%-100s calls %6s avg_us %9.2f pct %5s"%(r["Name"].replace("marl::","")[:100],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
