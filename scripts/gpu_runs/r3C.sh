# round 3, third GPU call: the opt-in split-fp16 learner - its gate (tests/test_gpu_split16.py), then its bench row next to the default
O=$GRAFT_REPO_ROOT/gpurun_out/r3C; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_split16.py tests/test_gpu_bench_path_vs_oracle.py -q -m gpu > $O/tests_split16.log 2>&1; echo "split16 tests rc=$?"; tail -25 $O/tests_split16.log | cut -c1-250
B="python $R/bench.py --no-cpu-baseline --no-modes"
timeout 200 $B --steps 20 --warmup 3 > $O/bench_f32.json 2>$O/bench_f32.err
timeout 200 $B --steps 20 --warmup 3 --split16 > $O/bench_split16.json 2>$O/bench_split16.err; tail -3 $O/bench_split16.err
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3C"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "frac %.3f"%(r.get("frac") or 0), "us %.0f"%(r.get("avg_launch_us") or 0))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_split16 --output-format csv -- $B --steps 10 --warmup 2 --split16 > $O/stats_split16.log 2>&1
cd $R; python - <<'PY'
import csv,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3C"
for f in glob.glob(O+"/stats_split16/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:7]: print("%-80s calls %6s avg_us %9.2f pct %5s"%(r["Name"][:80],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
