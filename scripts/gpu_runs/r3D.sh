# round 3, fourth GPU call: the whole suite on the tree with the opt-in split-fp16 learner in its best measured form, the default
# bench line (modes incl. the split16 row), rocprof stats of the split16 row
O=$GRAFT_REPO_ROOT/gpurun_out/r3D; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3D"
d=json.loads([l for l in open(O+"/bench_default.json").read().strip().splitlines() if l.startswith("{")][-1])
print("default %.3f M frac %.3f"%(d["value"]/1e6, d["roofline"]["frac"]))
for k,v in d["modes"].items(): print("  %-100s %.3f M  frac %.3f  us %.0f"%(k[:100], v["value"]/1e6, v["roofline"]["frac"] or 0, v["roofline"]["avg_launch_us"] or 0))
print("  cpu", d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"])
PY
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-modes"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_split16 --output-format csv -- $B --steps 10 --warmup 2 --split16 > $O/stats_split16.log 2>&1
cd $R; find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
