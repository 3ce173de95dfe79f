# round 3, last tree: full -m gpu suite, smoke(), default bench line
O=$GRAFT_REPO_ROOT/gpurun_out/r3AF; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-300
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3AF"
d=json.loads([l for l in open(O+"/bench_default.json").read().splitlines() if l.startswith("{")][-1])
print("value %.3f M frac %.3f"%(d["value"]/1e6, d["roofline"]["frac"]), {k[:20]:round(m["value"]/1e6,3) for k,m in d["modes"].items()}, round(d["cpu_baseline"]["value"]), d["cpu_baseline"]["cores"])
PY
