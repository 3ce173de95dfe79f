# round 3: the default bench line (modes + measured whole-host CPU baseline), timed by the shell as the driver would
O=$GRAFT_REPO_ROOT/gpurun_out/r3N; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
nproc; free -g | head -2
S=$(date +%s)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
E=$(date +%s); echo "wall $((E - S)) s"
tail -c 1500 $O/bench_default.err
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3N"
d=json.loads([l for l in open(O+"/bench_default.json").read().splitlines() if l.startswith("{")][-1])
print("value %.3f M"%(d["value"]/1e6), "frac %.3f"%d["roofline"]["frac"])
for k,m in d["modes"].items(): print("  mode", k[:60], "%.3f M"%(m["value"]/1e6) if isinstance(m,dict) and "value" in m else m)
c=d["cpu_baseline"]; print({k:(v if k not in("sample","one_thread") else str(v)[:260]) for k,v in c.items()})
PY
ps aux | grep -c "bench.py" 
