# round 3: no gradient-norm launch where nobody uses the norm (QMIX's mixer block, unclipped actor-critic steps)
O=$GRAFT_REPO_ROOT/gpurun_out/r3AE; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu -k "ac_update or qmix or parity or gru or standardise or two_ranks or host_api or layers or checkpoints" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
timeout 200 $B --steps 20 --warmup 3 --algo qmix > $O/qmix2p.json 2>/dev/null
timeout 200 $B --steps 50 --warmup 5 --algo ippo > $O/ippo64.json 2>/dev/null
timeout 200 $B --steps 100 --warmup 5 --algo ia2c > $O/ia2c64.json 2>/dev/null
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3AE"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "frac %.3f"%(r.get("frac") or 0))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
