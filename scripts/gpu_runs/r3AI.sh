# round 3: QMIX index draw inside the pack launch
O=$GRAFT_REPO_ROOT/gpurun_out/r3AI; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests -q -m gpu -k "qmix or rware or gru or layers or two_ranks or standardise or host_api" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
timeout 200 $B --steps 20 --warmup 3 --algo qmix > $O/qmix2p.json 2>/dev/null
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3AI"
for f in sorted(glob.glob(O+"/*.json")):
    d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"])
PY
