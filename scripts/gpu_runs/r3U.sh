# round 3: one-workgroup epilogue for small updates (reference cadence) - parity tests of the fused path, then the mode's row
O=$GRAFT_REPO_ROOT/gpurun_out/r3U; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu -k "fused or bench_path or parity or host_api or two_ranks or standardise or checkpoints" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
timeout 200 $B --steps 4 --warmup 1 --cadence reference > $O/ref.json 2>/dev/null
timeout 200 $B --steps 20 --warmup 3 > $O/default.json 2>/dev/null
timeout 200 $B --steps 4 --warmup 1 --cadence reference --algo vdn > $O/ref_vdn.json 2>/dev/null
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3U"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "frac %.3f"%(r.get("frac") or 0), {k:round(v["avg_us"],2) for k,v in d["kernels"].items()})
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
