# round 3: 128-tile GEMM slice depth 32 (default) against 16 (variant)
O=$GRAFT_REPO_ROOT/gpurun_out/r3Q; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests -q -m gpu -k "wide or layers or centralised or gemm or standardise" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
for v in kb32 kb16; do
  if [ $v = kb16 ]; then export MARLHIP_LIB=$R/codebase_amd/csrc/variants/libmarlhip_kb16.so; else unset MARLHIP_LIB; fi
  timeout 200 $B --steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128 > $O/maa2c8p_$v.json 2>/dev/null
  timeout 200 $B --steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/mapporw_$v.json 2>/dev/null
  timeout 200 $B --steps 4 --warmup 1 --hidden 256 > $O/idqn256_$v.json 2>/dev/null
done
unset MARLHIP_LIB
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3Q"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "frac %.3f"%(r.get("frac") or 0))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
