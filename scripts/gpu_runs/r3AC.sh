# round 3: tensor-parallel kernels with 8 waves per workgroup (one hidden tile each, two waves per SIMD) against 4
O=$GRAFT_REPO_ROOT/gpurun_out/r3AC; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
export MARLHIP_LIB=$R/codebase_amd/csrc/variants/libmarlhip_tpw8.so
timeout 1200 python -m pytest tests -q -m gpu -k "parity or ac_update or qmix or layers or rware" > $O/tests_w8.log 2>&1; echo "tests(w8) rc=$?"; tail -4 $O/tests_w8.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
for v in w8 w4; do
  if [ $v = w4 ]; then unset MARLHIP_LIB; fi
  timeout 200 $B --hidden 128 --steps 20 --warmup 3 > $O/h128_$v.json 2>/dev/null
  timeout 200 $B --steps 50 --warmup 5 --algo ia2c --hidden 128 > $O/ia2c128_$v.json 2>/dev/null
  timeout 200 $B --hidden 128 --steps 6 --warmup 2 --algo vdn --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192 > $O/vdn128_$v.json 2>/dev/null
  timeout 200 $B --steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/ia2c_rware_$v.json 2>/dev/null
done
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3AC"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "frac %.3f"%(r.get("frac") or 0), {k[:24]:round(v["avg_us"],1) for k,v in d["kernels"].items()})
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
