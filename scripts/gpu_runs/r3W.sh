# round 3: two-pass form of the LDS-resident learner with stored hidden layers (VDN / QMIX / standardised IDQN at hidden 64)
O=$GRAFT_REPO_ROOT/gpurun_out/r3W; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu -k "qmix or vdn or standardise or parity or bench_path or fused or two_ranks or rware or host_api or checkpoints or layers" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
for v in stored nostore; do
  if [ $v = nostore ]; then export MARLHIP_LIB=$R/codebase_amd/csrc/variants/libmarlhip_nostore.so; else unset MARLHIP_LIB; fi
  timeout 200 $B --steps 6 --warmup 2 --algo vdn --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192 > $O/vdn4p_$v.json 2>/dev/null
  timeout 200 $B --steps 10 --warmup 2 --algo qmix > $O/qmix2p_$v.json 2>/dev/null
  timeout 200 $B --steps 6 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192 > $O/qmix4p_$v.json 2>/dev/null
  timeout 200 $B --steps 3 --warmup 1 --algo qmix --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64 > $O/qmix_rware_$v.json 2>/dev/null
done
unset MARLHIP_LIB
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3W"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "frac %.3f"%(r.get("frac") or 0), {k:round(v["avg_us"],1) for k,v in d["kernels"].items()})
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
