# round 3: split16 forward-block scheduling hints A/B (libmarlhip_s{1..4}.so = -DMARL_H16_SCHED=1..4)
O=$GRAFT_REPO_ROOT/gpurun_out/r3E; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
B="python $R/bench.py --no-cpu-baseline --no-modes --split16 --steps 20 --warmup 3"
for v in default s1 s2 s3 s4; do
  if [ $v = default ]; then unset MARLHIP_LIB; else export MARLHIP_LIB=$R/codebase_amd/csrc/variants/libmarlhip_$v.so; fi
  timeout 200 $B > $O/bench_$v.json 2>/dev/null
  timeout 300 python -m pytest tests/test_gpu_split16.py -q -m gpu -x > $O/tests_$v.log 2>&1; echo "$v tests rc=$? $(tail -1 $O/tests_$v.log | cut -c1-80)"
done
unset MARLHIP_LIB
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3E"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "us %.1f"%(r.get("avg_launch_us") or 0))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
