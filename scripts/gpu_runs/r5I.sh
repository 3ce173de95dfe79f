#!/bin/bash
# round 5, run I: the actors' forward pass kept by the rollout - bit-equality tests, the A2C at-size tests on the kept path, the AC rows
O=$GRAFT_REPO_ROOT/gpurun_out/r5I; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_ac_keep.py -x -q 2>&1 | tail -15 > $O/keep_tests.txt
cat $O/keep_tests.txt
timeout 600 python -m pytest tests/test_gpu_at_size_vs_oracle.py -x -q -k "config4 or maa2c" 2>&1 | tail -8 | tee $O/at_size.txt
for mode in keep nokeep; do
  if [ $mode = nokeep ]; then export MARLHIP_AC_NO_KEEP=1; else unset MARLHIP_AC_NO_KEEP; fi
  timeout 300 python bench.py --algo ia2c --env-name rware:rware-tiny-4ag-v2 --envs 2048 --time-limit 500 --hidden 128 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > $O/ia2c_rware_$mode.json
  timeout 300 python bench.py --algo ia2c --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > $O/ia2c_lbf64_$mode.json
  timeout 300 python bench.py --algo ia2c --hidden 128 --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > $O/ia2c_lbf128_$mode.json
  timeout 300 python bench.py --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --hidden 128 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > $O/maa2c_$mode.json
done
python - <<'PY'
import json, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5I"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads(open(f).read())
        print(os.path.basename(f), round(d["value"]/1e6,3), "M", round(d["ms_per_step"],3), "ms", d.get("roofline",{}).get("frac"))
    except Exception as e:
        print(os.path.basename(f), "ERR", open(f).read()[-300:])
PY
