#!/bin/bash
# round 5, run J: kept forward pass incl. PPO (old log-probs + first epoch) - tests, then the PPO rows with / without
O=$GRAFT_REPO_ROOT/gpurun_out/r5J; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_ac_keep.py tests/test_gpu_ac_update.py tests/test_ac_collector.py tests/test_gpu_rware.py -x -q 2>&1 | tail -15 | tee $O/tests.txt
for mode in keep nokeep; do
  if [ $mode = nokeep ]; then export MARLHIP_AC_NO_KEEP=1; else unset MARLHIP_AC_NO_KEEP; fi
  timeout 300 python bench.py --algo mappo --env-name rware:rware-tiny-4ag-v2 --envs 2048 --time-limit 500 --hidden 128 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > $O/mappo_rware_$mode.json
  timeout 300 python bench.py --algo ippo --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > $O/ippo_lbf64_$mode.json
done
python - <<'PY'
import json, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5J"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads(open(f).read())
        print(os.path.basename(f), round(d["value"]/1e6,3), "M", round(d["ms_per_step"],3), "ms", d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("actor_forward_kept"))
    except Exception as e:
        print(os.path.basename(f), "ERR", open(f).read()[-300:])
PY
