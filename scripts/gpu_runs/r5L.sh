#!/bin/bash
# round 5, run L: the critics' half of an A2C update next to the following rollout (defer_critic_backward) - tests, then the rows with / without
O=$GRAFT_REPO_ROOT/gpurun_out/r5L; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_ac_keep.py tests/test_gpu_ac_update.py -x -q 2>&1 | tail -15 | tee $O/tests.txt
for mode in overlap nooverlap; do
  if [ $mode = nooverlap ]; then export MARLHIP_AC_NO_OVERLAP=1; else unset MARLHIP_AC_NO_OVERLAP; fi
  timeout 300 python bench.py --algo ia2c --env-name rware:rware-tiny-4ag-v2 --envs 2048 --time-limit 500 --hidden 128 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > $O/ia2c_rware_$mode.json
  timeout 300 python bench.py --algo ia2c --steps 60 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > $O/ia2c_lbf64_$mode.json
  timeout 300 python bench.py --algo ia2c --hidden 128 --steps 60 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > $O/ia2c_lbf128_$mode.json
  timeout 300 python bench.py --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --hidden 128 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > $O/maa2c_$mode.json
done
python - <<'PY'
import json, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5L"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads(open(f).read())
        print(os.path.basename(f), round(d["value"]/1e6,3), "M", round(d["ms_per_step"],3), "ms", round(d["roofline"]["frac"],3), d["roofline"].get("critic_backward_overlaps_next_rollout"), {k[:10]: round(v["avg_us"],1) for k,v in d["kernels"].items()})
    except Exception as e:
        print(os.path.basename(f), "ERR", open(f).read()[-400:])
PY
