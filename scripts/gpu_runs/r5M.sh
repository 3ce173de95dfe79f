#!/bin/bash
# round 5, run M: the critics' stream owns half of the compute units (hipExtStreamCreateWithCUMask) - tests, then config 4 with either half
O=$GRAFT_REPO_ROOT/gpurun_out/r5M; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_ac_keep.py -x -q -k "overlap or clip" 2>&1 | tail -15 | tee $O/tests.txt
run() { timeout 300 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1; }
RW="--algo ia2c --env-name rware:rware-tiny-4ag-v2 --envs 2048 --time-limit 500 --hidden 128 --steps 6 --warmup 2"
MARLHIP_SIDE_PATTERN=0 run $RW > $O/rware_p0.json
MARLHIP_SIDE_PATTERN=1 run $RW > $O/rware_p1.json
MARLHIP_AC_NO_OVERLAP=1 run $RW > $O/rware_off.json
RW64="--algo ia2c --env-name rware:rware-tiny-4ag-v2 --envs 2048 --time-limit 500 --hidden 64 --steps 6 --warmup 2"
MARLHIP_SIDE_PATTERN=0 run $RW64 > $O/rware64_p0.json
MARLHIP_SIDE_PATTERN=1 run $RW64 > $O/rware64_p1.json
MARLHIP_AC_NO_OVERLAP=1 run $RW64 > $O/rware64_off.json
L="--algo ia2c --envs 2048 --hidden 128 --steps 60 --warmup 5"
MARLHIP_SIDE_PATTERN=0 run $L > $O/lbf2048_p0.json
MARLHIP_SIDE_PATTERN=1 run $L > $O/lbf2048_p1.json
MARLHIP_AC_NO_OVERLAP=1 run $L > $O/lbf2048_off.json
python - <<'PY'
import json, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5M"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads(open(f).read())
        print(os.path.basename(f), round(d["value"]/1e6,3), "M", round(d["ms_per_step"],3), "ms", round(d["roofline"]["frac"],3), d["roofline"].get("critic_backward_overlaps_next_rollout"), {k[:10]: round(v["avg_us"],1) for k,v in d["kernels"].items()})
    except Exception as e:
        print(os.path.basename(f), "ERR", open(f).read()[-400:])
PY
