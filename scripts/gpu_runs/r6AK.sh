#!/bin/bash
# round 6, run AK: the multi-GPU bench lines as a first real run would print them - two ranks sharing this box's one MI355X over gloo
# (plumbing evidence, not a scaling point): `bench.py --gpus 2` (idqn: both exchanges, configs 4 / 5 rows) and `--algo ia2c`
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6AK"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
export MARLHIP_BENCH_BACKEND=gloo MARLHIP_BENCH_ONE_DEVICE=1 MARLHIP_P2P_SHARED_DEVICE=1 MARLHIP_P2P_TIMEOUT_MS=20000
timeout 1500 python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline 2>$O/idqn.err | grep '^{' > $O/bench_gpus2_one_device_idqn.json
timeout 900 python bench.py --gpus 2 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 1024 --hidden 128 --steps 10 --warmup 2 --no-cpu-baseline 2>$O/ia2c.err | grep '^{' > $O/bench_gpus2_one_device_ia2c.json
python - <<'PY'
import json,os
o=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r6AK")
d=json.loads(open(o+"/bench_gpus2_one_device_idqn.json").read().strip().splitlines()[-1])
print("idqn", d["value"], d["n_gpus"], json.dumps(d["rccl_ranks"])[:900])
print({k: (v.get("value"), v.get("error")) for k,v in d.get("modes",{}).items() if k.startswith("BASELINE")})
d=json.loads(open(o+"/bench_gpus2_one_device_ia2c.json").read().strip().splitlines()[-1])
print("ia2c", d["value"], d["ms_per_step"], json.dumps(d["rccl_ranks"])[:500])
PY
tail -3 $O/idqn.err | cut -c1-300
