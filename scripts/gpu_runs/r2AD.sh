B="python bench.py --no-cpu-baseline"
for envs in 4096 8192 16384; do for nw in 1 4; do
MARLHIP_ACOL_NW=$nw timeout 300 $B --steps 3 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs $envs --hidden 128 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ia2c rware H128 envs $envs nw $nw', round(r['value']/1e6,2), round(r['ms_per_step'],1))"
done; done
for envs in 8192 16384 32768; do for nw in 1 2; do
MARLHIP_ACOL_NW=$nw timeout 300 $B --steps 30 --warmup 3 --algo ia2c --envs $envs --hidden 128 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ia2c lbf H128 envs $envs nw $nw', round(r['value']/1e6,2), round(r['ms_per_step'],2))"
MARLHIP_COL_NW=$nw timeout 300 $B --steps 30 --warmup 3 --cadence env-only --envs $envs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('idqn env-only lbf H64 envs $envs nw $nw', round(r['value']/1e6,2), round(r['ms_per_step'],3))"
done; done
