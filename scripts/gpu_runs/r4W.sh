# two waves per agent for 4 agents / packs read from L2 (mlp_forward_g_h2): parity, region counters, rows with A/B against HS off
O=$GRAFT_REPO_ROOT/gpurun_out/r4W; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_collector_variants.py tests/test_ac_collector.py tests/test_gpu_rware.py tests/test_gpu_parity.py tests/test_gpu_host_api.py -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.log
V=$R/codebase_amd/csrc/variants/libmarlhip_acolprof.so
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 128 2048 2>&1 | tail -11 | tee $O/prof_rw128.txt
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 64 2048 2>&1 | tail -11 | tee $O/prof_rw64.txt
B="python $R/bench.py --no-cpu-baseline --no-modes"
for e in "X=1" "MARLHIP_COL_HS=1 MARLHIP_ACOL_HS=1"; do
for a in "--steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64" "--steps 3 --warmup 1 --algo idqn --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64" "--steps 100 --warmup 5 --algo ia2c --hidden 128" "--steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 30 --warmup 3 --hidden 128"; do
  env $e timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('$e',d['metric'][25:],'->',round(d['value']/1e6,2),'M', round(d['ms_per_step'],3),'ms', {k[:14]:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
done; done 2>&1 | tee $O/rows.txt
