# LBF step without the per-action switch + one env block per workgroup: parity (collector tests, env parity, host API) + region counters + rows
O=$GRAFT_REPO_ROOT/gpurun_out/r4G; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_collector_variants.py tests/test_ac_collector.py tests/test_gpu_rware.py tests/test_gpu_host_api.py tests/test_action_masks.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
V=$R/codebase_amd/csrc/variants/libmarlhip_acolprof.so
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 64 4096 lbforaging:Foraging-8x8-2p-3f-v3 2>&1 | tail -11 | tee $O/prof_lbf64.txt
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 64 2048 2>&1 | tail -11 | tee $O/prof_rw64.txt
B="python bench.py --no-cpu-baseline --no-modes"
for a in "--steps 60 --warmup 5 --cadence env-only" "--steps 100 --warmup 5 --algo ia2c" "--steps 100 --warmup 5 --algo ia2c --hidden 128" "--steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64" "--steps 40 --warmup 5"; do
  timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('$a','->',round(d['value']/1e6,2),'M', round(d['ms_per_step'],3),'ms', {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
done 2>&1 | tee $O/rows.txt
