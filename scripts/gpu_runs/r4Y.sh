# replicated pack sets for the collectors that read their packs from L2 every step: 4 copies (default build) vs 1 vs 8; parity first
O=$GRAFT_REPO_ROOT/gpurun_out/r4Y; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_collector_variants.py tests/test_ac_collector.py tests/test_gpu_rware.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-modes"
for v in "" "$R/codebase_amd/csrc/variants/libmarlhip_c1.so" "$R/codebase_amd/csrc/variants/libmarlhip_c8.so"; do
for a in "--steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 100 --warmup 5 --algo ia2c --hidden 128" "--steps 5 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128" "--steps 20 --warmup 3 --algo ia2c --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 4096 --hidden 128" "--steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64"; do
  MARLHIP_LIB=$v timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('lib=${v##*/}',d['metric'][25:],'->',round(d['value']/1e6,2),'M', round(d['ms_per_step'],3),'ms', {k[:14]:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
done; done 2>&1 | tee $O/rows.txt
