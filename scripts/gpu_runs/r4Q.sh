# one wave per agent for 8 LBF agents (512-thread workgroups): collector parity, region counters, the rows that use it
O=$GRAFT_REPO_ROOT/gpurun_out/r4Q; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_collector_variants.py tests/test_ac_collector.py tests/test_gpu_parity.py tests/test_gpu_qmix.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
V=$R/codebase_amd/csrc/variants/libmarlhip_acolprof.so
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 128 4096 lbforaging:Foraging-15x15-8p-5f-v3 2>&1 | tail -11 | tee $O/prof_lbf8p_128.txt
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 64 4096 lbforaging:Foraging-15x15-8p-5f-v3 2>&1 | tail -11 | tee $O/prof_lbf8p_64.txt
B="python $R/bench.py --no-cpu-baseline --no-modes"
for a in "--steps 5 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128" "--steps 5 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 64" "--steps 5 --warmup 1 --algo ia2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128" "--steps 5 --warmup 2 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128" "--steps 5 --warmup 2 --algo idqn --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128" "--steps 5 --warmup 2 --algo idqn --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 64"; do
  timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print(d['metric'][25:],'H',d['config'].get('hidden'),'->',round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms frac', round(d['roofline']['frac'],3), {k[:24]:round(v['avg_us'],1) for k,v in d.get('kernels',{}).items()})"
done 2>&1 | tee $O/rows.txt
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 128 2048 2>&1 | tail -11 | tee $O/prof_rw128.txt
for a in "--steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 40 --warmup 5 --algo ia2c --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 4096 --hidden 128" "--steps 20 --warmup 3 --hidden 128 --env-name lbforaging:Foraging-15x15-4p-5f-v3"; do
  timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print(d['metric'][25:],'->',round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms frac', round(d['roofline']['frac'],3), {k[:24]:round(v['avg_us'],1) for k,v in d.get('kernels',{}).items()})"
done 2>&1 | tee $O/rows2.txt
