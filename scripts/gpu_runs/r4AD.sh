# tp_bwd_kernel with BOTH hidden layers read back from the forward-rows pass (STORED1; two row blocks per step up to 80-wide rows): parity + rows
O=$GRAFT_REPO_ROOT/gpurun_out/r4AD; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_gpu_ac_update.py tests/test_gpu_standardise.py tests/test_gpu_sharing.py tests/test_gpu_two_ranks.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-modes"
for a in "--steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 100 --warmup 5 --algo ia2c --hidden 128" "--steps 50 --warmup 5 --algo ippo --hidden 128" "--steps 8 --warmup 2 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128" "--steps 20 --warmup 3 --algo ia2c --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 16384 --hidden 128" "--steps 3 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 16384 --hidden 128"; do
  timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print(d['metric'][25:],'->',round(d['value']/1e6,2),'M', round(d['ms_per_step'],3),'ms frac', round(d['roofline']['frac'],3), {k[:14]:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
done 2>&1 | tee $O/rows.txt
