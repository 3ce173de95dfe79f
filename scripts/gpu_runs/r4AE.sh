# DQN family at hidden 128: pass B reads both hidden layers back from pass F (two row blocks per step up to 80-wide rows): parity + rows
O=$GRAFT_REPO_ROOT/gpurun_out/r4AE; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_path_vs_oracle.py tests/test_gpu_qmix.py tests/test_gpu_standardise.py tests/test_gpu_sharing.py tests/test_gpu_two_ranks.py tests/test_gpu_fused_epilogue.py tests/test_gpu_optimizers.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-modes"
for a in "--steps 30 --warmup 3 --hidden 128" "--steps 6 --warmup 2 --algo vdn --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192 --hidden 128" "--steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128" "--steps 3 --warmup 1 --algo idqn --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 5 --warmup 1 --algo qmix --hidden 128"; do
  timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print(d['metric'][25:],'->',round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms frac', round(d['roofline']['frac'],3), {k[:14]:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
done 2>&1 | tee $O/rows.txt
