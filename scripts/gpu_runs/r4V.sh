# two waves per agent in the collectors (HS = 2; 2 agents, LDS-resident packs, <= 4096 envs): parity + rows, with the one-wave form for A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r4V; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_collector_variants.py tests/test_ac_collector.py tests/test_gpu_parity.py tests/test_gpu_host_api.py tests/test_gpu_bench_path_vs_oracle.py -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-modes"
for e in "X=1" "MARLHIP_COL_HS=1 MARLHIP_ACOL_HS=1"; do
for a in "--steps 60 --warmup 5 --cadence env-only" "--steps 100 --warmup 5 --algo ia2c" "--steps 100 --warmup 5 --algo ia2c --hidden 128" "--steps 40 --warmup 5"; do
  env $e timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('$e','$a','->',round(d['value']/1e6,2),'M', round(d['ms_per_step'],3),'ms', {k[:14]:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
done; done 2>&1 | tee $O/rows.txt
