# two waves per agent also when the output layer's operands ride in registers (2 agents, hidden 128): parity + rows with A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r4Z; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_collector_variants.py tests/test_ac_collector.py tests/test_gpu_parity.py tests/test_gpu_host_api.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-modes"
for e in "X=1" "MARLHIP_COL_HS=1 MARLHIP_ACOL_HS=1"; do
for a in "--steps 100 --warmup 5 --algo ia2c --hidden 128" "--steps 50 --warmup 5 --algo ippo --hidden 128" "--steps 100 --warmup 5 --algo maa2c --hidden 128" "--steps 30 --warmup 3 --hidden 128" "--steps 100 --warmup 5 --algo ia2c" "--steps 60 --warmup 5 --cadence env-only"; do
  env $e timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('$e',d['metric'][25:],'$a','->',round(d['value']/1e6,2),'M', round(d['ms_per_step'],3),'ms', {k[:14]:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
done; done 2>&1 | tee $O/rows.txt
