# direct-operand GEMM (no LDS, no barriers) against the staged kernels on the GEMM-path rows: parity tests under the switch, then the rows
O=$GRAFT_REPO_ROOT/gpurun_out/r4J; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
MARLHIP_WIDE_DIRECT=13 timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_ac_update.py tests/test_gpu_qmix.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest13.log
MARLHIP_WIDE_DIRECT=12 timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_ac_update.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest12.log
B="python bench.py --no-cpu-baseline --no-modes"
for d in 0 2 3 12 13; do
 for a in "--steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128" "--steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 10 --warmup 2 --hidden 256"; do
  MARLHIP_WIDE_DIRECT=$d timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('direct=$d',d['metric'][25:],'->',round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms frac', round(d['roofline']['frac'],3))"
 done
done 2>&1 | tee $O/rows.txt
