# tp_bwd_kernel on 71-wide rows: two row blocks per step (42 - 102 spilled registers) vs one (default)
O=$GRAFT_REPO_ROOT/gpurun_out/r4AA; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
B="python $R/bench.py --no-cpu-baseline --no-modes"
MARLHIP_LIB=$R/codebase_amd/csrc/variants/libmarlhip_nb2.so timeout 600 python -m pytest tests/test_gpu_ac_update.py tests/test_gpu_rware.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
for v in "" "$R/codebase_amd/csrc/variants/libmarlhip_nb2.so"; do
for a in "--steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128"; do
  MARLHIP_LIB=$v timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('lib=${v##*/}',d['metric'][25:],'->',round(d['value']/1e6,2),'M', round(d['ms_per_step'],3),'ms', {k[:14]:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
done; done 2>&1 | tee $O/rows.txt
