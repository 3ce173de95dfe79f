# wc_wgrad_kernel: 16 rows per LDS slice (default) vs 32 (variant ks32)
O=$GRAFT_REPO_ROOT/gpurun_out/r4AC; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
B="python $R/bench.py --no-cpu-baseline --no-modes"
MARLHIP_LIB=$R/codebase_amd/csrc/variants/libmarlhip_ks32.so timeout 600 python -m pytest tests/test_gpu_ac_update.py -m gpu -x -q -k wide 2>&1 | tail -2 | tee $O/pytest.log
for v in "" "$R/codebase_amd/csrc/variants/libmarlhip_ks32.so"; do
for a in "--steps 8 --warmup 2 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128" "--steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128"; do
  MARLHIP_LIB=$v timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('lib=${v##*/}',d['metric'][25:],'->',round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms frac', round(d['roofline']['frac'],3), round(d['kernels']['ac_update (fwd rows x3, elementwise, bwd rows x2)']['avg_us'],1))"
done; done 2>&1 | tee $O/rows.txt
