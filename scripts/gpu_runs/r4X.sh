# prefetch depth of the L2-pack forward pass (mlp_forward_g / mlp_forward_g_h2): 3 (default build) vs 4 vs 5 groups in flight
O=$GRAFT_REPO_ROOT/gpurun_out/r4X; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
B="python $R/bench.py --no-cpu-baseline --no-modes"
for v in "" "$R/codebase_amd/csrc/variants/libmarlhip_d4.so" "$R/codebase_amd/csrc/variants/libmarlhip_d5.so"; do
for a in "--steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 100 --warmup 5 --algo ia2c --hidden 128" "--steps 5 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128" "--steps 20 --warmup 3 --algo ia2c --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 4096 --hidden 128"; do
  MARLHIP_LIB=$v timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('lib=${v##*/}',d['metric'][25:],'->',round(d['value']/1e6,2),'M', round(d['ms_per_step'],3),'ms', {k[:14]:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
done; done 2>&1 | tee $O/rows.txt
MARLHIP_ACOL_HS=1 timeout 300 $B --steps 100 --warmup 5 --algo ia2c --hidden 128 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('HS off',d['metric'][25:],'->',round(d['value']/1e6,2),'M', {k[:14]:round(v['avg_us'],1) for k,v in d['kernels'].items()})" | tee -a $O/rows.txt
