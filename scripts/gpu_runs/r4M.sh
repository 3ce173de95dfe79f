# GEMM path A/B: prefetch depth 2 (default build) vs 1 (variant pf1), both with the batched epilogue loads
O=$GRAFT_REPO_ROOT/gpurun_out/r4M; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
B="python bench.py --no-cpu-baseline --no-modes"
for v in "" "$R/codebase_amd/csrc/variants/libmarlhip_pf1.so"; do
for a in "--steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128" "--steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 10 --warmup 2 --hidden 256" "--steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192 --hidden 128"; do
  MARLHIP_LIB=$v timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('lib=${v##*/}',d['metric'][25:],'->',round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms frac', round(d['roofline']['frac'],3))"
done; done 2>&1 | tee $O/rows.txt
