# stored-form threshold 96 (default build) vs 80: MAA2C with an 84-wide centralised critic (4 agents x 21), two row blocks per step vs one
O=$GRAFT_REPO_ROOT/gpurun_out/r4AF; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
B="python $R/bench.py --no-cpu-baseline --no-modes"
for v in "" "$R/codebase_amd/csrc/variants/libmarlhip_nb80.so"; do
  MARLHIP_LIB=$v timeout 200 $B --steps 20 --warmup 3 --algo maa2c --env-name lbforaging:Foraging-10x10-4p-3f-v3 --envs 8192 --hidden 128 2>$O/err.txt | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('lib=${v##*/}',d['metric'][25:],'->',round(d['value']/1e6,2),'M', round(d['ms_per_step'],3),'ms frac', round(d['roofline']['frac'],3), {k[:14]:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
  tail -2 $O/err.txt | cut -c1-200
done 2>&1 | tee $O/rows.txt
