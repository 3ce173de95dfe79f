# A/B of the k-major weight-gradient tiles in wide_gemm128_kernel on shapes that reach it (hidden 256 at >= 8192 envs), and the hidden-256 IA2C row
O=$GRAFT_REPO_ROOT/gpurun_out/r4U; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
B="python $R/bench.py --no-cpu-baseline --no-modes"
for v in "" "$R/codebase_amd/csrc/variants/libmarlhip_km0.so"; do
for a in "--steps 4 --warmup 1 --hidden 256 --envs 16384 --update-batch 16384" "--steps 5 --warmup 1 --algo ia2c --hidden 256 --envs 16384"; do
  MARLHIP_LIB=$v timeout 300 $B $a 2>$O/err.txt | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('lib=${v##*/}','$a',d['metric'][25:],'->',round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms frac', round(d['roofline']['frac'],3))"
  tail -3 $O/err.txt
done; done 2>&1 | tee $O/rows.txt
