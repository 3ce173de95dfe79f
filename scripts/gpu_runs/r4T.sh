# k-major LDS tiles for the generic weight-gradient GEMMs + per-round step counting of the AC bench loop without a reduction
O=$GRAFT_REPO_ROOT/gpurun_out/r4T; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_ac_update.py tests/test_gru.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-modes"
for a in "--steps 10 --warmup 2 --hidden 256" "--steps 100 --warmup 5 --algo ia2c" "--steps 100 --warmup 5 --algo ia2c --hidden 128" "--steps 100 --warmup 5 --algo maa2c --hidden 128" "--steps 5 --warmup 1 --algo ia2c --hidden 256"; do
  timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('$a',d['metric'][25:],'->',round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms frac', round(d['roofline']['frac'],3))"
done 2>&1 | tee $O/rows.txt
