# wgrad row ranges sized for full rounds + the multi-tile shapes of the wide-critic parity test; the two rows
O=$GRAFT_REPO_ROOT/gpurun_out/r4S; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_ac_update.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-modes"
for a in "--steps 5 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128" "--steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128"; do
  timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print(d['metric'][25:],'->',round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms frac', round(d['roofline']['frac'],3), 'needed', round(d['roofline']['frac_needed'],3))"
done 2>&1 | tee $O/rows.txt
