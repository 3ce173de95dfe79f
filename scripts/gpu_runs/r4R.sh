# fused wide critics (wide_critic.h): parity tests, then the rows and their per-launch breakdown
O=$GRAFT_REPO_ROOT/gpurun_out/r4R; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_ac_update.py -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-modes"
for a in "--steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128" "--steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 64"; do
  timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print(d['metric'][25:],'->',round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms frac', round(d['roofline']['frac'],3))"
done 2>&1 | tee $O/rows.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/tr_maa2c --output-format csv -- $B --steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128 > $O/maa2c8p.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $O/tr_mappo --output-format csv -- $B --steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/mappo.log 2>&1
cd $R; python scripts/trace_by_grid.py $O tr_maa2c tr_mappo
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
