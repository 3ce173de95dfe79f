# wc_bwd with W2^T resident in LDS (default build) vs streamed per tile (variant res0): parity, rows, per-launch times
O=$GRAFT_REPO_ROOT/gpurun_out/r4AB; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_ac_update.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-modes"
for v in "" "$R/codebase_amd/csrc/variants/libmarlhip_res0.so"; do
for a in "--steps 5 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128" "--steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 5 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 64"; do
  MARLHIP_LIB=$v timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('lib=${v##*/}',d['metric'][25:],'->',round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms frac', round(d['roofline']['frac'],3))"
done; done 2>&1 | tee $O/rows.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/tr_maa2c --output-format csv -- $B --steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128 > $O/maa2c8p.log 2>&1
cd $R; python scripts/trace_by_grid.py $O tr_maa2c | head -12
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
