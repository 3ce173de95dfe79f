# round 5, call E: the QMIX weight-gradient kernel's instruction diet (two blocks per iteration, incremental block positions, no selects) - parity, then timing
O=$GRAFT_REPO_ROOT/gpurun_out/r5E; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 900 python -m pytest tests/test_gpu_qmix.py tests/test_gpu_rware.py "tests/test_gpu_at_size_vs_oracle.py::test_config5_qmix_15x15_8p5f_H128_B8192_through_the_trainer_vs_oracle_port" -m gpu -q --maxfail=6 ) 2>&1 | tail -25 | tee $O/pytest.log
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-modes --no-kernel-timing"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats8p --output-format csv -- $B --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128 --steps 2 --warmup 1 > $O/stats8p.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats4p --output-format csv -- $B --algo qmix --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192 --steps 3 --warmup 1 > $O/stats4p.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/statsrw --output-format csv -- $B --algo qmix --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --steps 2 --warmup 1 > $O/statsrw.log 2>&1
for f in $(find $O -name "*kernel_stats.csv"); do echo "== $f"; grep -i "qmix\|Name" $f | cut -c1-70,150-400 | sed 's/([^"]*)"/"/' | awk -F',' '{print $1, $(NF-6), $(NF-4)}' | head -8; done
grep '^{' $O/stats8p.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('8p',round(d['value']/1e6,3),'M',round(d['ms_per_step'],2),'ms')"
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
