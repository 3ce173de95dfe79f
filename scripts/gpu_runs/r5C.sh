# round 5, call C: the optimiser step inside the learner launch's prologue (UpdPro) - parity (bitwise vs the three-launch form, the oracle
# comparisons of the bench path), then the rows it moves, A/B against MARLHIP_NO_PROLOGUE_ADAM=1, kernel stats of the reference cadence
O=$GRAFT_REPO_ROOT/gpurun_out/r5C; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 1200 python -m pytest tests/test_gpu_fused_epilogue.py tests/test_gpu_bench_path_vs_oracle.py tests/test_gpu_parity.py tests/test_gpu_sharing.py tests/test_gpu_split16.py tests/test_gpu_two_ranks.py -m gpu -q --maxfail=8 --durations=8 ) 2>&1 | tail -60 | tee $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-modes"
for e in "X=1" "MARLHIP_NO_PROLOGUE_ADAM=1"; do
for a in "--cadence reference --steps 3 --warmup 1" "--steps 60 --warmup 5" "--steps 60 --warmup 5 --hparams tuned" "--steps 20 --warmup 3 --updates-per-round 128 --update-batch 4096" "--cadence reference --steps 3 --warmup 1 --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 2048"; do
  env $e timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('$e','$a','->',round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms', {k[:14]:round(v['avg_us'],2) for k,v in d['kernels'].items()}, 'frac', round((d.get('roofline') or {}).get('frac',0),4))"
done; done 2>&1 | tee $O/rows.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_ref --output-format csv -- $B --cadence reference --steps 2 --warmup 1 --no-kernel-timing > $O/stats_ref.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_ratio --output-format csv -- $B --steps 20 --warmup 3 --no-kernel-timing > $O/stats_ratio.log 2>&1
for f in $(find $O/stats_ref $O/stats_ratio -name "*kernel_stats.csv"); do echo "== $f"; head -6 $f | cut -c1-60,200-330; done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
