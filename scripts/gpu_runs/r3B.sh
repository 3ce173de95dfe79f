# round 3, second GPU call: full suite again (two-rank worker fixed, 6-layer nets, other recurrent widths), forced-dist profile with the
# vectorised norm, GRU-128 weight-gradient prefetch A/B (libmarlhip_pf.so), bf16 MFMA micro-benchmark
O=$GRAFT_REPO_ROOT/gpurun_out/r3B; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
PF=$R/codebase_amd/csrc/variants/libmarlhip_pf.so
MARLHIP_LIB=$PF timeout 600 python -m pytest tests/test_gru.py -x -q -m gpu > $O/tests_pf.log 2>&1; echo "pf tests rc=$?"; tail -2 $O/tests_pf.log | cut -c1-200
for lib in default pf; do
  if [ $lib = pf ]; then export MARLHIP_LIB=$PF; else unset MARLHIP_LIB; fi
  timeout 300 $B --steps 5 --warmup 1 --rnn --hidden 128 > $O/gru128_$lib.json 2>/dev/null
  timeout 300 $B --steps 10 --warmup 2 --rnn --hidden 128 --algo ia2c > $O/gru128_ia2c_$lib.json 2>/dev/null
done
unset MARLHIP_LIB
timeout 200 $B --steps 20 --warmup 3 > $O/bench_h64.json 2>/dev/null
MARLHIP_BENCH_FORCE_DIST=1 timeout 200 $B --steps 20 --warmup 3 > $O/bench_forcedist.json 2>/dev/null
MARLHIP_BENCH_FORCE_DIST=1 timeout 200 $B --steps 10 --warmup 2 --hidden 128 > $O/bench_forcedist_h128.json 2>/dev/null
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3B"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "ms %.3f"%d["ms_per_step"], "frac %.3f"%(r.get("frac") or 0), "us %.0f"%(r.get("avg_launch_us") or 0))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
cd /tmp; export TMPDIR=/tmp
MARLHIP_BENCH_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_forcedist --output-format csv -- $B --steps 10 --warmup 2 > $O/stats_forcedist.log 2>&1
for lib in default pf; do
  if [ $lib = pf ]; then export MARLHIP_LIB=$PF; else unset MARLHIP_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_gru128_$lib --output-format csv -- $B --steps 3 --warmup 1 --rnn --hidden 128 > $O/stats_gru128_$lib.log 2>&1
done
unset MARLHIP_LIB
cd $R
./scripts/_bin/mfma_ubench5 > $O/mfma_ubench5.txt 2>&1; cat $O/mfma_ubench5.txt
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete; du -sh $O
