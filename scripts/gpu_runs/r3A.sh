# round 3, first GPU call: full suite on the new tree (incl. bench-path-vs-oracle and the torchrun entry-point tests), smoke, default
# bench line (with `modes` + reference cpu baseline), hidden-128 before/after evidence, A/B of the two OFF-by-default macros
# (libmarlhip_flat.so = -DMARLHIP_WIDE_FLATLOAD=1 -DMARLHIP_GRU_WGRAD_FLAT=1), forced-dist single-rank profile, HBM micro-benchmark
O=$GRAFT_REPO_ROOT/gpurun_out/r3A; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -1 $O/bench_default.json | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
timeout 200 $B --hidden 128 --steps 20 --warmup 3 > $O/bench_h128.json 2>/dev/null; cut -c1-200 $O/bench_h128.json
FLAT=$R/codebase_amd/csrc/variants/libmarlhip_flat.so
# the flat-load variant must still be right: the tests that touch the GEMM path and the recurrent learner
MARLHIP_LIB=$FLAT timeout 900 python -m pytest tests/test_gru.py tests/test_gpu_layers.py tests/test_gpu_ac_update.py tests/test_gpu_standardise.py -x -q -m gpu > $O/tests_flat.log 2>&1; echo "flat tests rc=$?"; tail -2 $O/tests_flat.log | cut -c1-200
for lib in default flat; do
  if [ $lib = flat ]; then export MARLHIP_LIB=$FLAT; else unset MARLHIP_LIB; fi
  timeout 300 $B --steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128 > $O/maa2c8p_$lib.json 2>/dev/null
  timeout 300 $B --steps 3 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/mappo_rware_$lib.json 2>/dev/null
  timeout 300 $B --steps 5 --warmup 1 --rnn --hidden 128 > $O/gru128_$lib.json 2>/dev/null
  timeout 300 $B --steps 10 --warmup 2 --rnn > $O/gru64_$lib.json 2>/dev/null
  timeout 300 $B --steps 10 --warmup 2 --algo idqn --hidden 256 > $O/idqn_h256_$lib.json 2>/dev/null
done
unset MARLHIP_LIB
python - <<'PY'
import json, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3A"
for f in sorted(glob.glob(O+"/*_default.json")+glob.glob(O+"/*_flat.json")+[O+"/bench_h128.json"]):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline") or {}
        print(os.path.basename(f), "%.3f M"%(d["value"]/1e6), "frac %.3f"%(r.get("frac") or 0), "us %.0f"%(r.get("avg_launch_us") or 0))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_h64 --output-format csv -- $B --steps 20 --warmup 3 > $O/stats_h64.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_h128 --output-format csv -- $B --steps 10 --warmup 2 --hidden 128 > $O/stats_h128.log 2>&1
for lib in default flat; do
  if [ $lib = flat ]; then export MARLHIP_LIB=$FLAT; else unset MARLHIP_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_gru128_$lib --output-format csv -- $B --steps 3 --warmup 1 --rnn --hidden 128 > $O/stats_gru128_$lib.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_maa2c8p_$lib --output-format csv -- $B --steps 2 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128 > $O/stats_maa2c8p_$lib.log 2>&1
done
unset MARLHIP_LIB
# the N > 1 code path with one rank (RCCL all-reduce of one rank per update): kernels per update, fill / copy calls
MARLHIP_BENCH_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/stats_forcedist --output-format csv -- $B --steps 10 --warmup 2 > $O/stats_forcedist.log 2>&1
MARLHIP_BENCH_FORCE_DIST=1 timeout 200 $B --steps 20 --warmup 3 > $O/bench_forcedist.json 2>/dev/null; cut -c1-200 $O/bench_forcedist.json
cd $R
timeout 300 python scripts/ubench_hbm.py > $O/hbm_ubench.txt 2>&1; tail -1 $O/hbm_ubench.txt | cut -c1-1500
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete; du -sh $O
