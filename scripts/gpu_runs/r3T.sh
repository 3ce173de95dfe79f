# round 3, final-tree check: full -m gpu suite, smoke(), the evidence collection (profiles/r03_*), the default bench line
O=$GRAFT_REPO_ROOT/gpurun_out/r3T; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-300
bash scripts/collect_profiles_r03.sh > $O/collect.log 2>&1; tail -2 $O/collect.log
cd $R; timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 600 $O/bench_default.json | cut -c1-600
