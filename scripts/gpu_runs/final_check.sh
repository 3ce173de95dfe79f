mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/final/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/final/tests.log | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/collect_profiles.sh > gpurun_out/final/collect.log 2>&1; wc -l gpurun_out/prof2/matrix.jsonl
timeout 300 python bench.py > gpurun_out/final/bench_default.json 2>/dev/null; tail -1 gpurun_out/final/bench_default.json | cut -c1-260
