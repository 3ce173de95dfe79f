mkdir -p gpurun_out/r2AJ
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2AJ/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r2AJ/tests.log | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
V="python tests/tools/learning_parity.py vec"
timeout 300 $V 0 2.7e8 4096 32 4096 algorithm.lr=3e-3 algorithm.target_update_interval_or_tau=0.1 2>/dev/null | grep '^{' > gpurun_out/r2AJ/learn_ratio.jsonl; tail -2 gpurun_out/r2AJ/learn_ratio.jsonl | cut -c1-220
timeout 300 $V 1 2.7e7 64 64 32 2>/dev/null | grep '^{' > gpurun_out/r2AJ/learn_ref.jsonl; tail -1 gpurun_out/r2AJ/learn_ref.jsonl | cut -c1-220
timeout 300 python bench.py > gpurun_out/r2AJ/bench_default.json 2>/dev/null; tail -1 gpurun_out/r2AJ/bench_default.json | cut -c1-1500
