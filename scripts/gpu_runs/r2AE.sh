mkdir -p gpurun_out/r2AE
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2AE/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r2AE/tests.log | cut -c1-200
timeout 300 python bench.py > gpurun_out/r2AE/bench_default.json 2>/dev/null; tail -1 gpurun_out/r2AE/bench_default.json | cut -c1-300
