mkdir -p gpurun_out/r2S
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_fused_epilogue.py -x -q -m gpu > gpurun_out/r2S/new_tests.log 2>&1; echo "new tests rc=$?"; tail -5 gpurun_out/r2S/new_tests.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2S/gpu_tests.log 2>&1; echo "all tests rc=$?"; tail -5 gpurun_out/r2S/gpu_tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py > gpurun_out/r2S/bench_default.json 2> gpurun_out/r2S/bench_default.err; tail -1 gpurun_out/r2S/bench_default.json | cut -c1-400
timeout 300 python bench.py --hidden 128 > gpurun_out/r2S/bench_h128.json 2>/dev/null; tail -1 gpurun_out/r2S/bench_h128.json | cut -c1-300
timeout 200 python bench.py --algo vdn --steps 20 > gpurun_out/r2S/bench_vdn.json 2>/dev/null; tail -1 gpurun_out/r2S/bench_vdn.json | cut -c1-300
