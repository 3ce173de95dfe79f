mkdir -p gpurun_out/r2W
timeout 600 python -m pytest tests/test_gpu_fused_epilogue.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r2W/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r2W/tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2W/bench_coop.json 2>/dev/null; tail -1 gpurun_out/r2W/bench_coop.json | cut -c1-230
MARLHIP_TWO_LAUNCH_EPILOGUE=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2W/bench_nocoop.json 2>/dev/null; tail -1 gpurun_out/r2W/bench_nocoop.json | cut -c1-230
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2W/bench_coop2.json 2>/dev/null; tail -1 gpurun_out/r2W/bench_coop2.json | cut -c1-230
timeout 300 python bench.py --no-cpu-baseline --cadence reference --steps 4 --warmup 1 2>/dev/null | tail -1 | cut -c1-230
