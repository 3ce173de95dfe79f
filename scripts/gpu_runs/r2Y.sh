mkdir -p gpurun_out/r2Y
timeout 900 python -m pytest tests/test_gpu_layers.py -q -m gpu > gpurun_out/r2Y/tests.log 2>&1; echo "tests rc=$?"; tail -30 gpurun_out/r2Y/tests.log | cut -c1-250
