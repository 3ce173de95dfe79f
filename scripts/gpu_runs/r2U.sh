mkdir -p gpurun_out/r2U
timeout 900 python -m pytest tests/test_gpu_ac_update.py tests/test_gpu_layers.py -q -m gpu > gpurun_out/r2U/ac_tests.log 2>&1; echo "ac tests rc=$?"; tail -15 gpurun_out/r2U/ac_tests.log
