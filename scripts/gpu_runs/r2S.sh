# GEMM-path standardise_returns: the std goldens through the wide learners + neighbours
mkdir -p gpurun_out/r2S
timeout 600 python -m pytest tests/test_gpu_standardise.py tests/test_gpu_layers.py tests/test_gpu_qmix.py -x -q -m gpu > gpurun_out/r2S/tests.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/r2S/tests.log | cut -c1-300
