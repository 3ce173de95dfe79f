timeout 900 python -m pytest tests/test_gpu_optimizers.py tests/test_gpu_parity.py tests/test_gpu_ac_update.py tests/test_gpu_qmix.py -q -m gpu 2>&1 | tail -15 | cut -c1-250
