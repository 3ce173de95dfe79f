timeout 900 python -m pytest tests/test_gpu_qmix.py tests/test_gpu_layers.py tests/test_gru.py -q -m gpu 2>&1 | tail -12 | cut -c1-300
