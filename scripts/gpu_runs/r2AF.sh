timeout 900 python -m pytest tests/test_gpu_collector_variants.py -x -q -m gpu 2>&1 | tail -15 | cut -c1-300
