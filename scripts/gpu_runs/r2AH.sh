timeout 900 python -m pytest tests/test_gpu_standardise.py -q -m gpu 2>&1 | grep -v "^$" | tail -45 | cut -c1-300
