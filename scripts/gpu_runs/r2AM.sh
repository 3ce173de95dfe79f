timeout 900 python -m pytest tests/test_ac_collector.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -25 | cut -c1-250
