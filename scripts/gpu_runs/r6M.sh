#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-modes --steps 2 --warmup 1 --hidden 128 2>/dev/null | grep "TPPROF" | tail -8
python bench.py --no-cpu-baseline --no-modes --steps 2 --warmup 1 --hidden 128 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('H128', round(d['value']/1e6,3), d['kernels'])"
