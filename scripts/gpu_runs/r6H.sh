#!/bin/bash
# round 6, run H: the tests touched since run E (at-size overlap with the actors' bound, QMIX wide golden, two-rank entry points with the step
# count through the exchange, bench --gpus 2), H128 row on the reverted tp kernels
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6H"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
timeout 1800 python -m pytest tests/test_gpu_two_ranks.py tests/test_bench_launch.py tests/test_gpu_qmix.py "tests/test_gpu_at_size_vs_oracle.py::test_config4_two_rounds_through_update_async_overlap_exactly_as_bench_drives_them" tests/test_gpu_ac_update.py -x -q -m gpu 2>&1 | tail -8
python bench.py --no-cpu-baseline --no-modes --steps 10 --warmup 2 --hidden 128 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('H128', round(d['value']/1e6,3), d['kernels'])"
