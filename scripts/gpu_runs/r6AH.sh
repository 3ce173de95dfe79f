#!/bin/bash
# round 6, run AH: stacked GRU layers at the bench batch against the float64 port
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6AH"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 1500 python -m pytest tests/test_gru_stacked.py -m gpu -q -k "bench_batch" --durations=3 -s ) > $O/pytest.log 2>&1
grep "at-size" $O/pytest.log | cut -c1-260; tail -8 $O/pytest.log | cut -c1-300
