#!/bin/bash
# round 6, run AG: the whole GPU suite twice in a row on the final tree (flakiness check) + smoke
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6AG"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
for k in 1 2; do
  ( time timeout 2700 python -m pytest tests -m gpu -q --maxfail=10 --durations=5 -rA -s ) > $O/pytest_gpu_full_$k.log 2>&1
  echo "pass $k: $(grep -E '^[0-9]+ (passed|failed)|passed|failed' $O/pytest_gpu_full_$k.log | tail -1) ; retries $(grep -c 'second attempt' $O/pytest_gpu_full_$k.log)"
done
grep "at-size\|\[plan\]" $O/pytest_gpu_full_2.log | cut -c1-230 > $O/observed_deviations.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
