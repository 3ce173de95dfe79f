#!/bin/bash
# round 6, run AJ: the whole GPU suite + smoke + the default bench line on the final tree (after the reduce kernel's change)
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6AJ"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 2700 python -m pytest tests -m gpu -q --maxfail=10 --durations=5 -rA -s ) > $O/pytest_gpu_full.log 2>&1
echo "suite: $(grep -E 'passed|failed' $O/pytest_gpu_full.log | tail -1) ; retries $(grep -c 'second attempt' $O/pytest_gpu_full.log)"
grep "at-size\|\[plan\]" $O/pytest_gpu_full.log | cut -c1-230 > $O/observed_deviations.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default_line.err ); python -c "
import json; d=json.loads(open('$O/bench_default_line.json').read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['cpu_baseline']['value'])"
