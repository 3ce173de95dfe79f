#!/bin/bash
# round 6, run J: the whole GPU suite + smoke + the default bench line (traffic quoted from profiles/r06_pmc_traffic.json) on the tree of commit e34b607
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6J"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 --durations=8 -rA -s ) > $O/pytest_gpu_full.log 2>&1
grep "at-size\|\[plan\]" $O/pytest_gpu_full.log | cut -c1-230 > $O/observed_deviations.txt
grep -v "^PASSED\|at-size\|^\[" $O/pytest_gpu_full.log | tail -22
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 900 python $R/bench.py > $O/bench_default_line.json 2> $O/bench_default_line.err ) 2>&1 | tail -3
python - <<'PY'
import json, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6J"
d=json.loads([l for l in open(O+"/bench_default_line.json") if l.startswith("{")][-1])
print(round(d["value"]/1e6,2), d["roofline"]["frac"], d["roofline"]["traffic"], (d["roofline"]["traffic_source"] or {}).get("file"))
for k,v in d["modes"].items():
    print(k[:70], "|", v.get("error") or (round(v["value"]/1e6,2), (v["roofline"] or {}).get("traffic"), (v["roofline"] or {}).get("traffic_source")))
PY
