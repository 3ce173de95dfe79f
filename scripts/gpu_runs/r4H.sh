# full -m gpu suite + smoke + default bench line on the current tree
O=$GRAFT_REPO_ROOT/gpurun_out/r4H; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json;d=json.load(open('$O/bench_default.json'));print(d['value'],d['roofline']['frac'],d['roofline']['traffic'],d['config']['lr'],{k[:24]:round(v['value']/1e6,2) for k,v in d['modes'].items()},d['cpu_baseline']['value'])"
