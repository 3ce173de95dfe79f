# round 5, call H: the QMIX rows and kernel stats again on the final tree (LDS-staged first-layer and weight-gradient kernels), smoke(), the default line
O=$GRAFT_REPO_ROOT/gpurun_out/r5H; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-modes"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_qmix8p --output-format csv -- $B --steps 2 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128 > $O/stats_qmix8p.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_qmixrw --output-format csv -- $B --steps 2 --warmup 1 --algo qmix --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64 > $O/stats_qmixrw.log 2>&1
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
: > $O/matrix_qmix.jsonl
run() { timeout 400 $B "$@" 2>/dev/null | grep '^{' >> $O/matrix_qmix.jsonl; }
run --steps 20 --warmup 3 --algo qmix
run --steps 6 --warmup 1 --algo qmix --env-name lbforaging:Foraging-10x10-3p-3f-v3 --envs 8192
run --steps 6 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192
run --steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128
run --steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128 --mixer-fp16
run --steps 3 --warmup 1 --algo qmix --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64
run --steps 60 --warmup 5 --cadence env-only
( timeout 600 python $R/bench.py > $O/bench_default_line.json 2> $O/bench_default_line.err )
python - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5H"
for l in open(O+"/matrix_qmix.jsonl"):
    d=json.loads(l); print(round(d['value']/1e6,3),'M',round(d['ms_per_step'],2),'ms',d['config']['workload'][:70], {k[:12]:round(v['avg_us'],1) for k,v in d['kernels'].items()}, round((d['roofline'] or {}).get('frac',0),3))
d=json.loads([l for l in open(O+"/bench_default_line.json") if l.startswith("{")][-1])
print("HEADLINE", round(d["value"]/1e6,2), d["roofline"]["frac"], d["roofline"]["traffic"])
for k,v in d["modes"].items(): print("  ", k[:70], v.get("error") or round(v["value"]/1e6,3))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
