# round 5, call G: the whole GPU suite on the final tree (LDS-staged mixer weight gradients), the QMIX rows of the matrix again, the default line
O=$GRAFT_REPO_ROOT/gpurun_out/r5G; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 ) 2>&1 | tail -15 | tee $O/pytest_gpu.log
B="python $R/bench.py --no-cpu-baseline --no-modes"
: > $O/matrix_qmix.jsonl
run() { timeout 400 $B "$@" 2>/dev/null | grep '^{' >> $O/matrix_qmix.jsonl; }
run --steps 20 --warmup 3 --algo qmix
run --steps 6 --warmup 1 --algo qmix --env-name lbforaging:Foraging-10x10-3p-3f-v3 --envs 8192
run --steps 6 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192
run --steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128
run --steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128 --mixer-fp16
run --steps 3 --warmup 1 --algo qmix --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64
run --steps 10 --warmup 2 --rnn --algo qmix
python - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5G/matrix_qmix.jsonl"):
    d=json.loads(l); print(round(d['value']/1e6,3),'M',round(d['ms_per_step'],2),'ms',d['config']['workload'][:80], {k[:12]:round(v['avg_us'],1) for k,v in d['kernels'].items()})
PY
( timeout 600 python $R/bench.py > $O/bench_default_line.json 2> $O/bench_default_line.err ); python - <<'PY'
import json,os
d=json.loads([l for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5G/bench_default_line.json") if l.startswith("{")][-1])
print("HEADLINE", round(d["value"]/1e6,2), d["roofline"]["frac"], d["roofline"]["traffic"], d["config"]["mean_episode_length"])
for k,v in d["modes"].items(): print("  ", k[:70], v.get("error") or round(v["value"]/1e6,3))
PY
