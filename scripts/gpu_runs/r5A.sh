# round 5, call A: the at-size oracle tests (configs 3/4/5 + wide critics), the p2p changes on two ranks of one GPU, the new default bench line
# (modes: trained policy, configs 3-5), the HBM micro-benchmarks on the final round-4 tree
O=$GRAFT_REPO_ROOT/gpurun_out/r5A; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 900 python -m pytest tests/test_gpu_at_size_vs_oracle.py -m gpu -q --durations=10 ) 2>&1 | tail -40 | tee $O/pytest_at_size.log
( time timeout 600 python -m pytest tests/test_gpu_two_ranks.py tests/test_abi.py tests/test_gpu_rware.py -m gpu -x -q ) 2>&1 | tail -15 | tee $O/pytest_p2p.log
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -5 $O/bench_default.err
timeout 300 python scripts/ubench_hbm.py > $O/hbm_ubench.json 2> $O/hbm_ubench.err; tail -3 $O/hbm_ubench.err
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r5A"
try:
    d=json.loads([l for l in open(O+"/bench_default.json") if l.startswith("{")][-1])
    print("HEADLINE", round(d["value"]/1e6,2), "M", d["ms_per_step"], d["roofline"]["frac"], d["config"].get("mean_episode_length"))
    for k,v in d["modes"].items():
        print("  MODE", k[:90], "->", v.get("error") or (round(v["value"]/1e6,3), "M", round(v["ms_per_step"],3), "ms", "len", v.get("mean_episode_length"), "ret", v.get("mean_episode_return_last_round"), "frac", (v.get("roofline") or {}).get("frac")))
    print("  CPU", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
except Exception as e: print("bench parse failed", e)
try:
    u=json.load(open(O+"/hbm_ubench.json"))
    for k,v in u.items(): print("  UBENCH", k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("us","achieved_GBs","frac_of_8TBs")})
except Exception as e: print("ubench parse failed", e)
PY
