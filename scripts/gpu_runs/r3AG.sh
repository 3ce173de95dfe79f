# round 3: the headline cadence (U = 32 updates of B = 4096 per 4096 episodes, lr 3e-3, Polyak 0.1) still learns on the final tree:
# three seeds x 270 M env-steps, return at 10 .. 90 % (round 2: 0.63 / 0.87 / 0.88 at the end, 12.7 - 14.5 s of wall-clock)
O=$GRAFT_REPO_ROOT/gpurun_out/r3AG; mkdir -p $O; cd $GRAFT_REPO_ROOT
for s in 0 1 2; do
  timeout 300 python tests/tools/learning_parity.py vec $s 2.7e8 4096 32 4096 algorithm.lr=3e-3 algorithm.target_update_interval_or_tau=0.1 2>/dev/null | grep '^{' > $O/ratio_seed$s.jsonl
  python - $O/ratio_seed$s.jsonl <<'PY'
import json,sys
r=[json.loads(l) for l in open(sys.argv[1])]
print("seed", r[0]["seed"], " ".join("%.2f"%x["mean_return"] for x in r), "wall %.1fs"%r[-1]["wall_s"], "steps %.0fM"%(r[-1]["env_steps"]/1e6))
PY
done
