"""Learning curves of IDQN / VDN / QMIX on the device path (cooperative Level-Based Foraging):
    python scripts/dqn_family_curves.py [total_steps] [envs] [env name] [extra overrides, comma separated] [algorithms]
prints mean evaluation return per checkpoint for each algorithm (eval at epsilon 0.05, 512 episodes)."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codebase_amd import run  # noqa: E402

steps = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
name = sys.argv[3] if len(sys.argv) > 3 else "lbforaging:Foraging-8x8-2p-3f-v3"
extra = sys.argv[4].split(",") if len(sys.argv) > 4 else []           # e.g. algorithm.model.use_rnn=True
algos = sys.argv[5].split(",") if len(sys.argv) > 5 else ("idqn", "vdn", "qmix")
for algo in algos:
    os.environ["MARLHIP_RUN_DIR"] = tempfile.mkdtemp()
    t0 = time.time()
    df = run.main([f"+algorithm={algo}", f"env.name={name}", "env.time_limit=25", f"env.parallel_envs={envs}",
                   "algorithm.model.layers=[64,64]", "seed=0", f"algorithm.total_steps={steps}", f"algorithm.eval_interval={steps // 10}",
                   "algorithm.eval_episodes=512", "algorithm.updates_per_round=64", f"algorithm.update_batch_size={envs}",
                   "algorithm.eps_decay_over=0.3"] + extra)
    r = df["mean_episode_returns"].to_numpy()
    print(f"CURVE {algo} wall={time.time() - t0:.1f}s returns=" + " ".join(f"{x:.3f}" for x in r), flush=True)
