"""Learning-curve probe for the vectorised IDQN path (GPU): python scripts/learn_curve.py N U B STEPS [lr]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codebase_amd import run

N, U, B, STEPS = (int(x) for x in sys.argv[1:5])
extra = sys.argv[5:]
d = tempfile.mkdtemp()
os.environ["MARLHIP_RUN_DIR"] = d
t0 = time.time()
df = run.main(["+algorithm=idqn", "env.name=lbforaging:Foraging-8x8-2p-3f-v3", "env.time_limit=25", f"env.parallel_envs={N}",
               "algorithm.model.layers=[64,64]", "seed=1", f"algorithm.total_steps={STEPS}", f"algorithm.eval_interval={STEPS // 10}",
               "algorithm.eval_episodes=512", f"algorithm.updates_per_round={U}", f"algorithm.update_batch_size={B}"] + extra)
print(f"N={N} U={U} B={B} steps={STEPS} wall={time.time() - t0:.1f}s")
print(df[["updates", "mean_episode_returns", "loss", "epsilon"]].to_string())
