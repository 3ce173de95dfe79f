python - <<'PY'
import time, os, tempfile
os.environ["MARLHIP_RUN_DIR"] = tempfile.mkdtemp()
from codebase_amd import run
for algo, extra, steps in (("ia2c", ["algorithm.model.actor.layers=[64,64]", "algorithm.model.critic.layers=[64,64]"], 200_000_000),
                           ("idqn", ["algorithm.model.layers=[64,64]", "algorithm.updates_per_round=32", "algorithm.batch_size=4096", "algorithm.lr=3e-3", "algorithm.target_update_interval_or_tau=0.1"], 100_000_000)):
    os.environ["MARLHIP_RUN_DIR"] = tempfile.mkdtemp()
    t0 = time.time()
    df = run.main([f"+algorithm={algo}", "env.name=lbforaging:Foraging-8x8-2p-3f-v3", "env.time_limit=25", "env.parallel_envs=4096", "seed=1",
                   f"algorithm.total_steps={steps}", f"algorithm.eval_interval={steps // 4}", "algorithm.eval_episodes=256"] + extra)
    dt = time.time() - t0
    print(f"DRIVER {algo}: {steps / dt / 1e6:.1f} M env-steps/s wall-clock incl. start-up ({dt:.1f} s), last mean return {float(df['mean_episode_returns'].iloc[-1]):.3f}")
PY
