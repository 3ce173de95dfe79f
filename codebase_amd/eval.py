"""Evaluate a saved run - the reference's `python eval.py path=<run dir> [load_step=N] [seed=S]` (marlbase/eval.py:17-65),
minus the video: reads `<path>/config.yaml` (written by the FileSystemLogger of either implementation), drops
`env.parallel_envs`, rebuilds the model through the config's `_target_`, loads `<path>/checkpoints/model_s{step}.pt`
(latest when `load_step` is not given; `torch.save(model.state_dict())` format, marlbase/dqn/train.py:340-343) and reports
the mean episode return over `episodes` evaluation episodes collected by the fused HIP collector.

    python -m codebase_amd.eval path=outputs/run1 [load_step=100000] [seed=0] [episodes=512]
"""
import sys
from pathlib import Path

import numpy as np
import torch
import yaml

from . import config as C
from . import hip as _hip


def latest_step(path):
    steps = [int(f.stem.split("_")[-1][1:]) for f in (Path(path) / "checkpoints").glob("model_s*.pt")]
    assert steps, f"No checkpoints under {path}/checkpoints"
    return max(steps)


def evaluate(model, lbf_cfg, episodes, time_limit, epsilon=0.0, round_idx=0):
    """mean per-episode team return and length of `episodes` fresh episodes (greedy DQN-family models: epsilon as given;
    actor-critic models sample from their policy like the reference's act())"""
    cfg = type(lbf_cfg).from_buffer_copy(lbf_cfg)
    cfg.n_envs = int(episodes)
    cfg.reward_stats = None  # only RecordEpisodeStatistics' raw returns are reported
    dev = model.device
    ret = torch.zeros(model.n_agents, episodes, device=dev)
    ln = torch.zeros(episodes, dtype=torch.int32, device=dev)
    if hasattr(model, "actor_params") and getattr(model, "recurrent", False):  # use_rnn actors: the modular rollout loop
        from .ac.train import _collect_trajectories_recurrent
        from .utils.envs import HipForagingVecEnv

        _, _, r, l = _collect_trajectories_recurrent(HipForagingVecEnv(cfg), model, int(time_limit), False, round_idx)
        ret.copy_(r)
        ln.copy_(l)
    elif hasattr(model, "actor_params"):
        P, D, T = model.n_agents, model.spec.obs_dim, int(time_limit)
        t_max = torch.zeros(1, dtype=torch.int32, device=dev)
        _hip.ac_collect(cfg, model.spec, model.actor_params, round_idx, T, False, torch.empty(T + 1, episodes, P * D, device=dev),
                        torch.empty(T, episodes, P, dtype=torch.int64, device=dev), torch.empty(T, episodes, P, device=dev),
                        torch.empty(T + 1, episodes, dtype=torch.uint8, device=dev), torch.empty(T, episodes, device=dev), ret, ln, t_max)
    elif getattr(model, "recurrent", False):  # use_rnn: the hidden state lives between steps -> the modular collection loop
        from .dqn.train import VectorisedIDQN

        tr = VectorisedIDQN(cfg, model, episodes, int(time_limit), 1, 0)
        tr._collect_recurrent(cfg, epsilon, round_idx, None, 0, ret, ln)
    else:
        replay = _hip.DeviceReplay(episodes, model.n_agents, model.spec.obs_dim, int(time_limit), device=dev)
        _hip.idqn_collect(cfg, model.spec, model.params, epsilon, round_idx, replay, 0, ret, ln, write_replay=False)
    return float(ret.sum(0).mean().item()), float(ln.float().mean().item())


def main(argv=None):
    args = dict(a.split("=", 1) for a in (sys.argv[1:] if argv is None else argv))
    path = Path(args["path"])
    assert path.is_dir(), f"Path {path} is not a directory."
    cfg_path = path / "config.yaml"
    assert cfg_path.exists(), f"Config file {cfg_path} does not exist."
    run_config = C.to_cfg(yaml.safe_load(open(cfg_path)))
    run_config.env.pop("parallel_envs", None)
    seed = int(args["seed"]) if "seed" in args else run_config.get("seed")
    env = C.call(run_config.env, seed=seed)
    if seed is not None:
        torch.manual_seed(seed)
        np.random.seed(seed)
    step = int(args["load_step"]) if "load_step" in args else latest_step(path)
    ckpt = path / "checkpoints" / f"model_s{step}.pt"
    assert ckpt.exists(), f"Checkpoint {ckpt} does not exist."
    algo = run_config.algorithm
    # run directories written by the reference carry `model.device: cpu` (configs/algorithm/*.yaml); the networks here live on
    # the HIP device whatever the run was trained on, so the stored device is replaced (device=<...> on the command line wins)
    if "device" in algo.model:
        algo.model["device"] = args.get("device", "cuda")
    obs_space = getattr(env, "single_observation_space", None) or env.observation_space
    act_space = getattr(env, "single_action_space", None) or env.action_space
    model = C.instantiate(algo.model, obs_space, act_space, algo)
    print(f"Loading model from {ckpt}")
    model.load_state_dict(torch.load(ckpt, weights_only=True))
    episodes = int(args.get("episodes", 512))
    mean_ret, mean_len = evaluate(model, env.cfg, episodes, run_config.env.time_limit, float(algo.get("eps_evaluation", 0.0)))
    print(f"step {step}: mean episode return {mean_ret:.4f}, mean length {mean_len:.2f} over {episodes} episodes")
    env.close()
    return {"step": step, "mean_episode_returns": mean_ret, "mean_episode_length": mean_len}


if __name__ == "__main__":
    main()
