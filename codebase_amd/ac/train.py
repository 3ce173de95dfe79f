"""Actor-critic rollout collector with the reference's entry point shape - `_collect_trajectories`
(marlbase/ac/train.py:24-119) - on the fused HIP collector, and the reference's `main` loop (ac/train.py:155-228) around it; the update
itself is codebase_amd.ac.model.A2CNetwork / PPONetwork (csrc/a2c.hip)."""
from collections import namedtuple

import numpy as np
import torch

from .. import hip as _hip
from ..dqn.model import init_flat_params
from ..spaces import flatdim

Batch = namedtuple("Batch", ["obss", "actions", "rewards", "dones", "filled", "action_masks"])  # ac/train.py:14-16


class ActorNetworks:
    """The actor half of A2CNetwork (ac/model.py:44-60): per-agent FC nets, flat fp32 blocks on the device,
    reference state_dict key names (`actor.independent.{i}.network.{0,2,4}.{weight,bias}`)."""

    def __init__(self, obs_space, action_space, layers, use_orthogonal_init=True, device="cuda"):
        obs_dims = [flatdim(o) for o in obs_space]
        act_dims = [flatdim(a) for a in action_space]
        hidden = [int(h) for h in layers]
        if len(hidden) != 2 or hidden[0] != hidden[1] or len(set(obs_dims)) != 1 or len(set(act_dims)) != 1:
            raise NotImplementedError("actor shapes other than D-H-H-A shared by all agents")
        self.n_agents = len(obs_dims)
        self.device = torch.device(device)
        self.spec = _hip.NetSpec(self.n_agents, obs_dims[0], hidden[0], act_dims[0])
        params, _ = init_flat_params(obs_dims, hidden, act_dims, use_orthogonal_init)
        self.actor_params = params.to(self.device).contiguous()

    def logits(self, obs):
        """obs f32 [P][N][D] on the device -> logits [P][N][A]"""
        P, N, _ = obs.shape
        out = torch.empty(P, N, self.spec.n_actions, device=obs.device)
        _hip.dqn_act(self.spec, self.actor_params, obs, 0.0, u=torch.ones(N, device=obs.device),
                     rand_actions=torch.zeros(P, N, dtype=torch.int32, device=obs.device), q_out=out)
        return out


def _collect_trajectories(envs, model, max_ep_length, parallel_envs, n_agents, device, use_proper_termination, round_idx=0):
    """(t, Batch, infos) exactly as the reference returns them; `envs` is a HipForagingVecEnv, `model` carries
    `actor_params` / `spec` (ActorNetworks).  One kernel launch; the only host sync is reading `t` and the
    episode statistics at the end."""
    cfg = envs.cfg
    N, P, D, T = cfg.n_envs, cfg.n_agents, model.spec.obs_dim, int(max_ep_length)
    dev = model.actor_params.device
    b_obs = torch.empty(T + 1, N, P * D, device=dev)
    b_act = torch.empty(T, N, P, dtype=torch.int64, device=dev)
    b_rew = torch.empty(T, N, P, device=dev)
    b_done = torch.empty(T + 1, N, dtype=torch.uint8, device=dev)
    b_fill = torch.empty(T, N, device=dev)
    fin_ret = torch.zeros(P, N, device=dev)
    fin_len = torch.zeros(N, dtype=torch.int32, device=dev)
    t_max = torch.zeros(1, dtype=torch.int32, device=dev)
    _hip.ac_collect(cfg, model.spec, model.actor_params, round_idx, T, use_proper_termination, b_obs, b_act, b_rew, b_done,
                    b_fill, fin_ret, fin_len, t_max)
    t = int(t_max.item())
    ret, ln = fin_ret.cpu().numpy(), fin_len.cpu().numpy()
    infos = []
    for i in range(N):
        d = {"episode_returns": ret[:, i].copy(), "episode_length": int(ln[i])}
        for p in range(P):
            d[f"agent{p}/episode_returns"] = ret[p, i]
        infos.append(d)
    return t, Batch(b_obs, b_act, b_rew, b_done.bool(), b_fill, None), infos


def _log_progress(infos, step, updates, logger):
    infos.append({"updates": updates, "environment_steps": step})
    logger.log_metrics(infos)


def main(envs, eval_env, logger, time_limit, **cfg):
    """marlbase/ac/train.py:155-228: collect one rollout from every env, one update, log at eval_interval.
    `envs` is a HipForagingVecEnv (env.parallel_envs); the collector and the update are one library call each."""
    from pathlib import Path

    from ..config import instantiate
    from ..dqn.train import _cfg_get

    g = lambda k, d=None: _cfg_get(cfg, k, d)  # noqa: E731
    model = instantiate(g("model"), envs.single_observation_space, envs.single_action_space, cfg)
    logger.watch(model)
    parallel_envs = envs.observation_space[0].shape[0]
    device = g("model.device", "cuda")
    step = updates = last_eval = last_save = 0
    while step < g("total_steps") + 1:
        t, batch, infos = _collect_trajectories(envs, model, time_limit, parallel_envs, model.n_agents, device,
                                                g("use_proper_termination", False), round_idx=updates)
        infos.append(model.update(batch, step))
        if (step - last_eval) >= g("eval_interval"):
            _log_progress(infos, step, updates, logger)
            last_eval = step
        if g("save_interval") and (step - last_save) >= g("save_interval"):
            Path("checkpoints").mkdir(exist_ok=True)
            torch.save(model.state_dict(), f"checkpoints/model_s{step}.pt")
            last_save = step
        if g("video_interval"):
            raise NotImplementedError("video recording is outside the HIP hot path")
        updates += 1
        step += t * parallel_envs
    envs.close()
    return model
