"""Actor-critic rollout collector with the reference's entry point shape - `_collect_trajectories`
(marlbase/ac/train.py:24-119) - on the fused HIP collector.  The A2C / PPO update itself
(marlbase/ac/model.py:189-352) is a "next" row; `main` says so instead of falling back."""
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


def main(envs, eval_env, logger, time_limit, **cfg):
    raise NotImplementedError("the A2C/PPO update (marlbase/ac/model.py:189-352) is a 'next' row (DESIGN.md); "
                              "codebase_amd.ac.train._collect_trajectories (the rollout collector) is built")
