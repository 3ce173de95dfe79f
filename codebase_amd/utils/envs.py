"""Env factory with the reference's signature - drop-in for `env._target_: utils.envs.make_env`
(marlbase/configs/default.yaml:28-35, marlbase/utils/envs.py:68-119) - backed by the batched HIP
Level-Based Foraging / multi-robot warehouse (rware) envs instead of gym.make + wrapper classes.

    make_env(seed, enable_video=False, name=..., time_limit=..., clear_info=False, observe_id=False,
             standardise_rewards=False, wrappers=None[, parallel_envs=N])

`parallel_envs` absent/0 -> HipForagingEnv: ONE env with the Gymnasium-style MARL API the reference's
drivers use (tuple observations, LIST rewards, scalar done/truncated, README.md:69).  Every call
synchronises with the GPU - it exists so that reference-style python loops (and eval.py) run
unchanged, not for speed.
`parallel_envs` = N -> HipForagingVecEnv: N envs in HBM; the vectorised IDQN driver
(codebase_amd/dqn/train.py) hands its config to the fused collector and never steps it from python;
`step()` offers AsyncVectorEnv semantics (auto-reset, info["final_info"]) for AC-style collectors
(marlbase/ac/train.py:24-119).

The wrapper stack of utils/envs.py:93-109 is folded into the kernels: TimeLimit -> `truncated`,
RecordEpisodeStatistics -> info on episode end, CooperativeReward -> cfg.cooperative, StandardiseReward ->
cfg.reward_stats (one streaming record per env), ObserveID -> cfg.observe_id.  FlattenObservation (the envs here
already return flat vectors), other wrapper names and video raise NotImplementedError rather than silently doing
something else.  `rware:` ids build the same two classes over the warehouse kernels (5 actions, 71-float observations).
"""
import random
from time import perf_counter

import numpy as np
import torch

from .. import hip as _hip
from .. import spaces


def _space_pair(cfg):
    P = cfg.n_agents
    if _hip.is_rware(cfg):  # rware: Box(-inf, inf, (71,)) per agent, Discrete(5)
        D, A = _hip.env_dims(cfg)
        obs = spaces.Tuple([spaces.Box(np.full(D, -np.inf, np.float32), np.full(D, np.inf, np.float32)) for _ in range(P)])
        return obs, spaces.Tuple([spaces.Discrete(A) for _ in range(P)])
    F = cfg.n_food
    low = np.array([-1, -1, 0] * (F + P), np.float32)
    high = np.array([cfg.rows - 1, cfg.cols - 1, cfg.max_player_level * min(P, 3)] * F
                    + [cfg.rows - 1, cfg.cols - 1, cfg.max_player_level] * P, np.float32)
    if cfg.observe_id:  # ObserveID widens every agent's Box by n_agents (utils/wrappers.py:81-95: unbounded there)
        low, high = np.concatenate((np.zeros(P, np.float32), low)), np.concatenate((np.ones(P, np.float32), high))
    obs = spaces.Tuple([spaces.Box(low, high) for _ in range(P)])
    act = spaces.Tuple([spaces.Discrete(6) for _ in range(P)])
    return obs, act


def _build_cfg(name, time_limit, clear_info, observe_id, standardise_rewards, wrappers, seed, n_envs, kwargs):
    if "Foraging" not in name and "rware" not in name:
        raise NotImplementedError(f"{name}: only Level-Based Foraging and rware ids have a HIP env "
                                  "(smaclite is listed under 'next' in DESIGN.md)")
    kwargs = dict(kwargs, observe_id=int(bool(observe_id)))  # ObserveID (utils/wrappers.py:73-103): one-hot prefix, in-kernel
    cooperative = False
    for w in wrappers or []:
        if w == "CooperativeReward":
            cooperative = True
        else:
            raise NotImplementedError(f"wrapper {w} is not available on the HIP env")
    if seed is None:
        seed = random.randint(0, 99999)  # utils/envs.py:58-59
    cfg = _hip.env_config(name, n_envs, time_limit, seed=seed, cooperative=cooperative, **kwargs)
    if standardise_rewards:  # StandardiseReward (utils/wrappers.py:111-142): one streaming record per env, kept with the cfg
        _hip.attach_reward_stats(cfg)
    return cfg


class HipForagingEnv:
    """One env, Gymnasium-style MARL API (see module docstring)."""

    metadata = {"render_modes": []}

    def __init__(self, cfg, clear_info=False):
        self.cfg = cfg
        self.batched = _hip.BatchedForaging(cfg)
        self.n_agents = cfg.n_agents
        self.n_envs = 1
        self.clear_info = clear_info
        self.observation_space, self.action_space = _space_pair(cfg)
        self._actions = torch.zeros(cfg.n_agents, 1, dtype=torch.int32, device=self.batched.device)
        self.t0 = perf_counter()

    @property
    def unwrapped(self):
        return self

    def _obs_tuple(self, obs):
        o = obs.cpu().numpy()
        return tuple(o[p, 0].copy() for p in range(self.n_agents))

    def reset(self, seed=None, options=None):
        if seed is not None:  # re-key the Philox streams and restart the episode counter
            self.cfg.seed = int(seed) & (2**64 - 1)
            self.batched.episode.zero_()
        self.t0 = perf_counter()
        return self._obs_tuple(self.batched.reset()), {}

    def step(self, actions):
        self._actions.copy_(torch.as_tensor(np.asarray(actions, dtype=np.int32).reshape(-1, 1)))
        obs, rew, done, trunc = self.batched.step(self._actions)
        rew = rew.cpu().numpy()[:, 0]
        done, trunc = bool(done.item()), bool(trunc.item())
        info = {}
        if (done or trunc) and not self.clear_info:
            ret = self.batched.fin_return.cpu().numpy()[:, 0].copy()
            info["episode_returns"] = ret  # wrappers.py:35-41
            for i, r in enumerate(ret):
                info[f"agent{i}/episode_returns"] = r
            info["episode_length"] = int(self.batched.fin_length.item())
            info["episode_time"] = perf_counter() - self.t0
        return self._obs_tuple(obs), [float(r) for r in rew], done, trunc, info

    def render(self):
        raise NotImplementedError("rendering / video is outside the HIP hot path")

    def close(self):
        pass


class HipForagingVecEnv:
    """N envs in HBM.  Tensors stay on the device; see module docstring."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.batched = _hip.BatchedForaging(cfg)
        self.n_agents = cfg.n_agents
        self.n_envs = self.num_envs = cfg.n_envs
        self.single_observation_space, self.single_action_space = _space_pair(cfg)
        D = self.batched.D
        self.observation_space = spaces.Tuple([spaces.Box(-1.0, 255.0, shape=(cfg.n_envs, D)) for _ in range(cfg.n_agents)])
        self.action_space = spaces.Tuple([spaces.Discrete(self.batched.A) for _ in range(cfg.n_agents)])

    @property
    def unwrapped(self):
        return self

    def reset(self, seed=None, options=None):
        if seed is not None:
            self.cfg.seed = int(seed) & (2**64 - 1)
            self.batched.episode.zero_()
        obs = self.batched.reset()
        return tuple(obs[p] for p in range(self.n_agents)), {}

    def step(self, actions):
        """actions: int tensor/array [P][N] (or list of N joint actions).  AsyncVectorEnv semantics:
        finished envs are reset inside the call; info["final_info"][i] carries their statistics."""
        a = torch.as_tensor(actions, device=self.batched.device).to(torch.int32)
        if a.shape != (self.n_agents, self.n_envs):
            a = a.reshape(self.n_envs, self.n_agents).t()
        obs, rew, done, trunc = self.batched.step(a.contiguous(), auto_reset=True)
        info = {}
        fin = ((done | trunc) > 0)
        if bool(fin.any()):
            idx = torch.nonzero(fin).flatten().cpu().numpy()
            ret = self.batched.fin_return.cpu().numpy()
            ln = self.batched.fin_length.cpu().numpy()
            final = [None] * self.n_envs
            for i in idx:
                d = {"episode_returns": ret[:, i].copy(), "episode_length": int(ln[i])}
                for p in range(self.n_agents):
                    d[f"agent{p}/episode_returns"] = ret[p, i]
                final[i] = d
            info["final_info"] = final
            info["final_observation"] = self.batched.final_obs
        return tuple(obs[p] for p in range(self.n_agents)), rew.t(), done.bool(), trunc.bool(), info

    def close(self):
        pass


def make_env(seed, enable_video=False, **env_config):
    env_config = dict(env_config)
    parallel = env_config.pop("parallel_envs", None)
    name = env_config.pop("name")
    time_limit = env_config.pop("time_limit")
    clear_info = env_config.pop("clear_info", False)
    observe_id = env_config.pop("observe_id", False)
    standardise_rewards = env_config.pop("standardise_rewards", False)
    wrappers = env_config.pop("wrappers", None)
    env_config.pop("_target_", None)
    if enable_video:
        raise NotImplementedError("video recording is outside the HIP hot path")
    cfg = _build_cfg(name, time_limit, clear_info, observe_id, standardise_rewards, wrappers, seed,
                     int(parallel) if parallel else 1, env_config)
    if parallel:
        return HipForagingVecEnv(cfg)
    return HipForagingEnv(cfg, clear_info=clear_info)
