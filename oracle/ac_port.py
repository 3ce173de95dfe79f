"""CPU restatement of the reference's actor-critic rollout collector and of the vector env it drives.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PINNED: `oracle/make_golden.py::ac_fixture` runs the reference's own `_collect_trajectories`
(marlbase/ac/train.py:24-119, imported unmodified) on `OracleVecEnv` with a scripted policy and freezes the
Batch it returns; tests/test_oracle_learner.py re-checks `collect_trajectories` below against it.

  OracleVecEnv          gymnasium(<1.0) AsyncVectorEnv semantics over oracle.lbf.MarlbaseEnv instances
                        (utils/envs.py:11-65): step() auto-resets a finished env, returns the NEW episode's
                        first observation and puts the finished episode's info into info["final_info"][i]
  collect_trajectories  marlbase/ac/train.py:24-119 in numpy
  sample_inverse_cdf    the HIP collector's Categorical sampler (fp32 softmax, inverse CDF, one uniform)
"""
import numpy as np

from oracle.lbf import MarlbaseEnv
from oracle.philox import DrawStream, act_noise


class OracleVecEnv:
    def __init__(self, name, n_envs, time_limit, seed, cooperative=False, **overrides):
        self.envs = [MarlbaseEnv(name, time_limit, cooperative=cooperative, **overrides) for _ in range(n_envs)]
        self.n_envs, self.seed = n_envs, seed
        self.n_agents = self.envs[0].n_agents
        self.obs_dim = self.envs[0].env.obs_dim
        self.episode = [0] * n_envs  # Philox reset-stream index of the NEXT reset of each env

    def _reset_one(self, i):
        obs, _ = self.envs[i].reset(DrawStream(self.seed, i, self.episode[i]))
        self.episode[i] += 1
        return obs

    def set_episode(self, k):
        self.episode = [k] * self.n_envs

    def reset(self):
        obs = [self._reset_one(i) for i in range(self.n_envs)]
        return tuple(np.stack([o[p] for o in obs]) for p in range(self.n_agents)), {}

    def step(self, actions):
        """actions [N][P] ints -> (tuple of P arrays [N,D], rewards [N,P], done [N], truncated [N], info)"""
        obs, rews, dones, truncs = [], [], [], []
        final = [None] * self.n_envs
        for i, e in enumerate(self.envs):
            o, r, d, tr, info = e.step([int(a) for a in actions[i]])
            if d or tr:
                final[i] = info
                o = self._reset_one(i)
            obs.append(o)
            rews.append(r)
            dones.append(d)
            truncs.append(tr)
        info = {"final_info": final} if any(f is not None for f in final) else {}
        return (tuple(np.stack([o[p] for o in obs]) for p in range(self.n_agents)), np.array(rews, np.float32),
                np.array(dones), np.array(truncs), info)


def collect_trajectories(envs, act_fn, max_ep_length, use_proper_termination=False):
    """act_fn(obss tuple of [N,D], t) -> int actions [N][P].  Returns (t, batch dict, infos)."""
    N, P, D = envs.n_envs, envs.n_agents, envs.obs_dim
    running = np.ones(N, bool)
    obss, _ = envs.reset()
    b_obs = np.zeros((max_ep_length + 1, N, P * D), np.float32)
    b_done = np.zeros((max_ep_length + 1, N), bool)
    b_act = np.zeros((max_ep_length, N, P), np.int64)
    b_rew = np.zeros((max_ep_length, N, P), np.float32)
    b_fill = np.zeros((max_ep_length, N), np.float32)
    b_obs[0] = np.concatenate(obss, axis=-1)
    t, infos = 0, []
    while running.any():
        actions = np.asarray(act_fn(obss, t))
        next_obss, rewards, done, truncated, info = envs.step(actions)
        if not use_proper_termination:
            done = np.logical_or(done, truncated)
        b_obs[t + 1, running] = np.concatenate(next_obss, axis=1)[running]
        b_act[t, running] = actions[running]
        b_done[t + 1, running] = done[running]
        b_rew[t, running] = rewards[running]
        b_fill[t, running] = 1
        if done.any():
            for i, d in enumerate(done):
                if d:
                    # (the reference appends final_info for every done env, running or not - ac/train.py:101-110)
                    infos.append((i, info["final_info"][i]))
                    running[i] = False
        t += 1
        obss = next_obss
    return t, dict(obss=b_obs, actions=b_act, rewards=b_rew, dones=b_done, filled=b_fill), infos


def sample_inverse_cdf(logits, u):
    """first a with cumsum(exp(l - max))[a] > u * sum, in fp32, sequential sums (csrc/ac_collect.hip sample_rows)"""
    l = np.asarray(logits, np.float32)
    e = np.exp(l - l.max(), dtype=np.float32)
    s = np.float32(0)
    for v in e:
        s = np.float32(s + v)
    thr = np.float32(np.float32(u) * s)
    c = np.float32(0)
    for a, v in enumerate(e):
        c = np.float32(c + v)
        if c > thr:
            return a
    return len(e) - 1


def step_uniforms(seed, env, episode, t, n_agents):
    """the collector's uniform of agent p at step t: word 1+p of the action-noise block (u01 of its top 24 bits)"""
    from oracle.philox import MASK, STREAM_ACT, philox4x32_10, u01_f32

    key = (seed & MASK, (seed >> 32) & MASK)
    words = []
    for k in range((1 + n_agents + 3) // 4):
        words += philox4x32_10((env, episode, t | (k << 16), STREAM_ACT), key)
    return [u01_f32(words[1 + p]) for p in range(n_agents)]
