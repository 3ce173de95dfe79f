"""IDQN driver with the reference's entry point - drop-in for `algorithm._target_: dqn.train.main`
(marlbase/configs/algorithm/idqn.yaml:4, marlbase/dqn/train.py:264-345).

Two paths behind the same `main(env, eval_env, logger, time_limit, **cfg)`:
  * `env` is a single env (make_env without parallel_envs): the reference's loop verbatim in
    behaviour - collect ONE episode through `model.act`/`env.step`, then at most one update of
    `batch_size` episodes (train.py:298-312) - every piece executing in the HIP library.
  * `env` is a HipForagingVecEnv (env.parallel_envs=N): the vectorised loop.  One ROUND = the fused
    collector gathers one episode from each of the N envs in a single launch, then U updates of B
    sampled episodes run back to back on the device (no host sync until evaluation).  Cadence is
    explicit in the config (`algorithm.updates_per_round`, `algorithm.batch_size`); the default
    keeps the reference's replay ratio: batch_size sampled episodes per collected episode, i.e.
    U = batch_size * N / B with B = N.
"""
import math
from pathlib import Path

import numpy as np
import torch

from .. import hip as _hip
from ..hip import Batch  # noqa: F401  (same field names as dqn/train.py:14-16)
from ..parallel import rank_sample_seed


class ReplayBuffer:
    """marlbase/dqn/train.py:19-124 on top of the device-resident episode-major store."""

    def __init__(self, buffer_size, n_agents, observation_space, action_space, max_episode_length, device,
                 store_action_masks=False):
        self.store_action_masks = store_action_masks
        self.buffer_size, self.n_agents, self.max_episode_length = buffer_size, n_agents, max_episode_length
        self.device = torch.device(device)
        D = int(np.prod(observation_space[0].shape))
        self.store = _hip.DeviceReplay(buffer_size, n_agents, D, max_episode_length, device=device)
        if store_action_masks:  # train.py:55-63: [P][T+1][cap][A] f32, kept next to the episode-major store, gathered by sample()
            self.action_masks = torch.zeros(n_agents, max_episode_length + 1, buffer_size, int(action_space[0].n), device=self.device)
        self.pos = self.cur_pos = self.t = 0
        self._slot = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._t = torch.zeros(1, dtype=torch.int32, device=self.device)

    def __len__(self):
        return min(self.pos, self.buffer_size)

    def _obs(self, obss):
        return torch.as_tensor(np.stack([np.asarray(o, np.float32) for o in obss])).reshape(self.n_agents, 1, -1).to(self.device)

    def init_episode(self, obss, action_masks=None):
        self.t = 0
        self._slot.fill_(self.cur_pos)
        self.store.init_episode(self._slot, self._obs(obss))
        if action_masks is not None:
            assert self.store_action_masks, "Action masks not stored in buffer!"
            self.action_masks[:, 0, self.cur_pos] = torch.as_tensor(np.asarray(action_masks, np.float32)).to(self.device)

    def add(self, obss, acts, rews, done, action_masks=None):
        assert self.t < self.max_episode_length, "Episode longer than given max length!"
        self._slot.fill_(self.cur_pos)
        self._t.fill_(self.t)
        dev = self.device
        self.store.add(self._slot, self._t, self._obs(obss),
                       torch.as_tensor(np.asarray(acts, np.int32).reshape(-1, 1)).to(dev),
                       torch.as_tensor(np.asarray(rews, np.float32).reshape(-1, 1)).to(dev),
                       torch.tensor([1 if done else 0], dtype=torch.uint8, device=dev))
        if action_masks is not None:
            assert self.store_action_masks, "Action masks not stored in buffer!"
            self.action_masks[:, self.t + 1, self.cur_pos] = torch.as_tensor(np.asarray(action_masks, np.float32)).to(dev)
        self.t += 1
        if done:
            self.pos += 1
            self.cur_pos = self.pos % self.buffer_size
            self.t = 0

    def can_sample(self, batch_size):
        return self.pos >= batch_size

    def sample(self, batch_size):
        idx = np.random.randint(0, len(self), size=batch_size)  # legacy global RNG, as train.py:95
        idx_d = torch.as_tensor(idx, dtype=torch.int32).to(self.device)
        batch = self.store.sample(batch_size, idx=idx_d, fresh=True)
        if self.store_action_masks:  # train.py:118-124
            batch = batch._replace(action_mask=self.action_masks[:, :, idx_d.long()].contiguous())
        return batch


def _epsilon_schedule(decay_style, decay_over, eps_start, eps_end, exp_decay_rate, total_steps):
    """marlbase/dqn/train.py:127-174 (same checks, same arithmetic)."""
    styles = {"linear": "lin", "lin": "lin", "exponential": "exp", "exp": "exp"}
    assert decay_style in styles, "decay_style must be one of 'linear' or 'exponential'"
    assert 0 <= eps_start <= 1 and 0 <= eps_end <= 1, "eps must be in [0, 1]"
    assert eps_start >= eps_end, "eps_start must be >= eps_end"
    assert 0 < decay_over <= 1, "decay_over must be in (0, 1]"
    assert total_steps > 0, "total_steps must be > 0"
    assert exp_decay_rate > 0, "eps_decay must be > 0"
    span = eps_start - eps_end
    horizon = total_steps * decay_over
    if styles[decay_style] == "lin":
        return lambda steps_done: max(eps_end + span * (1 - steps_done / horizon), eps_end)
    rate = span / horizon * exp_decay_rate
    return lambda steps_done: max(eps_end + span * math.exp(-rate * steps_done), eps_end)


def _episode(env, model, epsilon, rb=None, use_proper_termination=False):
    """one episode through the scalar API: _collect_trajectory (train.py:202-237) when `rb` is given,
    else one iteration of _evaluate (train.py:177-199)."""
    obss, info = env.reset()
    mask = np.stack(info["action_mask"]).astype(np.float32) if "action_mask" in info else None  # train.py:204-208
    if rb is not None:
        rb.init_episode(obss, mask)
    hiddens = model.init_hiddens(1)
    done, t = False, 0
    while not done:
        actions, hiddens = model.act(obss, hiddens, epsilon, mask)
        obss, rews, term, truncated, info = env.step(actions)
        done = term or truncated
        mask = np.stack(info["action_mask"]).astype(np.float32) if "action_mask" in info else None
        if rb is not None:
            rb.add(obss, actions, rews, term if use_proper_termination else done, mask)
        t += 1
    return t, info


def _collect_trajectory(env, model, rb, epsilon, use_proper_termination):
    return _episode(env, model, epsilon, rb, use_proper_termination)


def _evaluate(env, model, eval_episodes, eval_epsilon):
    return [_episode(env, model, eval_epsilon)[1] for _ in range(eval_episodes)]


_NO_FUSED_LOOP = bool(__import__("os").environ.get("MARLHIP_NO_FUSED_LOOP"))  # diagnostics: time the per-update host loop at 1 GPU


class VectorisedIDQN:
    """Device-resident training state of the vectorised path: N envs, replay shard, learner."""

    def __init__(self, lbf_cfg, model, buffer_episodes, time_limit, batch_size, updates_per_round, seed=0,
                 use_proper_termination=False, clear_stale=False, dist=None):
        self.cfg, self.model, self.T = lbf_cfg, model, int(time_limit)
        self.N = lbf_cfg.n_envs
        self.capacity = max(int(buffer_episodes) // self.N, 1) * self.N  # whole rounds
        self.replay = _hip.DeviceReplay(self.capacity, model.n_agents, model.spec.obs_dim, self.T, device=model.device)
        self.B, self.U = int(batch_size), int(updates_per_round)
        self.seed = int(seed)
        self.proper, self.clear_stale = bool(use_proper_termination), bool(clear_stale)
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        dev = model.device
        self.fin_return = torch.zeros(model.n_agents, self.N, device=dev)
        self.fin_length = torch.zeros(self.N, dtype=torch.int32, device=dev)
        # env-steps counted on the device: ONE elementwise launch per round (per-env episode lengths added into an accumulator), reduced
        # when somebody reads `env_steps` - a sum + add + conversion per round made the collector-only loop host-bound on a busy host
        self._len_acc = torch.zeros(self.N, dtype=torch.int64, device=dev)
        self.rounds = 0
        self.sample_counter = 0
        self.last_loss = None
        self._fused = None  # all U updates of a round from one library call
        self._sync = None
        if dist is not None and model.standardise_returns and model.mode == 0:
            # per-agent RunningMeanStd: batch moments summed over the ranks before the running update (standardise_stream.py:15-20 on the
            # global batch); VDN / QMIX keep one (mean, var) per batch column - a rank's columns are its own part of the global batch
            model.updater.ret_stats.attach_exchange(lambda t: dist.all_reduce(t))

    @property
    def env_steps(self):
        """0-dim int64 device tensor: transitions collected since construction (or the last reset_env_steps)"""
        return self._len_acc.sum()

    def reset_env_steps(self):
        self._len_acc.zero_()

    def job_steps_last_round(self):
        """env-steps the WHOLE JOB collected in the last round (int; one host read).  N > 1 ranks: the per-rank count of the round (<= N T:
        exact in a float) crosses the ranks through the SAME exchange as the gradients - the in-library peer-to-peer kernel where it is
        set up (one 1-workgroup launch; no collective-library launch, no second communicator), torch.distributed's all-reduce otherwise -
        so a round's only inter-rank traffic is of one kind."""
        t = self.fin_length.sum().to(torch.float32).reshape(1)
        if self.dist is not None:
            if self._sync is None:
                from ..parallel import GradSync

                g = self.model.updater
                self._sync = GradSync(self.dist, max_floats=getattr(g, "joint_grad", g.grad).numel())
            self._sync(t)
        return int(round(float(t.item())))

    def _grad_sync(self, grad):
        from ..parallel import GradSync

        if getattr(self, "_sync", None) is None:
            # all-reduce(SUM): the in-library peer-to-peer exchange where it can be set up, else RCCL / gloo; clip+Adam applies 1/world
            self._sync = GradSync(self.dist, max_floats=grad.numel())
        self._sync(grad)

    def _collect_recurrent(self, cfg, epsilon, round_idx, replay, slot_base, fin_return, fin_length):
        """_collect_trajectory for N envs with recurrent networks (use_rnn): the hidden state has to live between steps, so the
        round runs through the modular entry points - reset, then T x (sequence kernel with one step -> joint epsilon-greedy from
        the values -> env step -> replay add); same Philox streams and replay contents as the fused collector produces for
        feed-forward nets (tests/test_gpu_parity.py checks that equivalence for them)."""
        m, N, T = self.model, cfg.n_envs, self.T
        key = (cfg.n_envs, int(cfg.seed))
        if getattr(self, "_renv_key", None) != key:
            self._renv, self._renv_key = _hip.BatchedForaging(cfg), key
        env = self._renv
        env.cfg = cfg
        env.episode.fill_(round_idx)
        obs = env.reset()
        env.episode.fill_(round_idx)  # the action noise is keyed by the running episode's index
        dev = m.device
        slot = ((torch.arange(N, device=dev) + slot_base) % max(self.capacity, 1)).to(torch.int32)
        if replay is not None:
            replay.init_episode(slot, obs)
        alive = torch.ones(N, dtype=torch.uint8, device=dev)
        hid = None
        for t in range(T):  # every env of the round starts together: t is the episode step of the envs that are still alive
            out = m.q_values(obs, hid)
            q, hid = out if isinstance(out, tuple) else (out, None)  # feed-forward networks on the GEMM path carry no state
            acts = _hip.act_from_q(q, epsilon, cfg.seed, env.episode, env.ep_length)
            obs, _, done, trunc = env.step(acts, active=alive)
            if replay is not None:  # ReplayBuffer.add of the alive envs + alive &= ~finished: one call
                replay.add_step(slot, t, env, acts, alive, proper=self.proper)
            else:
                alive &= ((done | trunc) == 0).to(torch.uint8)
        fin_return.copy_(env.fin_return)
        fin_length.copy_(env.fin_length)

    def round(self, epsilon, train=True):
        """collect N episodes (one launch), then U updates; nothing here synchronises with the host."""
        m = self.model
        slot_base = (self.rounds * self.N) % self.capacity
        if getattr(m, "recurrent", False) or m.spec.wide:  # no fused collector: the modular loop
            self._collect_recurrent(self.cfg, epsilon, self.rounds, self.replay, slot_base, self.fin_return, self.fin_length)
        else:
            _hip.idqn_collect(self.cfg, m.spec, m.params, epsilon, self.rounds, self.replay, slot_base, self.fin_return,
                              self.fin_length, write_replay=True, clear_stale=self.clear_stale,
                              use_proper_termination=self.proper)
        self._len_acc.add_(self.fin_length)
        self.rounds += 1
        if train and self.U > 0 and m.mode != 2 and not m.standardise_returns and not _NO_FUSED_LOOP and not getattr(m, "recurrent", False) and not m.spec.wide and m.updater.optimizer == 0:  # the n-updates library call has no mixer / no return statistics (those loop here)
            # one library call for all U updates; with N > 1 ranks its data-parallel form: the gradient all-reduce is the only host hop
            if self._fused is None:
                self._fused = _hip.FusedLearner(m.updater, self.replay, self.B, m.target_update_interval_or_tau, mode=m.mode)
                if self.dist is not None and self._sync is None:
                    from ..parallel import GradSync

                    self._sync = GradSync(self.dist, max_floats=m.updater.grad.numel())
            length = min(self.rounds * self.N, self.capacity)
            m.updates, m.last_target_update = self._fused.run(self.U, length, rank_sample_seed(self.seed, self.rank), self.sample_counter,
                                                              m.updates, m.last_target_update,
                                                              grad_sync=self._sync if self.dist is not None else None, world=self.world)
            self.sample_counter += self.U
            self.last_loss = m.updater.loss
        elif train and m.mode == 2 and type(m.updater) is _hip.QmixUpdater and not _NO_FUSED_LOOP:
            # QMIX on the fused agent kernels: the U updates (loss/grad with the in-kernel gather, the joint [critic | mixer] gradient exchange,
            # both optimiser steps, target copies) from ONE library call (marlhip_qmix_update_n) - the same launches as the loop below
            if self._fused is None:
                self._fused = _hip.FusedQmixLearner(m.updater, self.replay, self.B, m.target_update_interval_or_tau)
                if self.dist is not None and self._sync is None:
                    from ..parallel import GradSync

                    self._sync = GradSync(self.dist, max_floats=m.updater.joint_grad.numel())
            length = min(self.rounds * self.N, self.capacity)
            m.updates, m.last_target_update = self._fused.run(self.U, length, rank_sample_seed(self.seed, self.rank), self.sample_counter,
                                                              m.updates, m.last_target_update,
                                                              grad_sync=self._sync if self.dist is not None else None, world=self.world)
            self.sample_counter += self.U
            self.last_loss = m.updater.loss
        elif train:
            length = min(self.rounds * self.N, self.capacity)
            sync = self._grad_sync if self.dist is not None else None
            for _ in range(self.U):
                self.last_loss = m.update_async(self.B, grad_sync=sync, world=self.world, replay=self.replay, length=length,
                                                seed=rank_sample_seed(self.seed, self.rank), counter=self.sample_counter)
                self.sample_counter += 1

    def evaluate(self, episodes, epsilon, round_idx=0):
        """_evaluate (train.py:177-199) for `episodes` envs in one collector launch (no replay writes);
        returns per-episode info dicts like RecordEpisodeStatistics emits.  With N > 1 ranks every rank plays its share of the
        episodes on its own env stream and the returns are gathered: the same list on every rank (rank 0 logs it)."""
        if self.dist is not None:
            from ..parallel import gather_stack

            per = -(-int(episodes) // self.world)
            ret, ln = self._evaluate_raw(per, epsilon, round_idx)
            ret = gather_stack(self.dist, ret).permute(1, 0, 2).reshape(ret.shape[0], -1)[:, :episodes]
            ln = gather_stack(self.dist, ln).reshape(-1)[:episodes]
        else:
            ret, ln = self._evaluate_raw(episodes, epsilon, round_idx)
        ret, ln = ret.cpu().numpy(), ln.cpu().numpy()
        infos = []
        for i in range(ret.shape[1]):
            d = {"episode_returns": ret[:, i].copy(), "episode_length": int(ln[i])}
            for p in range(ret.shape[0]):
                d[f"agent{p}/episode_returns"] = ret[p, i]
            infos.append(d)
        return infos

    def _evaluate_raw(self, episodes, epsilon, round_idx=0):
        """(returns [P][episodes], lengths [episodes]) on the device"""
        cfg = type(self.cfg).from_buffer_copy(self.cfg)
        cfg.n_envs = int(episodes)
        cfg.seed = (self.cfg.seed ^ 0x5DEECE66D) & (2**64 - 1)  # eval env: its own stream
        cfg.reward_stats = None  # the eval env has its own wrapper stack; only raw episode returns are reported
        dev = self.model.device
        ret = torch.zeros(self.model.n_agents, episodes, device=dev)
        ln = torch.zeros(episodes, dtype=torch.int32, device=dev)
        if getattr(self.model, "recurrent", False) or self.model.spec.wide:
            self._collect_recurrent(cfg, epsilon, round_idx, None, 0, ret, ln)
        else:
            _hip.idqn_collect(cfg, self.model.spec, self.model.params, epsilon, round_idx, self.replay, 0, ret, ln,
                              write_replay=False)
        return ret, ln


def _cfg_get(cfg, key, default=None):
    cur = cfg
    for part in key.split("."):
        if isinstance(cur, dict):
            if part not in cur:
                return default
            cur = cur[part]
        else:
            if not hasattr(cur, part) and not (hasattr(cur, "__contains__") and part in cur):
                return default
            cur = cur[part] if hasattr(cur, "__getitem__") else getattr(cur, part)
    return cur


def _make_model(cfg, env):
    from ..config import instantiate

    obs_space = getattr(env, "single_observation_space", None) or env.observation_space
    act_space = getattr(env, "single_action_space", None) or env.action_space
    return instantiate(_cfg_get(cfg, "model"), obs_space, act_space, cfg)


def main(env, eval_env, logger, time_limit, **cfg):
    # one process per GPU under torchrun (WORLD_SIZE > 1): envs and replay shard per rank, one gradient all-reduce per update, rank 0
    # logs / saves; `dist` is None on a single process and everything below is the single-GPU path
    from ..parallel import all_sum, init_distributed

    dist, rank, world, _ = init_distributed()
    model = _make_model(cfg, env)
    logger.watch(model)
    g = lambda k, d=None: _cfg_get(cfg, k, d)
    eps_sched = _epsilon_schedule(g("eps_decay_style"), g("eps_decay_over"), g("eps_start"), g("eps_end"),
                                  g("eps_exp_decay_rate"), g("total_steps"))
    total_steps = g("total_steps")
    vectorised = getattr(env, "n_envs", 1) > 1
    updates = step = last_eval = last_save = 0
    metrics = {}
    if vectorised:
        N = env.n_envs
        B = int(g("update_batch_size", 0) or N)
        U = int(g("updates_per_round", 0) or max(1, (g("batch_size") * N) // B))
        trainer = VectorisedIDQN(env.cfg, model, max(g("buffer_size"), N), time_limit, B, U, seed=env.cfg.seed,
                                 use_proper_termination=g("use_proper_termination", False), dist=dist)
        if dist is not None:  # identical replicas: every rank starts from rank 0's blocks (and Adam's zeros)
            for t in (model.params, model.target_params) + ((model.mixer_params, model.target_mixer_params) if model.mode == 2 else ()):
                dist.broadcast(t, 0)
    elif dist is not None:
        raise _hip.MarlHipError("multi-GPU training shards batched envs: set env.parallel_envs (the scalar reference loop is one process)")
    else:
        _, info0 = env.reset()  # train.py:265,281: the buffer stores masks iff the env's info carries them
        rb = ReplayBuffer(g("buffer_size"), env.unwrapped.n_agents, env.observation_space, env.action_space, time_limit,
                          _cfg_get(cfg, "model.device", "cuda"), store_action_masks="action_mask" in info0)
    while step < total_steps + 1:
        if vectorised:
            train = step > g("training_start") and trainer.rounds * N >= g("batch_size")
            trainer.round(eps_sched(step), train=train)
            # one host sync per round (N episodes); N > 1 ranks: the whole job's env-steps, the same number on every rank - it drives
            # the epsilon schedule, the training start and the loop's end, so all ranks issue the same collectives
            step += trainer.job_steps_last_round()
            if train:
                updates += trainer.U
                metrics = {"loss": float(trainer.last_loss[0].item())}
        else:
            t, _ = _collect_trajectory(env, model, rb, eps_sched(step), g("use_proper_termination", False))
            step += t
            if step > g("training_start") and rb.can_sample(g("batch_size")):
                metrics = model.update(rb.sample(g("batch_size")))
                updates += 1
            else:
                metrics = {}
        if g("eval_interval") and (step - last_eval) >= g("eval_interval"):
            if vectorised and getattr(trainer, "_sync", None) is not None:
                trainer._sync.check()  # every rank (`step` is the job's): a timed-out in-library exchange stops the run here, on all of them
            if vectorised:
                infos = trainer.evaluate(g("eval_episodes"), g("eps_evaluation"), round_idx=trainer.rounds)
            else:
                infos = _evaluate(eval_env, model, g("eval_episodes"), g("eps_evaluation"))
            if metrics:
                infos.append(metrics)
            infos.append({"updates": updates, "environment_steps": step, "epsilon": eps_sched(step)})
            if rank == 0:
                logger.log_metrics(infos)
            last_eval = step
        if g("video_interval"):
            raise NotImplementedError("video recording is outside the HIP hot path")
        if g("save_interval") and (step - last_save) >= g("save_interval"):
            if vectorised and getattr(trainer, "_sync", None) is not None:
                trainer._sync.check()  # never save replicas that have diverged
            if rank == 0:
                Path("checkpoints").mkdir(exist_ok=True)
                torch.save(model.state_dict(), f"checkpoints/model_s{step}.pt")
            last_save = step
    if vectorised and getattr(trainer, "_sync", None) is not None:
        trainer._sync.close()  # final check on every rank, then the exchange is freed behind a job-wide barrier
    env.close()
    return model
