"""Plain-torch / numpy CPU restatement of the reference's IDQN learner.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PINNED: `oracle/make_golden.py` runs these functions next to the reference's own
classes (imported unmodified from /root/reference with four sys.modules stubs) on
seeded inputs and freezes the results under tests/golden/; `tests/test_oracle_learner.py`
re-checks this file against those vectors on every run.

Follows, line for line in behaviour:
  ReplayBuffer            marlbase/dqn/train.py:19-124
  _epsilon_schedule       marlbase/dqn/train.py:127-174
  FCNetwork / orthogonal  marlbase/utils/models.py:8-48
  QNetwork.act            marlbase/dqn/model.py:94-116
  QNetwork._compute_loss  marlbase/dqn/model.py:118-163   (VDN: :224-269)
  QNetwork.update         marlbase/dqn/model.py:165-196
Parameters live in one flat fp32 block per agent in torch parameters() order
(W1 b1 W2 b2 W3 b3) - the layout the HIP library uses.
"""
import math

import numpy as np
import torch


def _widths(H):
    """H: one width (two equal hidden layers) or a list of hidden widths - FCNetwork takes any list (utils/models.py:34-42)"""
    return (H, H) if isinstance(H, int) else tuple(int(h) for h in H)


def _layer_shapes(D, H, A):
    dims = (D,) + _widths(H) + (A,)
    return [(dims[i + 1], dims[i]) for i in range(len(dims) - 1)]


def nparams(D, H, A):
    return sum(o * i + o for o, i in _layer_shapes(D, H, A))


def split(block, D, H, A):
    """flat [nparams] -> (W1[H,D], b1[H], W2[H,H], b2[H], W3[A,H], b3[A]) views"""
    o = 0
    out = []
    for shape in [s_ for (a, b) in _layer_shapes(D, H, A) for s_ in ((a, b), (a,))]:
        n = int(np.prod(shape))
        out.append(block[o:o + n].reshape(shape))
        o += n
    return out


def init_params(P, D, H, A, seed=0, orthogonal=True):
    """FCNetwork init (utils/models.py:8-11,34-42): orthogonal gain sqrt(2), zero bias."""
    g = torch.Generator().manual_seed(seed)
    blocks = []
    for _ in range(P):
        parts = []
        for (o, i) in _layer_shapes(D, H, A):
            w = torch.empty(o, i)
            if orthogonal:
                torch.nn.init.orthogonal_(w, gain=math.sqrt(2), generator=g)
            else:
                torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5), generator=g)
            parts += [w.reshape(-1), torch.zeros(o)]
        blocks.append(torch.cat(parts))
    return torch.stack(blocks)


def mlp(block, x, D, H, A):
    t = split(block, D, H, A)
    h = x
    for k in range(0, len(t) - 2, 2):
        h = torch.relu(torch.nn.functional.linear(h, t[k], t[k + 1]))
    return torch.nn.functional.linear(h, t[-2], t[-1])


def q_values(params, obss, D, H, A):
    """obss [P, ..., D] -> [P, ..., A]"""
    return torch.stack([mlp(params[p], obss[p], D, H, A) for p in range(params.shape[0])])


def act(params, obs, epsilon, u, rand_actions, D, H, A):
    """Batched QNetwork.act: obs [P,N,D]; ONE uniform u[N] per env for the joint action
    (model.py:105); greedy = torch.argmax (first max).  Returns actions [P,N] int64."""
    with torch.no_grad():
        q = q_values(params, obs, D, H, A)
    greedy = q.argmax(-1)
    explore = (epsilon > u).unsqueeze(0)
    return torch.where(explore, rand_actions, greedy), q


class RunningMeanStd:
    """marlbase/utils/standardise_stream.py:6-41 (parallel-variance update; `count` starts at epsilon = 1e-4)."""

    def __init__(self, shape, epsilon=1e-4):
        self.mean = torch.zeros(shape, dtype=torch.float32)
        self.var = torch.ones(shape, dtype=torch.float32)
        self.count = epsilon

    def update(self, arr):
        arr = arr.reshape(-1, arr.size(-1))
        bm, bv, bc = torch.mean(arr, dim=0), torch.var(arr, dim=0), arr.shape[0]
        delta = bm - self.mean
        tot = self.count + bc
        new_mean = self.mean + delta * bc / tot
        m2 = self.var * self.count + bv * bc + torch.square(delta) * self.count * bc / (self.count + bc)
        self.mean, self.var, self.count = new_mean, m2 / (self.count + bc), bc + self.count


def compute_loss(params, tparams, batch, gamma, double_q, D, H, A, mode="idqn", sharing=None, ret_ms=None):
    """QNetwork._compute_loss (idqn) / VDNetwork._compute_loss (vdn); batch = dict with
    obss [P,T+1,B,D], actions i64 [P,T,B], rewards [P,T,B], dones [T+1,B], filled [T,B].
    sharing: agent -> network index (MultiAgentSharedNetwork, utils/models.py:176-300); params are then [K][n]."""
    if sharing is not None:
        params, tparams = params[list(sharing)], tparams[list(sharing)]
    obss, actions, rewards = batch["obss"], batch["actions"].unsqueeze(-1), batch["rewards"]
    dones, filled = batch["dones"], batch["filled"]
    P = obss.shape[0]
    q = q_values(params, obss, D, H, A)
    chosen = q[:, :-1].gather(-1, actions).squeeze(-1)
    mask = batch.get("action_mask")  # [P][T+1][B][A] f32 or absent (model.py:124,133-143)
    with torch.no_grad():
        tq = q_values(tparams, obss, D, H, A)[:, 1:]
        if mask is not None:
            tq = torch.where(mask[:, 1:] == 0, torch.full_like(tq, -1e8), tq)
    if double_q:
        qd = q.detach()[:, 1:]
        if mask is not None:
            qd = torch.where(mask[:, 1:] == 0, torch.full_like(qd, -1e8), qd)
        a_prime = qd.argmax(-1)
        target_qs = tq.gather(-1, a_prime.unsqueeze(-1)).squeeze(-1)
    else:
        target_qs, _ = tq.max(dim=-1)
    if mode == "idqn":
        d = dones[1:].unsqueeze(0).repeat(P, 1, 1)
        if ret_ms is not None:  # standardise_returns (model.py:146-149): bootstrap values back to the raw scale
            target_qs = (target_qs.permute(1, 2, 0) * torch.sqrt(ret_ms.var) + ret_ms.mean).permute(2, 0, 1)
        returns = rewards + gamma * target_qs.detach() * (1 - d)
        if ret_ms is not None:  # model.py:153-158: update with EVERY entry (filled or not), then standardise
            r = returns.permute(1, 2, 0)
            ret_ms.update(r)
            returns = ((r - ret_ms.mean) / torch.sqrt(ret_ms.var)).permute(2, 0, 1)
        loss = torch.nn.functional.mse_loss(chosen, returns.detach(), reduction="none").sum(dim=0)
    else:  # vdn
        chosen = chosen.sum(dim=0)
        returns = rewards[0] + gamma * target_qs.sum(dim=0).detach() * (1 - dones[1:])
        loss = torch.nn.functional.mse_loss(chosen, returns.detach(), reduction="none")
    return (loss * filled).sum() / filled.sum()


class Learner:
    """QNetwork.update (model.py:165-196): backward, clip_grad_norm_ over all agents'
    parameters, torch.optim.Adam, hard / soft target update."""

    def __init__(self, params, D, H, A, lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True,
                 target_update_interval_or_tau=200, mode="idqn", sharing=None, standardise_returns=False, optimizer="Adam"):
        self.D, self.H, self.A = D, H, A
        self.sharing = sharing
        self.ret_ms = RunningMeanStd((params.shape[0],)) if standardise_returns else None
        P = params.shape[0]
        # one Parameter per tensor, in parameters() order, so clip/Adam see the reference's tensor list
        self.tensors = [torch.nn.Parameter(t.clone()) for p in range(P) for t in split(params[p], D, H, A)]
        self.P = P
        self.target = params.clone()
        self.opt = getattr(torch.optim, optimizer)(self.tensors, lr=lr)  # dqn/model.py:66-71
        self.gamma, self.grad_clip, self.double_q = gamma, grad_clip, double_q
        self.tui = target_update_interval_or_tau
        self.updates = 0
        self.last_target_update = 0
        self.mode = mode

    def flat(self):
        per = len(self.tensors) // self.P
        return torch.stack([torch.cat([t.reshape(-1) for t in self.tensors[p * per:(p + 1) * per]]) for p in range(self.P)])

    def update(self, batch, chunks=1):
        """chunks > 1 (the at-size tests: bounded memory in float64): the loss is a filled-weighted mean over batch columns, so the
        gradient is the sum over column chunks of (chunk loss) x (chunk's filled / total filled) - the same number, accumulated in .grad"""
        self.opt.zero_grad()
        if chunks > 1:
            assert self.ret_ms is None, "column chunks and standardise_returns (batch-wide statistics) do not combine"
            total, loss = batch["filled"].sum(), 0.0
            for cols in torch.arange(batch["filled"].shape[1]).chunk(chunks):
                sub = column_chunk(batch, cols)
                part = compute_loss(self.flat(), self.target, sub, self.gamma, self.double_q, self.D, self.H, self.A, self.mode,
                                    self.sharing, None) * (sub["filled"].sum() / total)
                part.backward()
                loss = loss + part.detach()
        else:
            loss = compute_loss(self.flat(), self.target, batch, self.gamma, self.double_q, self.D, self.H, self.A, self.mode,
                                self.sharing, self.ret_ms)
            loss.backward()
        gnorm = None
        per = len(self.tensors) // self.P  # the gradient as _compute_loss leaves it (before clipping), [P][n]: for tests that compare it
        self.last_grad = torch.stack([torch.cat([t.grad.reshape(-1) for t in self.tensors[p * per:(p + 1) * per]]) for p in range(self.P)]).clone()
        if self.grad_clip:
            gnorm = torch.nn.utils.clip_grad_norm_(self.tensors, self.grad_clip)
        self.opt.step()
        self.updates += 1
        if self.tui > 1.0 and (self.updates - self.last_target_update) >= self.tui:
            self.target = self.flat().detach().clone()
            self.last_target_update = self.updates
        elif self.tui < 1.0:
            self.target = (1 - self.tui) * self.target + self.tui * self.flat().detach()
        return {"loss": loss.item(), "grad_norm": None if gnorm is None else float(gnorm)}


def column_chunk(batch, cols):
    """the Batch restricted to the batch columns `cols` (obss / actions / rewards [P, T(+1), B, ...], dones / filled [T(+1), B])"""
    return {k: (v[:, :, cols] if k in ("obss", "actions", "rewards", "action_mask") else v[:, cols]) for k, v in batch.items()}


class ReplayBuffer:
    """marlbase/dqn/train.py:19-124 (time-major numpy storage, ring over episodes; slots are
    never cleared on re-use)."""

    def __init__(self, buffer_size, n_agents, obs_dim, max_episode_length):
        self.buffer_size, self.n_agents, self.T = buffer_size, n_agents, max_episode_length
        self.pos = self.cur_pos = self.t = 0
        T = max_episode_length
        self.observations = [np.zeros((T + 1, buffer_size, obs_dim), np.float32) for _ in range(n_agents)]
        self.actions = np.zeros((n_agents, T, buffer_size), np.int64)
        self.rewards = np.zeros((n_agents, T, buffer_size), np.float32)
        self.dones = np.zeros((T + 1, buffer_size), bool)
        self.filled = np.zeros((T, buffer_size), bool)

    def __len__(self):
        return min(self.pos, self.buffer_size)

    def init_episode(self, obss):
        self.t = 0
        for i in range(self.n_agents):
            self.observations[i][0, self.cur_pos] = obss[i]

    def add(self, obss, acts, rews, done):
        assert self.t < self.T, "Episode longer than given max length!"
        for i in range(self.n_agents):
            self.observations[i][self.t + 1, self.cur_pos] = obss[i]
        self.actions[:, self.t, self.cur_pos] = acts
        self.rewards[:, self.t, self.cur_pos] = rews
        self.dones[self.t + 1, self.cur_pos] = done
        self.filled[self.t, self.cur_pos] = True
        self.t += 1
        if done:
            self.pos += 1
            self.cur_pos = self.pos % self.buffer_size
            self.t = 0

    def can_sample(self, batch_size):
        return self.pos >= batch_size

    def sample_idx(self, idx):
        return dict(
            obss=torch.stack([torch.tensor(self.observations[i][:, idx]) for i in range(self.n_agents)]),
            actions=torch.tensor(self.actions[:, :, idx], dtype=torch.int64),
            rewards=torch.tensor(self.rewards[:, :, idx], dtype=torch.float32),
            dones=torch.tensor(self.dones[:, idx], dtype=torch.float32),
            filled=torch.tensor(self.filled[:, idx], dtype=torch.float32),
        )

    def sample(self, batch_size):
        return self.sample_idx(np.random.randint(0, len(self), size=batch_size))


def epsilon_schedule(decay_style, decay_over, eps_start, eps_end, exp_decay_rate, total_steps):
    """marlbase/dqn/train.py:127-174"""
    assert decay_style in ["linear", "lin", "exponential", "exp"]
    assert 0 <= eps_start <= 1 and 0 <= eps_end <= 1 and eps_start >= eps_end
    assert 0 < decay_over <= 1 and total_steps > 0 and exp_decay_rate > 0
    if decay_style in ["linear", "lin"]:
        return lambda s: max(eps_end + (eps_start - eps_end) * (1 - s / (total_steps * decay_over)), eps_end)
    eps_decay = (eps_start - eps_end) / (total_steps * decay_over) * exp_decay_rate
    return lambda s: max(eps_end + (eps_start - eps_end) * math.exp(-eps_decay * s), eps_end)


def synthetic_batch(P, T, B, D, A, seed=0):
    """Replay-shaped random batch: integer-valued observations like LBF's, ragged episode
    lengths (filled prefix, done at the last filled step)."""
    g = torch.Generator().manual_seed(seed)
    obss = torch.randint(-1, 8, (P, T + 1, B, D), generator=g).float()
    actions = torch.randint(0, A, (P, T, B), generator=g)
    rewards = torch.rand(P, T, B, generator=g) * (torch.rand(P, T, B, generator=g) < 0.2)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    tt = torch.arange(T).unsqueeze(1)
    filled = (tt < lens.unsqueeze(0)).float()
    dones = torch.zeros(T + 1, B)
    dones[lens, torch.arange(B)] = 1.0
    return dict(obss=obss, actions=actions, rewards=rewards.float(), dones=dones, filled=filled)
