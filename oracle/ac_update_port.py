"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's actor-critic learner step.

Follows marlbase/ac/model.py (independent actors / critics, no RNN, no action masks, standardise_returns False):
  A2CNetwork.update         :189-246  target-critic values on all T+1 observations, n-step returns, critic values and
                                      actor distributions on obs[:-1], policy-gradient + entropy + value loss, filled-masked
                                      means, clip_grad_norm_ over ALL parameters, Adam, target update keyed on the ENV STEP
                                      (`step % interval == 0` -> hard copy; interval < 1 -> Polyak)
  PPONetwork.update         :264-352  old log-probs once, num_epochs x (clipped surrogate, same value loss, Adam step)
  compute_nstep_returns     marlbase/utils/utils.py:38-63 (note the indexing: reward t+k is masked with done[t+k], the
                                      bootstrap uses next_values[t+n] and done[t+n], and is dropped when t+n >= T)
Batch layout = marlbase/ac/train.py:14-16,36-49: obss [T+1,N,P*D] (agents concatenated), actions i64 [T,N,P],
rewards [T,N,P], dones bool [T+1,N], filled f32 [T,N].
Pinned by tests/golden/learner_a2c_*.npz / learner_ppo_*.npz (oracle/make_golden_ac.py, from the reference's classes).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
import torch

from . import dqn_port as dp


def nstep_returns(rewards, done, next_values, nsteps, gamma):
    T = rewards.shape[0]
    out = torch.zeros_like(rewards)
    for t0 in range(T):
        acc = torch.zeros_like(rewards[0])
        for k in range(nsteps + 1):
            t = t0 + k
            if t >= T:
                break
            if k == nsteps:
                acc = acc + gamma ** k * next_values[t] * (1 - done[t])
            else:
                acc = acc + gamma ** k * rewards[t] * (1 - done[t])
        out[t0] = acc
    return out


def _split_obs(obss, P):
    D = obss.shape[-1] // P
    return [obss[..., p * D:(p + 1) * D] for p in range(P)]


def values(block, obss, D, H):
    """[P][n] value-net blocks, obss [..., P*D] -> [..., P]; a block sized for P*D inputs is a centralised critic
    (critic.centralised, ac/model.py:62-66,156-157): every agent's critic reads the whole concatenated row."""
    P = block.shape[0]
    if block.shape[1] == dp.nparams(P * D, H, 1) and P > 1:
        return torch.cat([dp.mlp(block[p], obss, P * D, H, 1) for p in range(P)], dim=-1)
    xs = _split_obs(obss, P)
    return torch.cat([dp.mlp(block[p], xs[p], D, H, 1) for p in range(P)], dim=-1)


def logits(block, obss, D, H, A):
    P = block.shape[0]
    xs = _split_obs(obss, P)
    return [dp.mlp(block[p], xs[p], D, H, A) for p in range(P)]


def _masked(lg, batch):
    """get_dist (ac/model.py:135-145) with batch["action_masks"][:-1] ([T+1][N][P][A]) when present"""
    m = batch.get("action_masks")
    if m is None:
        return lg
    return [l * m[:-1, :, p] + (1 - m[:-1, :, p]) * -1e8 for p, l in enumerate(lg)]


def evaluate(actor, critic, target, batch, D, H, A, n_steps, gamma, ret_ms=None, Hc=None):
    """Hc: the critics' own `layers` list when it differs from the actors' H (ac/model.py:45-97 builds each family from its own)"""
    Hc = H if Hc is None else Hc
    obss, actions = batch["obss"], batch["actions"]
    P = actor.shape[0]
    with torch.no_grad():
        next_value = values(target, obss, D, Hc)
    if ret_ms is not None:  # standardise_returns (model.py:195-196)
        next_value = next_value * torch.sqrt(ret_ms.var) + ret_ms.mean
    done = batch["dones"].float().unsqueeze(-1).repeat(1, 1, P)
    returns = nstep_returns(batch["rewards"], done, next_value, n_steps, gamma)
    if ret_ms is not None:  # model.py:202-204
        ret_ms.update(returns)
        returns = (returns - ret_ms.mean) / torch.sqrt(ret_ms.var)
    v = values(critic, obss[:-1], D, Hc)
    lg = _masked(logits(actor, obss[:-1], D, H, A), batch)
    dists = [torch.distributions.Categorical(logits=l) for l in lg]
    logp = torch.stack([d.log_prob(actions[..., p]) for p, d in enumerate(dists)], dim=-1)
    ent = torch.stack([d.entropy() for d in dists], dim=-1).sum(-1)
    return returns, v, logp, ent


def a2c_loss(actor, critic, target, batch, D, H, A, n_steps=5, gamma=0.99, entropy_coef=0.001, value_loss_coef=0.5, ret_ms=None, Hc=None):
    returns, v, logp, ent = evaluate(actor, critic, target, batch, D, H, A, n_steps, gamma, ret_ms, Hc)
    filled = batch["filled"]
    adv = returns - v
    actor_loss = -(logp * adv.detach()).sum(-1) - entropy_coef * ent
    actor_loss = (actor_loss * filled).sum() / filled.sum()
    value_loss = ((returns - v).pow(2).sum(-1) * filled).sum() / filled.sum()
    loss = actor_loss + value_loss_coef * value_loss
    return loss, {"loss": loss, "actor_loss": actor_loss, "value_loss": value_loss, "entropy": (ent * filled).sum() / filled.sum()}


def ppo_loss(actor, critic, returns, old_logp, batch, D, H, A, entropy_coef, value_loss_coef, ppo_clip, Hc=None):
    obss, actions, filled = batch["obss"], batch["actions"], batch["filled"]
    v = values(critic, obss[:-1], D, H if Hc is None else Hc)
    lg = _masked(logits(actor, obss[:-1], D, H, A), batch)
    dists = [torch.distributions.Categorical(logits=l) for l in lg]
    logp = torch.stack([d.log_prob(actions[..., p]) for p, d in enumerate(dists)], dim=-1)
    ent = torch.stack([d.entropy() for d in dists], dim=-1).sum(-1)
    adv = returns - v
    ratio = torch.exp(logp - old_logp)
    s1 = ratio * adv.detach()
    s2 = torch.clamp(ratio, 1.0 - ppo_clip, 1.0 + ppo_clip) * adv.detach()
    actor_loss = ((-torch.min(s1, s2).sum(-1) - entropy_coef * ent) * filled).sum() / filled.sum()
    value_loss = (adv.pow(2).sum(-1) * filled).sum() / filled.sum()
    loss = actor_loss + value_loss_coef * value_loss
    return loss, {"loss": loss, "actor_loss": actor_loss, "value_loss": value_loss, "entropy": (ent * filled).sum() / filled.sum()}


class Learner:
    """A2CNetwork / PPONetwork update: one Adam over actor + critic tensors in parameters() order."""

    def __init__(self, actor, critic, D, H, A, lr=3e-4, gamma=0.99, n_steps=5, entropy_coef=0.001, value_loss_coef=0.5,
                 grad_clip=False, target_update_interval_or_tau=200, num_epochs=0, ppo_clip=0.2, standardise_returns=False, Hc=None):
        self.ret_ms = dp.RunningMeanStd((actor.shape[0],)) if standardise_returns else None
        self.D, self.H, self.A, self.P = D, H, A, actor.shape[0]
        self.Hc = Hc = H if Hc is None else Hc  # the critics' own layer list (actor.layers != critic.layers)
        self.at = [torch.nn.Parameter(t.clone()) for p in range(self.P) for t in dp.split(actor[p], D, H, A)]
        dc = self.P * D if critic.shape[1] == dp.nparams(self.P * D, Hc, 1) and self.P > 1 else D  # centralised critic
        self.ct = [torch.nn.Parameter(t.clone()) for p in range(self.P) for t in dp.split(critic[p], dc, Hc, 1)]
        self.target = critic.clone()
        self.opt = torch.optim.Adam(self.at + self.ct, lr=lr)
        self.gamma, self.n_steps, self.ec, self.vc = gamma, n_steps, entropy_coef, value_loss_coef
        self.grad_clip, self.tui, self.num_epochs, self.ppo_clip = grad_clip, target_update_interval_or_tau, num_epochs, ppo_clip

    def _flat(self, ts):
        per = len(ts) // self.P
        return torch.stack([torch.cat([t.reshape(-1) for t in ts[p * per:(p + 1) * per]]) for p in range(self.P)])

    def actor(self):
        return self._flat(self.at)

    def critic(self):
        return self._flat(self.ct)

    def _step(self, loss):
        self.opt.zero_grad()
        loss.backward()
        if self.grad_clip:
            torch.nn.utils.clip_grad_norm_(self.at + self.ct, self.grad_clip)
        self.opt.step()

    def update(self, batch, step):
        D, H, A = self.D, self.H, self.A
        if self.num_epochs == 0:
            loss, m = a2c_loss(self.actor(), self.critic(), self.target, batch, D, H, A, self.n_steps, self.gamma, self.ec, self.vc,
                               self.ret_ms, self.Hc)
            self._step(loss)
            metrics = {k: v.item() for k, v in m.items()}
        else:
            with torch.no_grad():
                returns, _, old_logp, _ = evaluate(self.actor(), self.critic(), self.target, batch, D, H, A, self.n_steps, self.gamma,
                                                   self.ret_ms, self.Hc)
            acc = {}
            for _ in range(self.num_epochs):
                loss, m = ppo_loss(self.actor(), self.critic(), returns, old_logp, batch, D, H, A, self.ec, self.vc, self.ppo_clip, self.Hc)
                self._step(loss)
                for k, v in m.items():
                    acc.setdefault(k, []).append(v.item())
            metrics = {k: sum(v) / len(v) for k, v in acc.items()}
        if self.tui > 1.0 and step % self.tui == 0:
            self.target = self.critic().detach().clone()
        elif self.tui < 1.0:
            self.target = (1 - self.tui) * self.target + self.tui * self.critic().detach()
        return metrics


def synthetic_batch(P, T, N, D, A, seed=0):
    """rollout-shaped Batch: integer-valued observations, episode lengths in [2, T] (some run the full T: no done flag)"""
    g = torch.Generator().manual_seed(seed)
    obss = torch.randint(-1, 8, (T + 1, N, P * D), generator=g).float() * 0.25
    actions = torch.randint(0, A, (T, N, P), generator=g)
    rewards = torch.rand(T, N, P, generator=g) * (torch.rand(T, N, P, generator=g) < 0.3)
    lens = torch.randint(2, T + 4, (N,), generator=g).clamp(max=T)
    ended = torch.rand(N, generator=g) < 0.8
    dones = torch.zeros(T + 1, N, dtype=torch.bool)
    filled = torch.zeros(T, N)
    for i in range(N):
        L = int(lens[i])
        filled[:L, i] = 1
        if L < T or ended[i]:
            dones[L, i] = True
        obss[L + 1:, i] = 0
        actions[L:, i] = 0
        rewards[L:, i] = 0
    return dict(obss=obss, actions=actions, rewards=rewards, dones=dones, filled=filled)
