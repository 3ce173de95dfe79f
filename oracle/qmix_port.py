"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's QMIX learner.

Follows marlbase/dqn/model.py:
  QMixer.__init__  :272-312  (hypernet_layers == 2: Linear-ReLU-Linear hypernets, hyper_b_1, V)
  QMixer.forward   :313-331  (w1 = |hyper_w_1(s)|, hidden = elu(q.w1 + b1), y = hidden.|hyper_w_final(s)| + V(s))
  QMixNetwork._compute_loss :374-427 (state = concat of all agents' observations, reward of agent 0,
                                      Double-Q bootstrap through the TARGET mixer on obs[1:])
  QNetwork.update  :165-174  (clip_grad_norm_ over the CRITIC parameters only - the mixer's gradient is
                              not clipped and does not enter the norm; one Adam over critic + mixer)
  QMixNetwork.soft_update / hard_update :429-443
Pinned by tests/golden/learner_qmix_*.npz, generated from the reference's own QMixNetwork
(oracle/make_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.

Flat mixer parameter block = torch `mixer.parameters()` order:
  hyper_w_1.0.{weight[HE,SD],bias[HE]}  hyper_w_1.2.{weight[E*P,HE],bias[E*P]}
  hyper_w_final.0.{weight[HE,SD],bias[HE]}  hyper_w_final.2.{weight[E,HE],bias[E]}
  hyper_b_1.{weight[E,SD],bias[E]}  V.0.{weight[E,SD],bias[E]}  V.2.{weight[1,E],bias[1]}
"""
import torch
import torch.nn.functional as F

from . import dqn_port as dp


def mixer_shapes(P, SD, E=64, HE=32, L=2):
    """L = hypernet_layers (model.py:283-301): 1 -> hyper_w_1 / hyper_w_final are one Linear each on the state"""
    if L == 1:
        return [(E * P, SD), (E * P,), (E, SD), (E,), (E, SD), (E,), (E, SD), (E,), (1, E), (1,)]
    return [(HE, SD), (HE,), (E * P, HE), (E * P,), (HE, SD), (HE,), (E, HE), (E,), (E, SD), (E,), (E, SD), (E,), (1, E), (1,)]


def mixer_nparams(P, SD, E=64, HE=32, L=2):
    n = 0
    for s in mixer_shapes(P, SD, E, HE, L):
        k = 1
        for d in s:
            k *= d
        n += k
    return n


def mixer_split(flat, P, SD, E=64, HE=32, L=2):
    out, o = [], 0
    for s in mixer_shapes(P, SD, E, HE, L):
        k = 1
        for d in s:
            k *= d
        out.append(flat[o:o + k].reshape(s))
        o += k
    return out


def mixer_init(P, SD, E=64, HE=32, seed=0, L=2):
    """torch default nn.Linear init, modules built in QMixer.__init__'s order (model.py:283-312)."""
    torch.manual_seed(seed)
    if L == 1:
        mods = [torch.nn.Linear(SD, E * P), torch.nn.Linear(SD, E)]
    else:
        mods = [torch.nn.Linear(SD, HE), torch.nn.Linear(HE, E * P), torch.nn.Linear(SD, HE), torch.nn.Linear(HE, E)]
    mods += [torch.nn.Linear(SD, E), torch.nn.Linear(SD, E), torch.nn.Linear(E, 1)]
    return torch.cat([p.detach().reshape(-1) for m in mods for p in m.parameters()])


def mixer_forward(flat, agent_qs, states, P, E=64, HE=32, L=2):
    """QMixer.forward: agent_qs [P,T,B], states [T,B,SD] -> [T,B]."""
    T, B = agent_qs.shape[1:]
    SD = states.shape[-1]
    qs = agent_qs.permute(1, 2, 0).reshape(T * B, 1, P)
    s = states.reshape(-1, SD)
    if L == 1:
        W1, w1b, Wf, wfb, Bb, cb, Av, av, bv, cv = mixer_split(flat, P, SD, E, HE, 1)
        w1 = torch.abs(F.linear(s, W1, w1b)).view(-1, P, E)
        wf_pre = F.linear(s, Wf, wfb)
    else:
        A1, a1, B1, c1, Af, af, Bf, cf, Bb, cb, Av, av, bv, cv = mixer_split(flat, P, SD, E, HE)
        w1 = torch.abs(F.linear(torch.relu(F.linear(s, A1, a1)), B1, c1)).view(-1, P, E)
        wf_pre = F.linear(torch.relu(F.linear(s, Af, af)), Bf, cf)
    b1 = F.linear(s, Bb, cb).view(-1, 1, E)
    hidden = F.elu(torch.bmm(qs, w1) + b1)
    wf = torch.abs(wf_pre).view(-1, E, 1)
    v = F.linear(torch.relu(F.linear(s, Av, av)), bv, cv).view(-1, 1, 1)
    return (torch.bmm(hidden, wf) + v).view(T, B)


def compute_loss(params, tparams, mixer, tmixer, batch, gamma, double_q, D, H, A, E=64, HE=32, L=2):
    """QMixNetwork._compute_loss (model.py:374-427), standardise_returns False."""
    obss, actions = batch["obss"], batch["actions"].unsqueeze(-1)
    rewards, dones, filled = batch["rewards"][0], batch["dones"][1:], batch["filled"]
    P = obss.shape[0]
    q = dp.q_values(params, obss, D, H, A)
    chosen = mixer_forward(mixer, q[:, :-1].gather(-1, actions).squeeze(-1), torch.concat(list(obss[:, :-1]), dim=-1), P, E, HE, L)
    with torch.no_grad():
        tq = dp.q_values(tparams, obss, D, H, A)[:, 1:]
        if double_q:
            a_prime = q.detach()[:, 1:].argmax(-1)
            target_qs = tq.gather(-1, a_prime.unsqueeze(-1)).squeeze(-1)
        else:
            target_qs, _ = tq.max(dim=-1)
        target_tot = mixer_forward(tmixer, target_qs, torch.concat(list(obss[:, 1:]), dim=-1), P, E, HE, L)
    returns = rewards + gamma * target_tot * (1 - dones)
    loss = F.mse_loss(chosen, returns.detach(), reduction="none")
    return (loss * filled).sum() / filled.sum()


class Learner:
    """QMixNetwork.update: one Adam over critic + mixer tensors, clip over the critic tensors only."""

    def __init__(self, params, mixer, D, H, A, E=64, HE=32, lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True,
                 target_update_interval_or_tau=200, L=2):
        self.D, self.H, self.A, self.E, self.HE, self.L = D, H, A, E, HE, L
        self.P = P = params.shape[0]
        self.SD = P * D
        self.tensors = [torch.nn.Parameter(t.clone()) for p in range(P) for t in dp.split(params[p], D, H, A)]
        self.mtensors = [torch.nn.Parameter(t.clone()) for t in mixer_split(mixer, P, self.SD, E, HE, L)]
        self.target, self.tmixer = params.clone(), mixer.clone()
        self.opt = torch.optim.Adam(self.tensors + self.mtensors, lr=lr)
        self.gamma, self.grad_clip, self.double_q = gamma, grad_clip, double_q
        self.tui = target_update_interval_or_tau
        self.updates = self.last_target_update = 0

    def flat(self):
        per = len(self.tensors) // self.P
        return torch.stack([torch.cat([t.reshape(-1) for t in self.tensors[p * per:(p + 1) * per]]) for p in range(self.P)])

    def mflat(self):
        return torch.cat([t.reshape(-1) for t in self.mtensors])

    def update(self, batch, chunks=1):
        """chunks > 1: the gradient accumulated over column chunks of the batch (dqn_port.Learner.update: bounded memory at size)"""
        self.opt.zero_grad()
        if chunks > 1:
            total, loss = batch["filled"].sum(), 0.0
            for cols in torch.arange(batch["filled"].shape[1]).chunk(chunks):
                sub = dp.column_chunk(batch, cols)
                part = compute_loss(self.flat(), self.target, self.mflat(), self.tmixer, sub, self.gamma, self.double_q,
                                    self.D, self.H, self.A, self.E, self.HE, self.L) * (sub["filled"].sum() / total)
                part.backward()
                loss = loss + part.detach()
        else:
            loss = compute_loss(self.flat(), self.target, self.mflat(), self.tmixer, batch, self.gamma, self.double_q,
                                self.D, self.H, self.A, self.E, self.HE, self.L)
            loss.backward()
        per = len(self.tensors) // self.P  # the gradients as _compute_loss leaves them (before clipping): for tests that compare them
        self.last_grad = torch.stack([torch.cat([t.grad.reshape(-1) for t in self.tensors[p * per:(p + 1) * per]]) for p in range(self.P)]).clone()
        self.last_mixer_grad = torch.cat([t.grad.reshape(-1) for t in self.mtensors]).clone()
        if self.grad_clip:
            torch.nn.utils.clip_grad_norm_(self.tensors, self.grad_clip)
        self.opt.step()
        self.updates += 1
        if self.tui > 1.0 and (self.updates - self.last_target_update) >= self.tui:
            self.target, self.tmixer = self.flat().detach().clone(), self.mflat().detach().clone()
            self.last_target_update = self.updates
        elif self.tui < 1.0:
            self.target = (1 - self.tui) * self.target + self.tui * self.flat().detach()
            self.tmixer = (1 - self.tui) * self.tmixer + self.tui * self.mflat().detach()
        return {"loss": loss.item()}
