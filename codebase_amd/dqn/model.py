"""`QNetwork` with the reference's duck-typed model interface - drop-in for
`algorithm.model._target_: dqn.model.QNetwork` (marlbase/configs/algorithm/idqn.yaml:7,
marlbase/dqn/model.py:14-196) - whose every computation runs in libmarlhip.so.

Interface kept: constructor arguments, `init_hiddens`, `act(inputs, hiddens, epsilon,
action_masks=None) -> (list[int], hiddens)`, `update(batch) -> {"loss": float}`, `update_target /
hard_update / soft_update`, `parameters()`, `state_dict()/load_state_dict()` with the reference's key
names (`critic.independent.{i}.network.{0,2,4}.{weight,bias}`, `target.*`) so checkpoints written by
either implementation load in the other (marlbase/dqn/train.py:340-343, marlbase/eval.py:42-60).

Storage: one flat fp32 block per agent, `params[P][nparams]`, in torch parameters() order; the
state_dict tensors are slices of it.
"""
import math

import numpy as np
import random
from collections import OrderedDict

import torch
from torch import nn

from .. import hip as _hip
from ..spaces import flatdim


def _fc(dims, use_orthogonal_init):
    """Linear-ReLU-...-Linear exactly as FCNetwork builds it (utils/models.py:34-42), used only to
    draw the initial weights from torch's RNG in the reference's order."""
    mods = []
    for i in range(len(dims) - 1):
        lin = nn.Linear(dims[i], dims[i + 1])
        if use_orthogonal_init:
            nn.init.orthogonal_(lin.weight.data, gain=np.sqrt(2))  # np.float64 gain, as utils/models.py:9
            nn.init.constant_(lin.bias.data, 0)
        mods.append(lin)
    return mods


def sharing_indices(parameter_sharing, n_agents):
    """None for independent networks, else the agent -> network map of MultiAgentSharedNetwork
    (marlbase/utils/models.py:193-202): True -> all agents on network 0; a list -> SePS indices."""
    if parameter_sharing is False or parameter_sharing is None:
        return None
    idx = [0] * n_agents if parameter_sharing is True else [int(i) for i in parameter_sharing]
    if len(idx) != n_agents:
        raise ValueError("Expect same number of sharing indices as agents")
    seen = []
    for i in idx:
        if i not in seen:
            seen.append(i)
    if seen != list(range(len(seen))):  # the reference indexes its network list with these values (models.py:286-295)
        raise ValueError(f"sharing indices {idx}: networks must be numbered 0..K-1 in order of first appearance")
    return tuple(idx)


def init_flat_params(obs_dims, hidden_dims, act_dims, use_orthogonal_init=True, sharing=None):
    """Initial critic AND target blocks, consuming torch's global RNG like the reference constructor
    does (critic nets for agents 0..P-1 - or one per shared network in order of first appearance,
    utils/models.py:209-240 - then target nets; dqn/model.py:36-41) before hard_update
    overwrites the target.  Returns (critic[K][n], target[K][n]) on the CPU."""
    if sharing is not None:
        first = [sharing.index(k) for k in range(max(sharing) + 1)]
        obs_dims, act_dims = [obs_dims[i] for i in first], [act_dims[i] for i in first]
    blocks = []
    for _ in range(2):
        per_agent = []
        for d, a in zip(obs_dims, act_dims):
            lins = _fc([d] + list(hidden_dims) + [a], use_orthogonal_init)
            per_agent.append(torch.cat([t.detach().reshape(-1) for lin in lins for t in (lin.weight, lin.bias)]))
        blocks.append(torch.stack(per_agent))
    critic = blocks[0]
    return critic, critic.clone()  # hard_update (dqn/model.py:60)


def _gru_layout(D, H, A, L=1):
    """RNNNetwork's parameters() order (utils/models.py:83-92): nn.GRU lists layer l's four tensors behind layer l - 1's"""
    out = [("first_layer.weight", (H, D)), ("first_layer.bias", (H,))]
    for l in range(L):
        out += [(f"rnn.weight_ih_l{l}", (3 * H, H)), (f"rnn.weight_hh_l{l}", (3 * H, H)), (f"rnn.bias_ih_l{l}", (3 * H,)), (f"rnn.bias_hh_l{l}", (3 * H,))]
    return out + [("final_layer.weight", (A, H)), ("final_layer.bias", (A,))]


def init_flat_gru_params(obs_dims, hidden, act_dims, use_orthogonal_init=True, sharing=None, sets=2, num_layers=1):
    """Initial critic / target blocks of recurrent networks, consuming torch's global RNG like RNNNetwork.__init__ does
    (utils/models.py:83-94: nn.Linear and nn.GRU default inits, orthogonal gain sqrt(2) + zero bias on the output layer only),
    critic nets first, then target nets, then hard_update (dqn/model.py:36-41,60); with parameter sharing one network per distinct
    index in order of first appearance (utils/models.py:209-240)."""
    if sharing is not None:
        first = [sharing.index(k) for k in range(max(sharing) + 1)]
        obs_dims, act_dims = [obs_dims[i] for i in first], [act_dims[i] for i in first]
    blocks = []
    for _ in range(sets):
        per_agent = []
        for d, a in zip(obs_dims, act_dims):
            first = nn.Linear(d, hidden)
            rnn = nn.GRU(input_size=hidden, hidden_size=hidden, num_layers=num_layers, batch_first=False)
            final = nn.Linear(hidden, a)
            if use_orthogonal_init:
                nn.init.orthogonal_(final.weight.data, gain=np.sqrt(2))
                nn.init.constant_(final.bias.data, 0)
            per_agent.append(torch.cat([t.detach().reshape(-1) for t in [first.weight, first.bias] + list(rnn.parameters()) + [final.weight, final.bias]]))
        blocks.append(torch.stack(per_agent))
    return blocks[0], blocks[0].clone()


# ---- recurrent networks of widths other than the compiled 64 / 128 ------------------------------------------------------------
# RNNNetwork(dims=[D, h, h, A]) (utils/models.py:51-116) = Linear(D, h) - ReLU - one-layer GRU(h, h) - Linear(h, A) for ANY h.  A width
# h <= 128 runs EXACTLY on the kernel width Hk in {64, 128} by zero padding, like the feed-forward lists above: a padded unit has zero
# first-layer / gate weights and biases, so r = z = 1/2, n = tanh(0) = 0 and h' = (1 - z) n + z h = h / 2 = 0 from the zero initial
# state on; every gradient into the padding is a product with one of those zeros (zero columns of W3 / W_hh keep dL/dh of a padded unit
# at 0), and Adam maps zero gradient + zero moments to a zero step.  The gate matrices are [3 Hk][Hk] with one [Hk][Hk] block per gate
# (r, z, n): the live part is [:, :h, :h] of the (3, Hk, Hk) view.  Longer lists [h] * (L + 1) stack L GRU layers (csrc/gru_stack.h; a padded
# unit of layer l feeds zeros into layer l + 1, whose padded input columns are zero as well); h > 128 raises.
MAX_GRU_LAYERS = 4  # csrc/gru_stack.h: GRU_MAX_LAYERS


def recurrent_width(hidden):
    """`layers` of a use_rnn network -> (h, Hk): the live width and the kernel width it is padded to, or raise.  len(layers) - 1 GRU
    layers (recurrent_depth): RNNNetwork asserts equal sizes (utils/models.py:79-81) and builds nn.GRU(num_layers=len(layers) - 1)"""
    if len(hidden) < 2 or len(set(hidden)) != 1:
        raise NotImplementedError(f"use_rnn with layers={list(hidden)}: at least two equal sizes (RNNNetwork itself asserts equal sizes, "
                                  "utils/models.py:79-81; a single size would be nn.GRU(num_layers=0), which torch rejects)")
    if len(hidden) - 1 > MAX_GRU_LAYERS:
        raise NotImplementedError(f"use_rnn with layers={list(hidden)}: up to {MAX_GRU_LAYERS} stacked GRU layers (layers = [h] * 2 .. [h] * {MAX_GRU_LAYERS + 1})")
    h = int(hidden[0])
    if not 1 <= h <= 128:
        raise NotImplementedError(f"use_rnn with layers={list(hidden)}: recurrent widths up to 128 (zero-padded onto the 64 / 128 kernels)")
    return h, (64 if h <= 64 else 128)


def recurrent_depth(hidden):
    """number of stacked GRU layers of a use_rnn network: len(layers) - 1 (utils/models.py:77)"""
    return len(hidden) - 1


def gru_block_views(row, D, h, A, H, L=1):
    """(name, live view, reference shape) of one flat recurrent block laid out for kernel width H (H == h: the unpadded layout), in
    RNNNetwork's parameters() order (utils/models.py:83-92); L stacked GRU layers"""
    o, out = 0, []

    def take(n):
        nonlocal o
        v = row[o:o + n]
        o += n
        return v

    out.append(("first_layer.weight", take(H * D).view(H, D)[:h], (h, D)))
    out.append(("first_layer.bias", take(H)[:h], (h,)))
    for l in range(L):
        out.append((f"rnn.weight_ih_l{l}", take(3 * H * H).view(3, H, H)[:, :h, :h], (3 * h, h)))
        out.append((f"rnn.weight_hh_l{l}", take(3 * H * H).view(3, H, H)[:, :h, :h], (3 * h, h)))
        out.append((f"rnn.bias_ih_l{l}", take(3 * H).view(3, H)[:, :h], (3 * h,)))
        out.append((f"rnn.bias_hh_l{l}", take(3 * H).view(3, H)[:, :h], (3 * h,)))
    out.append(("final_layer.weight", take(A * H).view(A, H)[:, :h], (A, h)))
    out.append(("final_layer.bias", take(A), (A,)))
    assert o == row.numel(), (o, row.numel())
    return out


def pad_gru_blocks(flat, D, h, A, H, L=1):
    """[K][n(h)] recurrent parameter blocks -> [K][n(H)] zero-padded blocks"""
    if h == H:
        return flat
    n = H * D + H + L * (6 * H * H + 6 * H) + A * H + A
    out = torch.zeros(flat.shape[0], n, dtype=flat.dtype)
    for k in range(flat.shape[0]):
        for (_, dst, _), (_, src, _) in zip(gru_block_views(out[k], D, h, A, H, L), gru_block_views(flat[k], D, h, A, h, L)):
            dst.copy_(src)
    return out


def _tensor_layout(D, H, A):
    return [("network.0.weight", (H, D)), ("network.0.bias", (H,)), ("network.2.weight", (H, H)),
            ("network.2.bias", (H,)), ("network.4.weight", (A, H)), ("network.4.bias", (A,))]


# ---- hidden widths other than the compiled ones -------------------------------------------------------------------------------
# The fused kernels are instantiated for two equal hidden layers of 64 or 128 units; wider layers (up to 1024, e.g. [256, 256]) run on
# the GEMM path (csrc/wide_mlp.h, marlhip_wide_*: slower, any width) at max(h1, h2) rounded up to 16.  A two-layer list [h1, h2] runs
# EXACTLY on the kernel width H >= h1, h2 by zero padding: a padded unit has zero weights and bias, so its activation is relu(0) = 0,
# its relu mask is 0, every gradient that touches a padded entry is a product with one of those zeros, and Adam maps a zero gradient
# with zero moments to a zero step - the padding stays zero for the whole run, the live sub-network computes what FCNetwork([h1, h2])
# computes (utils/models.py:34-48), and state_dict exposes the live tensors in the reference's shapes.
def compiled_width(hidden):
    """hidden layer list -> the width the kernels run it at (<= 128 with two layers: a fused kernel; otherwise the GEMM path), or raise"""
    if not 1 <= len(hidden) <= 16:  # WideNet::MAXL (csrc/wide_mlp.h); FCNetwork itself takes any list (utils/models.py:14-48)
        raise NotImplementedError(f"layers={list(hidden)}: 1 to 16 hidden layers")
    h = max(hidden)
    if min(hidden) < 1 or h > 1024:
        raise NotImplementedError(f"layers={list(hidden)}: hidden widths 1..1024")
    if len(hidden) == 2 and h <= 128:
        return 64 if h <= 64 else 128
    return max((h + 15) // 16 * 16, 144 if len(hidden) == 2 else 16)  # the GEMM path (NetSpec.wide): any width, any of 1..16 layers


def is_wide(hidden):
    return len(hidden) != 2 or max(hidden) > 128


def pad_blocks(flat, D, hidden, A, H):
    """[K][n(hidden)] parameter blocks in FCNetwork's parameters() order -> [K][n(H, ..., H)] zero-padded blocks"""
    if all(h == H for h in hidden):
        return flat
    K = flat.shape[0]
    n = sum(o * i + o for o, i in zip([H] * len(hidden) + [A], [D] + [H] * len(hidden)))
    out = torch.zeros(K, n, dtype=flat.dtype)
    for k in range(K):
        for (_, dst), (_, src) in zip(block_views(out[k], D, hidden, A, H), block_views(flat[k], D, hidden, A, None)):
            dst.copy_(src)
    return out


def block_views(row, D, hidden, A, H):
    """(name, view) of the LIVE tensors inside one flat block laid out for width H (H None: the unpadded layout)"""
    live_out = list(hidden) + [A]
    live_in = [D] + list(hidden)
    full_out = live_out if H is None else [H] * len(hidden) + [A]
    full_in = live_in if H is None else [D] + [H] * len(hidden)
    o, out = 0, []
    for k, (fo, fi, lo, li) in enumerate(zip(full_out, full_in, live_out, live_in)):
        out.append((f"network.{2 * k}.weight", row[o:o + fo * fi].view(fo, fi)[:lo, :li]))
        o += fo * fi
        out.append((f"network.{2 * k}.bias", row[o:o + fo][:lo]))
        o += fo
    return out


class QNetwork:
    def __init__(self, obs_space, action_space, cfg, layers, parameter_sharing=False, use_rnn=False,
                 use_orthogonal_init=True, device="cuda"):
        hidden = [int(h) for h in layers]
        obs_dims = [flatdim(o) for o in obs_space]
        act_dims = [flatdim(a) for a in action_space]
        self.recurrent = bool(use_rnn)
        self.live_hidden = tuple(hidden)
        Hk = recurrent_width(hidden)[1] if use_rnn else compiled_width(hidden)  # the width the kernels run at (the layers zero-padded to it, see pad_blocks / pad_gru_blocks)
        hidden_k = [Hk] * len(hidden)
        if len(set(obs_dims)) != 1 or len(set(act_dims)) != 1:
            # (the reference's own learner cannot train such agents either: ReplayBuffer.sample stacks the agents' observation arrays,
            # dqn/train.py:98, and _compute_loss stacks their value tensors, dqn/model.py:128; the actor-critic classes take them)
            raise NotImplementedError("agents with different observation / action sizes: the DQN learners stack the agents' tensors (as the "
                                      "reference's do, dqn/model.py:128); codebase_amd.ac.model takes such agents")
        if str(device) == "cpu":
            raise _hip.MarlHipError("codebase_amd.dqn.model.QNetwork runs on the GPU only: set algorithm.model.device=cuda")
        get = (lambda k, d=None: cfg[k] if k in cfg else d) if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
        self.optimizer = get("optimizer", "Adam")  # getattr(optim, cfg.optimizer) (dqn/model.py:66-71): Adam, SGD, RMSprop, AdamW are built
        _hip.optimizer_id(self.optimizer)
        self.standardise_returns = bool(get("standardise_returns", False))
        self.action_space = action_space
        self.n_agents = len(obs_dims)
        self.device = torch.device(device)
        self.sharing = sharing_indices(parameter_sharing, self.n_agents)
        self.spec = _hip.NetSpec(self.n_agents, obs_dims[0], hidden_k[0], act_dims[0], self.sharing, wide=is_wide(hidden) and not use_rnn, n_hidden=len(hidden))
        if self.recurrent:  # RNNNetwork (utils/models.py:51-116): Linear -> ReLU -> GRU -> Linear
            self.nparams = _hip.gru_nparams(self.spec)
            self.rnn_layers = recurrent_depth(hidden)
            critic, target = init_flat_gru_params(obs_dims, hidden[0], act_dims, use_orthogonal_init, self.sharing, num_layers=self.rnn_layers)  # the reference's draws at the true width
            critic = pad_gru_blocks(critic, obs_dims[0], hidden[0], act_dims[0], Hk, self.rnn_layers)
            target = critic.clone()
        else:
            self.nparams = self.spec.nparams()
            critic, target = init_flat_params(obs_dims, hidden, act_dims, use_orthogonal_init, self.sharing)  # the reference's RNG draws
            critic = pad_blocks(critic, obs_dims[0], hidden, act_dims[0], Hk)
            target = critic.clone()
        assert critic.shape == (self.spec.n_blocks, self.nparams)
        self.params = critic.to(self.device).contiguous()
        self.target_params = target.to(self.device).contiguous()
        self.gamma = float(get("gamma", 0.99))
        self.grad_clip = get("grad_clip", 1.0)
        self.double_q = bool(get("double_q", True))
        self.target_update_interval_or_tau = get("target_update_interval_or_tau", 200)
        self.updater = (_hip.GruUpdater if self.recurrent else _hip.WideDqnUpdater if self.spec.wide else _hip.DqnUpdater)(self.spec, self.params, self.target_params, lr=float(get("lr", 3e-4)),
                                       gamma=self.gamma, grad_clip=self.grad_clip, double_q=self.double_q,
                                       standardise_returns=self.standardise_returns, optimizer=self.optimizer,
                                       **({"split16": True} if get("split16", False) else {}))  # `split16`: not a reference key - the opt-in split-fp16 learner (hip.DqnUpdater)
        self.updates = 0
        self.last_target_update = 0
        self.mode = 0  # IDQN
        self._obs1 = torch.zeros(self.n_agents, 1, self.spec.obs_dim, device=self.device)
        self._u1 = torch.ones(1, device=self.device)
        self._ra1 = torch.zeros(self.n_agents, 1, dtype=torch.int32, device=self.device)

    # ---- reference interface ---------------------------------------------------------------
    def forward(self, inputs):
        raise NotImplementedError("Forward not implemented. Use act or update instead!")

    @property
    def ret_ms(self):
        """RunningMeanStd on the device (dqn/model.py:88-89): shape (n_agents,) for QNetwork; VDNetwork / QMixNetwork: one (mean, var) per
        batch column once the first update has run (the reference's (1,)-shaped statistics broadcast to that shape there too)"""
        return self.updater.ret_stats

    def init_hiddens(self, batch_size):
        if self.recurrent:  # RNNNetwork.init_hiddens (utils/models.py:96-102): [num_layers, batch, H] zeros per agent
            return [torch.zeros(self.rnn_layers, batch_size, self.spec.hidden, device=self.device) for _ in range(self.n_agents)]
        return [None] * self.n_agents

    def q_values(self, obs, hiddens=None):
        """obs f32 [P][N][D] on the device -> Q [P][N][A] (critic forward, K2); recurrent networks also take / return the hidden
        state [P][N][H] ([L][P][N][H] for L > 1 stacked GRU layers): (Q, hiddens)"""
        if self.recurrent:
            q, h = _hip.gru_forward(self.spec, self.params, obs.unsqueeze(1).contiguous(), h_in=hiddens, want_h=True)
            return q[:, 0], h
        if self.spec.wide:
            return _hip.wide_forward(self.spec, self.params, obs.contiguous())
        q = torch.empty(obs.shape[0], obs.shape[1], self.spec.n_actions, device=obs.device)
        n = obs.shape[1]
        _hip.dqn_act(self.spec, self.params, obs, 0.0, u=torch.ones(n, device=obs.device),
                     rand_actions=torch.zeros(obs.shape[0], n, dtype=torch.int32, device=obs.device), q_out=q)
        return q

    def act(self, inputs, hiddens, epsilon, action_masks=None):
        """One env (dqn/model.py:94-116): ONE python `random.random()` draw decides the joint action."""
        if self.recurrent:  # the networks run even on a random step: the hidden state advances (model.py:99)
            for p, o in enumerate(inputs):
                self._obs1[p, 0].copy_(torch.as_tensor(o, dtype=torch.float32))
            L = self.rnn_layers  # per agent [num_layers, 1, H] (utils/models.py:96-102) <-> the kernels' [P][1][H] ([L][P][1][H] for a stack)
            h_in = None if hiddens is None or hiddens[0] is None else torch.stack([h.reshape(L, 1, -1) for h in hiddens], dim=1).to(self.device).contiguous()
            q, h = self.q_values(self._obs1, h_in if h_in is None or L > 1 else h_in[0])
            hiddens = [(h[:, p] if L > 1 else h[p]).reshape(L, 1, -1) for p in range(self.n_agents)]
            if epsilon > random.random():
                if action_masks is not None:
                    return [random.choice([i for i, m in enumerate(mask) if m == 1]) for mask in action_masks], hiddens
                return list(self.action_space.sample()), hiddens
            qv = q[:, 0]
            if action_masks is not None:
                m = torch.as_tensor(np.asarray(action_masks, np.float32)).to(qv.device)
                qv = qv * m + (1 - m) * -1e8
            return [int(a) for a in qv.argmax(-1).tolist()], hiddens
        if epsilon > random.random():
            if action_masks is not None:  # model.py:106-111: a random ALLOWED action per agent
                return [random.choice([i for i, m in enumerate(mask) if m == 1]) for mask in action_masks], hiddens
            return list(self.action_space.sample()), hiddens
        for p, o in enumerate(inputs):
            self._obs1[p, 0].copy_(torch.as_tensor(o, dtype=torch.float32))
        if action_masks is not None:  # model.py:100-104: value * mask + (1 - mask) * -1e8, then argmax
            q = self.q_values(self._obs1)[:, 0]
            m = torch.as_tensor(np.asarray(action_masks, np.float32)).to(q.device)
            return [int(a) for a in (q * m + (1 - m) * -1e8).argmax(-1).tolist()], hiddens
        if self.spec.wide:  # no fused act kernel for this shape: values from the GEMM path, first maximum (torch.argmax)
            return [int(a) for a in self.q_values(self._obs1)[:, 0].argmax(-1).tolist()], hiddens
        acts = _hip.dqn_act(self.spec, self.params, self._obs1, 0.0, u=self._u1, rand_actions=self._ra1)
        return [int(a) for a in acts[:, 0].tolist()], hiddens

    def act_batched(self, obs, epsilon, seed, episode, ep_length):
        """N envs on the device, Philox noise keyed like the fused collector."""
        if self.spec.wide:
            return _hip.act_from_q(self.q_values(obs), epsilon, seed, episode, ep_length)
        return _hip.dqn_act(self.spec, self.params, obs, epsilon, seed=seed, episode=episode, ep_length=ep_length)

    def _to_device_batch(self, batch):
        f = lambda t, dt: t.to(self.device, dt).contiguous()
        return _hip.Batch(f(batch.obss, torch.float32), f(batch.actions, torch.int64), f(batch.rewards, torch.float32),
                          f(batch.dones, torch.float32), f(batch.filled, torch.float32),
                          None if batch.action_mask is None else f(batch.action_mask, torch.float32))

    def update_async(self, batch, grad_sync=None, world=1, replay=None, **sample_kw):
        """loss/grad -> [grad_sync(grad)] -> clip+Adam -> target update; returns the device loss tensor.
        `batch` is a Batch, or (with replay=DeviceReplay) a batch SIZE: the episodes are then gathered
        inside the loss/grad kernel (sample_kw: length, seed, counter | idx)."""
        if replay is not None:
            loss, grad = self.updater.loss_grad_replay(replay, int(batch), mode=self.mode, **sample_kw)
        else:
            loss, grad = self.updater.loss_grad(self._to_device_batch(batch), mode=self.mode)
        if grad_sync is not None:
            grad_sync(grad)
        self.updates += 1
        tui = self.target_update_interval_or_tau
        hard = tui > 1.0 and (self.updates - self.last_target_update) >= tui  # dqn/model.py:176-185
        tau = float(tui) if tui < 1.0 else 0.0
        self.updater.apply(hard_update=hard, tau=tau, grad_scale=1.0 / world)
        if hard:
            self.last_target_update = self.updates
        return loss

    def update(self, batch):
        return {"loss": float(self.update_async(batch)[0].item())}  # .item(): the reference's per-update sync

    def update_target(self):
        tui = self.target_update_interval_or_tau
        if tui > 1.0 and (self.updates - self.last_target_update) >= tui:
            self.hard_update()
            self.last_target_update = self.updates
        elif tui < 1.0:
            self.soft_update(tui)

    def soft_update(self, tau):
        self.target_params.mul_(1 - tau).add_(self.params, alpha=tau)

    def hard_update(self):
        self.target_params.copy_(self.params)

    # ---- torch-module-like surface (checkpoints, logger.watch) --------------------------------
    def _views(self, block, prefix):
        out = OrderedDict()
        S = self.spec
        group = "independent" if self.sharing is None else "networks"  # utils/models.py:146 / :204
        if not self.recurrent:  # the live [h1, h2] tensors inside the (possibly zero-padded) blocks
            for i in range(S.n_blocks):
                for name, view in block_views(block[i], S.obs_dim, self.live_hidden, S.n_actions, S.hidden):
                    out[f"{prefix}.{group}.{i}.{name}"] = view
            return out
        for i in range(S.n_blocks):  # live tensors of the (possibly zero-padded) recurrent blocks; gate matrices as (3, h, h) views
            for name, view, _ in gru_block_views(block[i], S.obs_dim, self.live_hidden[0], S.n_actions, S.hidden, self.rnn_layers):
                out[f"{prefix}.{group}.{i}.{name}"] = view
        return out

    def _ref_shapes(self, prefix):
        """state_dict key -> the reference tensor's shape (the recurrent gate matrices are kept as (3, h, h) views of the padded block)"""
        S = self.spec
        if not self.recurrent:
            return {}
        group = "independent" if self.sharing is None else "networks"
        return {f"{prefix}.{group}.{i}.{name}": shape for i in range(S.n_blocks)
                for name, _, shape in gru_block_views(self.params[i], S.obs_dim, self.live_hidden[0], S.n_actions, S.hidden, self.rnn_layers)}

    def parameters(self):
        return list(self._views(self.params, "critic").values())

    def state_dict(self):
        sd = OrderedDict()
        shapes = {**self._ref_shapes("critic"), **self._ref_shapes("target")}
        for k, v in list(self._views(self.params, "critic").items()) + list(self._views(self.target_params, "target").items()):
            sd[k] = v.detach().clone().reshape(shapes.get(k, v.shape))
        return sd

    def load_state_dict(self, sd):
        for block, prefix in ((self.params, "critic"), (self.target_params, "target")):
            for k, view in self._views(block, prefix).items():
                view.copy_(sd[k].to(self.device).reshape(view.shape))

    def to(self, device):
        return self

    def __repr__(self):
        S = self.spec
        share = "" if self.sharing is None else f", sharing={list(self.sharing)}"
        pad = "" if all(h == S.hidden for h in self.live_hidden) else f" zero-padded to width {S.hidden}"
        return (f"QNetwork[HIP](agents={S.n_agents}, mlp={S.obs_dim}-{'-'.join(str(h) for h in self.live_hidden)}-{S.n_actions}{pad}, "
                f"params={self.nparams}/network{share})")


class VDNetwork(QNetwork):
    """Value-decomposition network - drop-in for `dqn.model.VDNetwork` (marlbase/dqn/model.py:199-269,
    configs/algorithm/vdn.yaml).  Same agent networks and interface as QNetwork; the loss couples the
    agents through the sum mixer (chosen_tot = sum_p Q_p, target_tot = sum_p target_p, reward of agent 0 -
    the env is expected to run under CooperativeReward).  On the device the step is
    agent-forward ("qsel") -> mixer kernel -> agent-backward with the mixer's dL/dchosen."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.mode = 1

    def __repr__(self):
        return super().__repr__().replace("QNetwork[HIP]", "VDNetwork[HIP]")


_MIXER_LAYOUT = ("hyper_w_1.0", "hyper_w_1.2", "hyper_w_final.0", "hyper_w_final.2", "hyper_b_1", "V.0", "V.2")


KERNEL_EMBED, KERNEL_HYPER = 64, 32  # the widths csrc/qmix.h is built for (configs/algorithm/qmix.yaml:14-17)


def mixer_is_padded(embed_dim, hypernet_layers, hypernet_embed):
    """two-layer hypernets no wider than csrc/qmix.h's fused kernels: the block is laid out at 64 / 32 and the mixer sits zero-padded in it
    (exact: mixer_live_views).  Everything else QMixer.__init__ builds (dqn/model.py:283-301: `hypernet_layers` 1, wider embeddings /
    hypernets) keeps its own sizes and runs on the generic mixer stage (csrc/qmix_gen.hip)."""
    return hypernet_layers == 2 and 0 < embed_dim <= KERNEL_EMBED and 0 < hypernet_embed <= KERNEL_HYPER


def _mixer_modules(n_agents, state_dim, embed_dim, hypernet_layers, hypernet_embed):
    """(state_dict name, (out, in)) of the QMixer's Linear layers in mixer.parameters() order (dqn/model.py:283-311)"""
    P, SD, E, HE = n_agents, state_dim, embed_dim, hypernet_embed
    if hypernet_layers == 1:
        hyper = [("hyper_w_1", (E * P, SD)), ("hyper_w_final", (E, SD))]
    elif hypernet_layers == 2:
        hyper = [("hyper_w_1.0", (HE, SD)), ("hyper_w_1.2", (E * P, HE)), ("hyper_w_final.0", (HE, SD)), ("hyper_w_final.2", (E, HE))]
    else:
        raise Exception("Error setting number of hypernet layers (please set `hypernet_layers=1` or `hypernet_layers=2`).")  # dqn/model.py:298-301
    return hyper + [("hyper_b_1", (E, SD)), ("V.0", (E, SD)), ("V.2", (1, E))]


def init_flat_mixer(n_agents, state_dim, embed_dim, hypernet_layers, hypernet_embed):
    """Initial mixer AND target-mixer blocks in mixer.parameters() order, consuming torch's global RNG like
    QMixNetwork.__init__ does (mixer, then target mixer, then hard_update; dqn/model.py:283-312, 361-363)."""
    if not (0 < embed_dim <= 1024 and (hypernet_layers == 1 or 0 < hypernet_embed <= 1024)):
        raise NotImplementedError(f"QMixer with embed_dim {embed_dim} / hypernet_embed {hypernet_embed}: widths up to 1024")
    layout = _mixer_modules(n_agents, state_dim, embed_dim, hypernet_layers, hypernet_embed)

    def one():
        mods = [nn.Linear(i, o) for _, (o, i) in layout]
        return torch.cat([t.detach().reshape(-1) for m in mods for t in (m.weight, m.bias)]), [m.weight.shape for m in mods]

    mixer, shapes = one()
    one()  # the target mixer's own draw, overwritten by hard_update
    return mixer, mixer.clone(), shapes


def mixer_plain_views(block, n_agents, state_dim, embed_dim, hypernet_layers, hypernet_embed):
    """[(state_dict name, view, reference shape)] of a QMixer stored at its own sizes (the generic mixer stage): plain slices"""
    out, o = [], 0
    for name, (n_out, n_in) in _mixer_modules(n_agents, state_dim, embed_dim, hypernet_layers, hypernet_embed):
        out.append((f"{name}.weight", block[o:o + n_out * n_in].view(n_out, n_in), (n_out, n_in)))
        o += n_out * n_in
        out.append((f"{name}.bias", block[o:o + n_out], (n_out,)))
        o += n_out
    assert o == block.numel(), (o, block.numel())
    return out


def mixer_live_views(block, n_agents, state_dim, embed_dim, hypernet_embed):
    """The reference-sized tensors of a QMixer with (embed_dim, hypernet_embed) <= (64, 32) as strided views INTO a block laid out for the
    kernels' widths (mixer.parameters() order at 64 / 32).  Everything outside the views is zero and stays zero: a padded embedding unit
    has w1 = |0| = 0, b1 = 0, hidden = elu(0) = 0, w_final = |0| = 0, a padded hypernet unit relu(0) = 0 with zero outgoing weights -
    nothing reaches the output, every gradient into the padding is a product with one of those zeros, and Adam leaves an exactly zero
    gradient's parameter where it is.  [(state_dict name, view, reference shape)]; the view's own shape differs from the reference's only
    for hyper_w_1.2 ((P, e, h) / (P, e) against (P * e, h) / (P * e,))."""
    P, SD, E, HE, e, h = n_agents, state_dim, KERNEL_EMBED, KERNEL_HYPER, embed_dim, hypernet_embed
    out, o = [], 0

    def take(n):
        nonlocal o
        t = block[o:o + n]
        o += n
        return t

    out.append(("hyper_w_1.0.weight", take(HE * SD).view(HE, SD)[:h], (h, SD)))
    out.append(("hyper_w_1.0.bias", take(HE)[:h], (h,)))
    out.append(("hyper_w_1.2.weight", take(E * P * HE).view(P, E, HE)[:, :e, :h], (P * e, h)))
    out.append(("hyper_w_1.2.bias", take(E * P).view(P, E)[:, :e], (P * e,)))
    out.append(("hyper_w_final.0.weight", take(HE * SD).view(HE, SD)[:h], (h, SD)))
    out.append(("hyper_w_final.0.bias", take(HE)[:h], (h,)))
    out.append(("hyper_w_final.2.weight", take(E * HE).view(E, HE)[:e, :h], (e, h)))
    out.append(("hyper_w_final.2.bias", take(E)[:e], (e,)))
    out.append(("hyper_b_1.weight", take(E * SD).view(E, SD)[:e], (e, SD)))
    out.append(("hyper_b_1.bias", take(E)[:e], (e,)))
    out.append(("V.0.weight", take(E * SD).view(E, SD)[:e], (e, SD)))
    out.append(("V.0.bias", take(E)[:e], (e,)))
    out.append(("V.2.weight", take(E).view(1, E)[:, :e], (1, e)))
    out.append(("V.2.bias", take(1), (1,)))
    assert o == block.numel(), (o, block.numel())
    return out


def pad_mixer(live_flat, n_agents, state_dim, embed_dim, hypernet_embed):
    """a reference-sized flat mixer block (mixer.parameters() order) inside a zeroed kernel-sized one"""
    P, SD, E, HE = n_agents, state_dim, KERNEL_EMBED, KERNEL_HYPER
    n = 2 * (HE * SD + HE) + E * P * HE + E * P + E * HE + E + 2 * (E * SD + E) + E + 1
    block = torch.zeros(n, dtype=live_flat.dtype, device=live_flat.device)
    o = 0
    for _, view, shape in mixer_live_views(block, P, SD, embed_dim, hypernet_embed):
        k = int(np.prod(shape))
        view.copy_(live_flat[o:o + k].view(view.shape))
        o += k
    assert o == live_flat.numel(), (o, live_flat.numel())
    return block


class QMixNetwork(QNetwork):
    """QMIX - drop-in for `dqn.model.QMixNetwork` (marlbase/dqn/model.py:334-443, configs/algorithm/qmix.yaml).
    Agent networks as QNetwork; the joint value is the monotonic mixer of the chosen Q-values conditioned on the
    state (= all agents' observations concatenated).  The mixer lives in a second flat block (`mixer_params`, in
    mixer.parameters() order) next to the per-agent blocks; on the device the step is agent-forward -> mixer stage
    (csrc/qmix.h) -> agent-backward; clip_grad_norm_ covers the critic only, Adam steps critic and mixer together."""

    def __init__(self, obs_space, action_space, cfg, layers, parameter_sharing=False, use_rnn=False, use_orthogonal_init=True,
                 mixing=None, device="cuda"):
        super().__init__(obs_space, action_space, cfg, layers, parameter_sharing, use_rnn, use_orthogonal_init, device)
        mixing = dict(mixing or dict(embed_dim=64, hypernet_layers=2, hypernet_embed=32))
        self.mixing = dict(embed_dim=int(mixing["embed_dim"]), hypernet_layers=int(mixing["hypernet_layers"]),
                           hypernet_embed=int(mixing["hypernet_embed"]))
        self.mixer_fp16 = bool(mixing.get("fp16", False))  # opt-in deviation (not a reference key): first mixer layers on the fp16 MFMA
        state_dim = sum(flatdim(o) for o in obs_space)
        mixer, tmixer, self._mixer_shapes = init_flat_mixer(self.n_agents, state_dim, **self.mixing)
        self._state_dim = state_dim
        self._mixer_padded = mixer_is_padded(**self.mixing)
        if self._mixer_padded:
            # blocks laid out for the fused kernels' widths (64 / 32); narrower mixers sit zero-padded inside them (exact: mixer_live_views)
            self.mixer_params = pad_mixer(mixer, self.n_agents, state_dim, self.mixing["embed_dim"], self.mixing["hypernet_embed"]).to(self.device).contiguous()
            self.target_mixer_params = pad_mixer(tmixer, self.n_agents, state_dim, self.mixing["embed_dim"], self.mixing["hypernet_embed"]).to(self.device).contiguous()
            kernel_mixing = dict(embed_dim=KERNEL_EMBED, hypernet_layers=2, hypernet_embed=KERNEL_HYPER, fp16=self.mixer_fp16)
        else:
            # hypernet_layers 1, or wider than the fused kernels: the block at its own sizes, on the generic mixer stage (csrc/qmix_gen.hip)
            if self.mixer_fp16:
                raise NotImplementedError("mixing.fp16 (opt-in) exists for two-layer hypernets up to embed_dim 64 / hypernet_embed 32")
            self.mixer_params, self.target_mixer_params = mixer.to(self.device).contiguous(), tmixer.to(self.device).contiguous()
            kernel_mixing = dict(self.mixing)
        up = self.updater
        self.updater = (_hip.GruQmixUpdater if self.recurrent else _hip.WideQmixUpdater if self.spec.wide else _hip.QmixUpdater)(self.spec, self.params, self.target_params, self.mixer_params, self.target_mixer_params,
                                        mixing=kernel_mixing, lr=up.lr, gamma=self.gamma, grad_clip=self.grad_clip,
                                        double_q=self.double_q, standardise_returns=self.standardise_returns, optimizer=self.optimizer)
        self.mode = 2

    def update_async(self, batch, grad_sync=None, world=1, replay=None, **sample_kw):
        sync = None
        if grad_sync is not None:
            def sync(grad):  # critic + mixer gradients are one contiguous buffer: one all-reduce per update
                grad_sync(self.updater.joint_grad)
        return super().update_async(batch, grad_sync=sync, world=world, replay=replay, **sample_kw)

    def soft_update(self, tau):
        super().soft_update(tau)
        self.target_mixer_params.mul_(1 - tau).add_(self.mixer_params, alpha=tau)

    def hard_update(self):
        super().hard_update()
        self.target_mixer_params.copy_(self.mixer_params)

    def _mixer_views(self, block, prefix):
        """state_dict key -> (live strided view into the kernel-sized block, the reference tensor's shape)"""
        if not self._mixer_padded:
            return OrderedDict((f"{prefix}.{name}", (view, shape)) for name, view, shape in
                               mixer_plain_views(block, self.n_agents, self._state_dim, **self.mixing))
        return OrderedDict((f"{prefix}.{name}", (view, shape)) for name, view, shape in
                           mixer_live_views(block, self.n_agents, self._state_dim, self.mixing["embed_dim"], self.mixing["hypernet_embed"]))

    def mixer_flat(self, block=None):
        """the mixer in mixer.parameters() order at the CONFIGURED widths (what the reference's optimiser sees)"""
        block = self.mixer_params if block is None else block
        return torch.cat([v.reshape(-1) for v, _ in self._mixer_views(block, "mixer").values()])

    def parameters(self):
        return super().parameters() + [v for v, _ in self._mixer_views(self.mixer_params, "mixer").values()]

    def state_dict(self):
        sd = super().state_dict()
        for k, (v, shape) in list(self._mixer_views(self.mixer_params, "mixer").items()) + \
                list(self._mixer_views(self.target_mixer_params, "target_mixer").items()):
            sd[k] = v.detach().reshape(shape).clone()
        return sd

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        for block, prefix in ((self.mixer_params, "mixer"), (self.target_mixer_params, "target_mixer")):
            for k, (view, shape) in self._mixer_views(block, prefix).items():
                src = sd[k].to(self.device)
                assert tuple(src.shape) == tuple(shape), (k, tuple(src.shape), shape)
                view.copy_(src.reshape(view.shape))

    def __repr__(self):
        return super().__repr__().replace("QNetwork[HIP]", "QMixNetwork[HIP]") + f" + mixer({sum(int(np.prod(sh)) for _, sh in self._mixer_views(self.mixer_params, 'mixer').values())} params)"
