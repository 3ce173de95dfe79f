"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's recurrent Q-network (`use_rnn: True`).

Follows marlbase/utils/models.py:51-116 (RNNNetwork: Linear(D, H) -> ReLU -> nn.GRU(H, H, num_layers = len(layers) - 1) -> Linear(H, A);
`layers: [H, H]` is one GRU layer, [H] * (L + 1) stacks L: the port reads L off the block's size) and the way the DQN family drives it:
  QNetwork.act            dqn/model.py:94-116   one step, hidden state [1, 1, H] per agent carried by the caller
  QNetwork._compute_loss  dqn/model.py:118-163  whole [T+1, B] sequences from a zero hidden state (`hiddens=None`)
torch.nn.GRU's cell (gate order r, z, n in weight_ih_l0 / weight_hh_l0):
  r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)      z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
  n = tanh(W_in x + b_in + r * (W_hn h + b_hn))   h' = (1 - z) * n + z * h
Parameters of one agent = one flat fp32 block in parameters() order:
  first_layer.weight [H, D] | first_layer.bias [H] | rnn.weight_ih_l0 [3H, H] | rnn.weight_hh_l0 [3H, H] |
  rnn.bias_ih_l0 [3H] | rnn.bias_hh_l0 [3H] | final_layer.weight [A, H] | final_layer.bias [A]
  ... then rnn.*_l1, rnn.*_l2, ... for a stack, final_layer last.
PINNED by tests/golden/learner_gru_*.npz (oracle/make_golden_gru.py runs the reference's own QNetwork(use_rnn=True), incl. layers [64] * 3).
"""
import numpy as np
import torch

from . import dqn_port as dp

def SHAPES(D, H, A, L=1):
    """RNNNetwork's parameters() order; nn.GRU(num_layers=L) lists layer l's four tensors behind layer l - 1's (utils/models.py:83-92)"""
    return ((H, D), (H,)) + ((3 * H, H), (3 * H, H), (3 * H,), (3 * H,)) * L + ((A, H), (A,))


NAMES = ("first_layer.weight", "first_layer.bias", "rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0",
         "final_layer.weight", "final_layer.bias")


def names(L=1):
    return (("first_layer.weight", "first_layer.bias") + tuple(f"rnn.{n}_l{l}" for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
            + ("final_layer.weight", "final_layer.bias"))


def nparams(D, H, A, L=1):
    return sum(int(np.prod(s)) for s in SHAPES(D, H, A, L))


def depth(block, D, H, A):
    """stacked GRU layers of a flat block: `layers = [H] * (L + 1)` -> nn.GRU(num_layers=L) (utils/models.py:74-90)"""
    extra = block.numel() - nparams(D, H, A)
    per = 6 * H * H + 6 * H
    assert extra >= 0 and extra % per == 0, (block.numel(), D, H, A)
    return 1 + extra // per


def split(block, D, H, A):
    out, o = [], 0
    for s in SHAPES(D, H, A, depth(block, D, H, A)):
        n = int(np.prod(s))
        out.append(block[o:o + n].reshape(s))
        o += n
    return out


def cell(parts, x, h):
    """one step: x [..., D], h [..., H] ([L, ..., H] for L > 1 stacked layers) -> (q [..., A], h' like h).  Layer l's input is layer
    l - 1's new hidden state (torch.nn.GRU; no dropout: RNNNetwork passes none)"""
    L = (len(parts) - 4) // 4
    W1, b1, W3, b3 = parts[0], parts[1], parts[-2], parts[-1]
    H = h.shape[-1]
    inp = torch.relu(torch.nn.functional.linear(x, W1, b1))
    hs = []
    for l in range(L):
        Wih, Whh, bih, bhh = parts[2 + 4 * l:6 + 4 * l]
        hl = h[l] if L > 1 else h
        gi = torch.nn.functional.linear(inp, Wih, bih)
        gh = torch.nn.functional.linear(hl, Whh, bhh)
        r = torch.sigmoid(gi[..., :H] + gh[..., :H])
        z = torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
        n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
        inp = (1 - z) * n + z * hl
        hs.append(inp)
    return torch.nn.functional.linear(inp, W3, b3), (torch.stack(hs) if L > 1 else hs[0])


def sequence(block, obss, D, H, A, h0=None):
    """obss [S, B, D] -> (q [S, B, A], h_S [B, H] or [L, B, H]) from h0 (zeros when None), one step after the other"""
    parts = split(block, D, H, A)
    L = (len(parts) - 4) // 4
    h = (torch.zeros(obss.shape[1], H, dtype=obss.dtype) if L == 1 else torch.zeros(L, obss.shape[1], H, dtype=obss.dtype)) if h0 is None else h0
    qs = []
    for t in range(obss.shape[0]):
        q, h = cell(parts, obss[t], h)
        qs.append(q)
    return torch.stack(qs), h


def q_values(params, obss, D, H, A):
    """obss [P, S, B, D] -> [P, S, B, A]"""
    return torch.stack([sequence(params[p], obss[p], D, H, A)[0] for p in range(params.shape[0])])


def compute_loss(params, tparams, batch, gamma, double_q, D, H, A, mode="idqn"):
    """dqn_port.compute_loss with the recurrent networks (same TD arithmetic, dqn/model.py:118-163 / 224-269)"""
    saved = dp.q_values
    dp.q_values = q_values
    try:
        return dp.compute_loss(params, tparams, batch, gamma, double_q, D, H, A, mode=mode)
    finally:
        dp.q_values = saved


def compute_qmix_loss(params, tparams, mixer, tmixer, batch, gamma, double_q, D, H, A):
    """qmix_port.compute_loss with the recurrent agent networks (QMixNetwork._compute_loss, dqn/model.py:374-427)"""
    from . import qmix_port as qp

    saved = dp.q_values
    dp.q_values = q_values
    try:
        return qp.compute_loss(params, tparams, mixer, tmixer, batch, gamma, double_q, D, H, A)
    finally:
        dp.q_values = saved


import contextlib


@contextlib.contextmanager
def recurrent_ac(L=1, Lc=None):
    """oracle.ac_update_port with recurrent actors and critics (ac/model.py:189-352 with use_rnn): its mlp / split / nparams hooks
    become the sequence forward and the recurrent block layout for the duration of the `with` block; L stacked GRU layers in the actors,
    Lc (default L) in the critics (the one-output networks: each family is built from its own `layers` list, ac/model.py:45-97)"""
    saved = dp.mlp, dp.split, dp.nparams
    Lc = L if Lc is None else Lc
    dp.mlp = lambda block, x, D, H, A: sequence(block, x, D, H, A)[0]  # x [S, N, D] from zero hidden states
    dp.split, dp.nparams = split, (lambda D, H, A: nparams(D, H, A, Lc if A == 1 else L))
    try:
        yield
    finally:
        dp.mlp, dp.split, dp.nparams = saved


@contextlib.contextmanager
def mixed_ac():
    """oracle.ac_update_port with actor.use_rnn != critic.use_rnn (ac/model.py:45-97: each family built from its own flag): the hooks look at
    the block they are given - a block of the recurrent layout's size runs the sequence forward from zero hidden states, any other the
    feed-forward network (the two layouts have different sizes for every (D, H, A))"""
    saved = dp.mlp, dp.split, dp.nparams

    def is_gru(block, D, H, A):  # (a stack of L layers has L (6 H^2 + 6 H) gate parameters where the feed-forward net has H^2 + H: never equal)
        return any(block.numel() == nparams(D, H, A, L) for L in range(1, 9))

    dp.mlp = lambda block, x, D, H, A: sequence(block, x, D, H, A)[0] if is_gru(block, D, H, A) else saved[0](block, x, D, H, A)
    dp.split = lambda block, D, H, A: split(block, D, H, A) if is_gru(block, D, H, A) else saved[1](block, D, H, A)
    try:
        yield
    finally:
        dp.mlp, dp.split, dp.nparams = saved
