"""Generate tests/golden/learner_gru_*.npz from the REFERENCE's own QNetwork / VDNetwork with use_rnn=True
(marlbase/utils/models.py:51-116, marlbase/dqn/model.py:94-163).  Runs only in the build container.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_gru

Each file: critic / target blocks in parameters() order (oracle/gru_port.py), the state_dict key list, one Batch, the
Q-values of the whole batch (critic) for a forward-only check, loss and gradient of _compute_loss, parameters after 2 x
update(), and an `act` trace: 6 greedy steps of one env with the hidden states the reference carries between them.
  learner_gru_idqn_H64.npz   2 agents x 15 obs, QNetwork, 64-64
  learner_gru_vdn_H64.npz    3 agents x 18 obs, VDNetwork, 64-64
  learner_gru_qmix_H64.npz   2 agents x 15 obs, QMixNetwork (loss, agent + mixer gradients, 2 updates)
  learner_gru_shared_H64.npz     3 agents x 18 obs, QNetwork, parameter_sharing=True (MultiAgentSharedNetwork, utils/models.py:176-300)
  learner_gru_seps_vdn_H128.npz  3 agents x 18 obs, VDNetwork, parameter_sharing=[0, 0, 1] (SePS), 128-128
  learner_gru_std_H64.npz        2 agents x 15 obs, QNetwork, standardise_returns=True (dqn/model.py:147-158): 3 updates on 3 batches
  learner_gru_idqn_L2_H64.npz    2 agents x 15 obs, QNetwork, layers [64] * 3: nn.GRU(num_layers=2) (utils/models.py:74-90)
  learner_gru_vdn_L3_h40.npz     3 agents x 18 obs, VDNetwork, layers [40] * 4: three stacked layers at a width the kernels pad
  learner_gru_idqn_L2_h72.npz    2 agents x 15 obs, QNetwork, layers [72] * 3 (runs on the 128-wide kernels)
  learner_gru_qmix_L2_H64.npz    2 agents x 15 obs, QMixNetwork, layers [64] * 3
"""
import contextlib
import io
import os

import numpy as np
import torch

from .dqn_port import synthetic_batch
from .make_golden import OUT, Box, Cfg, Discrete, flat_params, import_reference


def fixture(ref_model, ref_train, name, cls, P, D, H, B, seed, layers=None):
    T, A = 10, 6
    layers = [H, H] if layers is None else list(layers)  # [H] * (L + 1): nn.GRU(num_layers=L) (utils/models.py:74-90)
    torch.manual_seed(seed)
    cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, target_update_interval_or_tau=200, double_q=True,
              standardise_returns=False)
    with contextlib.redirect_stdout(io.StringIO()):
        net = cls([Box(D)] * P, [Discrete(A)] * P, cfg, layers, False, True, True, "cpu")
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in net.target.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
    out = dict(P=P, T=T, B=B, D=D, A=A, H=H, keys=np.array(list(net.state_dict().keys())),
               params0=flat_params(net.critic).numpy(), target0=flat_params(net.target).numpy())
    if len(layers) != 2:
        out["layers"] = np.array(layers)
    batch = synthetic_batch(P, T, B, D, A, seed=seed + 7)
    batch["obss"] = batch["obss"] * 0.25
    if cls is ref_model.VDNetwork:
        batch["rewards"][1:] = batch["rewards"][0]
    for k, v in batch.items():
        out[f"batch_{k}"] = v.numpy()
    bb = ref_train.Batch(batch["obss"], batch["actions"], batch["rewards"], batch["dones"], batch["filled"], None)
    with torch.no_grad():
        q, _ = net.critic(bb.obss, None)
        out["q0"] = torch.stack(q).numpy()
    loss = net._compute_loss(bb)
    net.optimizer.zero_grad()
    loss.backward()
    out["loss0"] = np.float32(loss.item())
    out["grad0"] = torch.stack([torch.cat([p.grad.reshape(-1) for p in m.parameters()]) for m in net.critic.independent]).numpy()
    net.optimizer.zero_grad()
    # act trace (epsilon 0): the reference carries `hiddens` between calls (dqn/train.py:210-216)
    ga = torch.Generator().manual_seed(seed + 3)
    obs = torch.randint(-1, 8, (6, P, D), generator=ga).float() * 0.25
    hid = net.init_hiddens(1)
    acts, hids = [], []
    for t in range(6):
        a, hid = net.act([o.numpy() for o in obs[t]], hid, 0.0)
        acts.append(a)
        hids.append(torch.stack([h.reshape(-1) for h in hid]).numpy())
    out["act_obs"], out["act_actions"], out["act_hiddens"] = obs.numpy(), np.array(acts, np.int64), np.array(hids)
    out["losses"] = np.array([net.update(bb)["loss"] for _ in range(2)], np.float32)
    out["params2"] = flat_params(net.critic).numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, float(out["loss0"]), out["losses"].tolist(), out["params0"].shape)


def shared_fixture(ref_model, ref_train, name, cls, P, D, H, B, sharing, seed, layers=None):
    """use_rnn=True on MultiAgentSharedNetwork: blocks [K][n] in `critic.networks` order (K = number of distinct networks), loss,
    gradient, 2 x update(), state_dict keys.  Same seeds-per-role as `fixture` (seed: init, +1: target noise, +7: batch)."""
    from .make_golden_sharing import flat_nets

    T, A = 8, 6
    torch.manual_seed(seed)
    cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, target_update_interval_or_tau=200, double_q=True,
              standardise_returns=False)
    with contextlib.redirect_stdout(io.StringIO()):
        net = cls([Box(D)] * P, [Discrete(A)] * P, cfg, [H, H] if layers is None else list(layers), sharing, True, True, "cpu")
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in net.target.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
    out = dict(P=P, T=T, B=B, D=D, A=A, H=H, sharing=np.array(net.critic.sharing_indices), keys=np.array(list(net.state_dict().keys())),
               params0=flat_nets(net.critic).numpy(), target0=flat_nets(net.target).numpy())
    if layers is not None:
        out["layers"] = np.array(layers)
    batch = synthetic_batch(P, T, B, D, A, seed=seed + 7)
    batch["obss"] = batch["obss"] * 0.25
    if cls is ref_model.VDNetwork:
        batch["rewards"][1:] = batch["rewards"][0]
    for k, v in batch.items():
        out[f"batch_{k}"] = v.numpy()
    bb = ref_train.Batch(batch["obss"], batch["actions"], batch["rewards"], batch["dones"], batch["filled"], None)
    loss = net._compute_loss(bb)
    net.optimizer.zero_grad()
    loss.backward()
    out["loss0"] = np.float32(loss.item())
    out["grad0"] = torch.stack([torch.cat([p.grad.reshape(-1) for p in m.parameters()]) for m in net.critic.networks]).numpy()
    net.optimizer.zero_grad()
    out["losses"] = np.array([net.update(bb)["loss"] for _ in range(2)], np.float32)
    out["params2"] = flat_nets(net.critic).numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "sharing", out["sharing"].tolist(), float(out["loss0"]), out["losses"].tolist(), out["params0"].shape)


def std_fixture(ref_model, ref_train, name, P, D, H, B, seed, layers=None):
    """QNetwork(use_rnn=True, standardise_returns=True): 3 x update() on 3 batches; losses, parameters and the RunningMeanStd
    (mean, var, count) after each (the recurrent sibling of make_golden_std.idqn)"""
    T, A = 8, 6
    torch.manual_seed(seed)
    cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, target_update_interval_or_tau=200, double_q=True,
              standardise_returns=True)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref_model.QNetwork([Box(D)] * P, [Discrete(A)] * P, cfg, [H, H] if layers is None else list(layers), False, True, True, "cpu")
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in net.target.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
    out = dict(P=P, T=T, B=B, D=D, A=A, H=H, params0=flat_params(net.critic).numpy(), target0=flat_params(net.target).numpy())
    if layers is not None:
        out["layers"] = np.array(layers)
    losses = []
    for i in range(3):
        b = synthetic_batch(P, T, B, D, A, seed=seed + 7 + i)
        b["obss"] = b["obss"] * 0.25
        for k, v in b.items():
            out[f"batch{i}_{k}"] = v.numpy()
        losses.append(net.update(ref_train.Batch(b["obss"], b["actions"], b["rewards"], b["dones"], b["filled"], None))["loss"])
        out[f"params{i + 1}"] = flat_params(net.critic).numpy()
        out[f"ret_mean{i + 1}"], out[f"ret_var{i + 1}"] = net.ret_ms.mean.numpy(), net.ret_ms.var.numpy()
        out[f"ret_count{i + 1}"] = np.float64(net.ret_ms.count)
    out["losses"] = np.array(losses, np.float32)
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, losses, out["ret_mean3"], out["ret_var3"], out["ret_count3"])


def qmix_fixture(ref_model, ref_train, name, P, D, H, B, seed, layers=None):
    """QMixNetwork(use_rnn=True): loss, agent and mixer gradients, 2 updates"""
    from .make_golden_qmix import mixer_flat, mixer_grad

    T, A = 8, 6
    torch.manual_seed(seed)
    cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, target_update_interval_or_tau=200, double_q=True,
              standardise_returns=False)
    mixing = dict(embed_dim=64, hypernet_layers=2, hypernet_embed=32)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref_model.QMixNetwork([Box(D)] * P, [Discrete(A)] * P, cfg, [H, H] if layers is None else list(layers), False, True, True, mixing, "cpu")
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in list(net.target.parameters()) + list(net.target_mixer.parameters()):
            p.add_(0.04 * torch.randn(p.shape, generator=g))
    out = dict(P=P, T=T, B=B, D=D, A=A, H=H, params0=flat_params(net.critic).numpy(), target0=flat_params(net.target).numpy(),
               mixer0=mixer_flat(net.mixer).numpy(), tmixer0=mixer_flat(net.target_mixer).numpy(), keys=np.array(list(net.state_dict().keys())))
    if layers is not None:
        out["layers"] = np.array(layers)
    batch = synthetic_batch(P, T, B, D, A, seed=seed + 7)
    batch["obss"] = batch["obss"] * 0.25
    batch["rewards"][1:] = batch["rewards"][0]
    for k, v in batch.items():
        out[f"batch_{k}"] = v.numpy()
    bb = ref_train.Batch(batch["obss"], batch["actions"], batch["rewards"], batch["dones"], batch["filled"], None)
    loss = net._compute_loss(bb)
    net.optimizer.zero_grad()
    loss.backward()
    out["loss0"] = np.float32(loss.item())
    out["grad0"] = torch.stack([torch.cat([p.grad.reshape(-1) for p in m.parameters()]) for m in net.critic.independent]).numpy()
    out["mgrad0"] = mixer_grad(net.mixer).numpy()
    net.optimizer.zero_grad()
    out["losses"] = np.array([net.update(bb)["loss"] for _ in range(2)], np.float32)
    out["params2"], out["mixer2"] = flat_params(net.critic).numpy(), mixer_flat(net.mixer).numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, float(out["loss0"]), out["losses"].tolist())


def stacked(rm, rt):
    """round 6: len(layers) - 1 > 1 stacked GRU layers (utils/models.py:74-90), the reference's own classes"""
    fixture(rm, rt, "learner_gru_idqn_L2_H64.npz", rm.QNetwork, P=2, D=15, H=64, B=37, seed=2700, layers=[64, 64, 64])
    fixture(rm, rt, "learner_gru_vdn_L3_h40.npz", rm.VDNetwork, P=3, D=18, H=40, B=21, seed=2800, layers=[40, 40, 40, 40])  # padded onto the 64 kernels
    fixture(rm, rt, "learner_gru_idqn_L2_h72.npz", rm.QNetwork, P=2, D=15, H=72, B=18, seed=2900, layers=[72, 72, 72])  # padded onto the 128 kernels
    qmix_fixture(rm, rt, "learner_gru_qmix_L2_H64.npz", P=2, D=15, H=64, B=19, seed=3000, layers=[64, 64, 64])
    # a stack under parameter sharing (SePS: agents 0, 1 share a network) and with standardise_returns
    shared_fixture(rm, rt, "learner_gru_seps_vdn_L2_h24.npz", rm.VDNetwork, P=3, D=18, H=24, B=17, sharing=[0, 0, 1], seed=3100, layers=[24, 24, 24])
    std_fixture(rm, rt, "learner_gru_std_L2_h24.npz", P=2, D=15, H=24, B=23, seed=3200, layers=[24, 24, 24])


if __name__ == "__main__":
    import sys

    torch.set_num_threads(1)
    import_reference()
    from marlbase.dqn import model as rm
    from marlbase.dqn import train as rt

    if "--stacked-only" in sys.argv:
        stacked(rm, rt)
        sys.exit(0)

    fixture(rm, rt, "learner_gru_idqn_H64.npz", rm.QNetwork, P=2, D=15, H=64, B=37, seed=2100)
    fixture(rm, rt, "learner_gru_vdn_H64.npz", rm.VDNetwork, P=3, D=18, H=64, B=21, seed=2200)
    qmix_fixture(rm, rt, "learner_gru_qmix_H64.npz", P=2, D=15, H=64, B=19, seed=2300)
    shared_fixture(rm, rt, "learner_gru_shared_H64.npz", rm.QNetwork, P=3, D=18, H=64, B=20, sharing=True, seed=2400)
    shared_fixture(rm, rt, "learner_gru_seps_vdn_H128.npz", rm.VDNetwork, P=3, D=18, H=128, B=17, sharing=[0, 0, 1], seed=2500)
    std_fixture(rm, rt, "learner_gru_std_H64.npz", P=2, D=15, H=64, B=23, seed=2600)
    stacked(rm, rt)
