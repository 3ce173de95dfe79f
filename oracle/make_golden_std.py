"""tests/golden/learner_std_*.npz: `standardise_returns=True` (RunningMeanStd, marlbase/utils/standardise_stream.py) in the
reference's own QNetwork (dqn/model.py:146-158) and A2CNetwork (ac/model.py:195-204).  Build container only.
    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_std
Per update: the loss / metrics, the parameter blocks, and the running statistics (mean, var, count) after it."""
import contextlib
import io
import os

import numpy as np
import torch

from .ac_update_port import synthetic_batch as ac_batch
from .dqn_port import synthetic_batch
from .make_golden import OUT, Box, Cfg, Discrete, flat_params, import_reference


def idqn(rm, rt):
    P, T, B, D, A, H = 2, 25, 32, 15, 6, 64
    torch.manual_seed(1000)
    cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, target_update_interval_or_tau=2, double_q=True,
              standardise_returns=True)
    with contextlib.redirect_stdout(io.StringIO()):
        net = rm.QNetwork([Box(D)] * P, [Discrete(A)] * P, cfg, [H, H], False, False, True, "cpu")
    g = torch.Generator().manual_seed(1001)
    with torch.no_grad():
        for p in net.critic.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
        for p in net.target.parameters():
            p.add_(0.08 * torch.randn(p.shape, generator=g))
    out = dict(P=P, T=T, B=B, D=D, A=A, H=H, params0=flat_params(net.critic).numpy(), target0=flat_params(net.target).numpy())
    batches = [synthetic_batch(P, T, B, D, A, seed=1100 + i) for i in range(3)]
    losses = []
    for i, b in enumerate(batches):
        losses.append(net.update(rt.Batch(b["obss"], b["actions"], b["rewards"], b["dones"], b["filled"], None))["loss"])
        out[f"params{i + 1}"] = flat_params(net.critic).numpy()
        out[f"mean{i + 1}"], out[f"var{i + 1}"], out[f"count{i + 1}"] = net.ret_ms.mean.numpy(), net.ret_ms.var.numpy(), np.float64(net.ret_ms.count)
        for k, v in b.items():
            out[f"batch{i}_{k}"] = v.numpy()
    out["losses"] = np.array(losses, np.float32)
    np.savez_compressed(os.path.join(OUT, "learner_std_idqn_H64.npz"), **out)
    print("learner_std_idqn_H64", losses, out["mean3"], out["var3"], out["count3"])


def a2c(ram, rat):
    P, T, N, D, A, H = 2, 25, 12, 15, 6, 64
    torch.manual_seed(1200)
    cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=False, n_steps=5, entropy_coef=0.001, value_loss_coef=0.5,
              standardise_returns=True, target_update_interval_or_tau=200)
    net_cfg = dict(layers=[H, H], parameter_sharing=False, use_orthogonal_init=True, use_rnn=False)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ram.A2CNetwork([Box(D)] * P, [Discrete(A)] * P, cfg, Cfg(net_cfg), Cfg(dict(net_cfg, centralised=False)), "cpu")
    g = torch.Generator().manual_seed(1201)
    with torch.no_grad():
        for p in list(net.actor.parameters()) + list(net.critic.parameters()):
            p.add_(0.05 * torch.randn(p.shape, generator=g))
        for p in net.target_critic.parameters():
            p.add_(0.08 * torch.randn(p.shape, generator=g))
    out = dict(P=P, T=T, N=N, D=D, A=A, H=H, n_steps=5, gamma=0.99, entropy_coef=0.001, value_loss_coef=0.5,
               actor0=flat_params(net.actor).numpy(), critic0=flat_params(net.critic).numpy(), target0=flat_params(net.target_critic).numpy())
    steps, metrics = [0, 250, 400], []
    for i, st in enumerate(steps):
        b = ac_batch(P, T, N, D, A, seed=1300 + i)
        m = net.update(rat.Batch(b["obss"], b["actions"], b["rewards"], b["dones"], b["filled"], None), st)
        metrics.append([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]])
        out[f"actor{i + 1}"], out[f"critic{i + 1}"] = flat_params(net.actor).numpy(), flat_params(net.critic).numpy()
        out[f"mean{i + 1}"], out[f"var{i + 1}"], out[f"count{i + 1}"] = net.ret_ms.mean.numpy(), net.ret_ms.var.numpy(), np.float64(net.ret_ms.count)
        for k, v in b.items():
            out[f"batch{i}_{k}"] = v.numpy()
    out["metrics"], out["steps"] = np.array(metrics), np.array(steps)
    np.savez_compressed(os.path.join(OUT, "learner_std_a2c_H64.npz"), **out)
    print("learner_std_a2c_H64", np.array(metrics).round(5).tolist(), out["mean3"], out["var3"])




def vdn_qmix(rm, rt):
    """learner_std_vdn_H64.npz / learner_std_qmix_H64.npz: VDNetwork / QMixNetwork with standardise_returns (dqn/model.py:221-222,
    256-264 / 357-358,415-422).  Their RunningMeanStd(shape=(1,)) is fed [T, B] returns: after the first update the statistics are
    per BATCH COLUMN (mean, var: [B]; count += T) - stored here exactly as the reference ends up holding them."""
    from .make_golden_qmix import build as build_qmix
    from .make_golden_qmix import mixer_flat

    P, T, B, D, A, H = 2, 25, 48, 15, 6, 64
    for kind in ("vdn", "qmix"):
        torch.manual_seed(2000 if kind == "vdn" else 2100)
        if kind == "vdn":
            cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, target_update_interval_or_tau=2, double_q=True,
                      standardise_returns=True)
            with contextlib.redirect_stdout(io.StringIO()):
                net = rm.VDNetwork([Box(D)] * P, [Discrete(A)] * P, cfg, [H, H], False, False, True, "cpu")
            g = torch.Generator().manual_seed(2001)
            with torch.no_grad():
                for p in net.critic.parameters():
                    p.add_(0.05 * torch.randn(p.shape, generator=g))
                for p in net.target.parameters():
                    p.add_(0.08 * torch.randn(p.shape, generator=g))
        else:
            net = build_qmix(rm, P, D, A, H, 2100)
            net.standardise_returns = True
            from marlbase.utils.standardise_stream import RunningMeanStd

            net.ret_ms = RunningMeanStd(shape=(1,))
        out = dict(P=P, T=T, B=B, D=D, A=A, H=H, params0=flat_params(net.critic).numpy(), target0=flat_params(net.target).numpy())
        if kind == "qmix":
            out["mixer0"], out["tmixer0"] = mixer_flat(net.mixer).numpy(), mixer_flat(net.target_mixer).numpy()
        losses = []
        for i in range(3):
            b = synthetic_batch(P, T, B, D, A, seed=2200 + i)
            b["rewards"][1:] = b["rewards"][0]  # CooperativeReward
            if kind == "qmix":
                b["obss"] = b["obss"] * 0.25
            losses.append(net.update(rt.Batch(b["obss"], b["actions"], b["rewards"], b["dones"], b["filled"], None))["loss"])
            out[f"params{i + 1}"] = flat_params(net.critic).numpy()
            if kind == "qmix":
                out[f"mixer{i + 1}"] = mixer_flat(net.mixer).numpy()
            out[f"mean{i + 1}"], out[f"var{i + 1}"] = net.ret_ms.mean.numpy(), net.ret_ms.var.numpy()
            out[f"count{i + 1}"] = np.float64(net.ret_ms.count)
            assert out[f"mean{i + 1}"].shape == (B,), out[f"mean{i + 1}"].shape  # per-batch-column statistics
            for k, v in b.items():
                out[f"batch{i}_{k}"] = v.numpy()
        out["losses"] = np.array(losses, np.float32)
        np.savez_compressed(os.path.join(OUT, f"learner_std_{kind}_H64.npz"), **out)
        print(f"learner_std_{kind}_H64", losses, out["mean3"][:3], out["var3"][:3], out["count3"])


def vdn_qmix_rnn(rm, rt):
    """learner_std_vdn_gru_H64.npz / learner_std_qmix_gru_H64.npz: the same with recurrent agents (use_rnn=True: RNNNetwork,
    utils/models.py:51-116; whole [T + 1, B] sequences from zero hidden states, dqn/model.py:118-163)"""
    from .make_golden_qmix import mixer_flat

    P, T, B, D, A, H = 2, 10, 32, 15, 6, 64
    mixing = dict(embed_dim=64, hypernet_layers=2, hypernet_embed=32)
    for kind in ("vdn", "qmix"):
        torch.manual_seed(2300 if kind == "vdn" else 2400)
        cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, target_update_interval_or_tau=2, double_q=True,
                  standardise_returns=True)
        with contextlib.redirect_stdout(io.StringIO()):
            if kind == "vdn":
                net = rm.VDNetwork([Box(D)] * P, [Discrete(A)] * P, cfg, [H, H], False, True, True, "cpu")
            else:
                net = rm.QMixNetwork([Box(D)] * P, [Discrete(A)] * P, cfg, [H, H], False, True, True, mixing, "cpu")
        g = torch.Generator().manual_seed(2301)
        with torch.no_grad():
            extra = list(net.target_mixer.parameters()) if kind == "qmix" else []
            for p in list(net.target.parameters()) + extra:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
        out = dict(P=P, T=T, B=B, D=D, A=A, H=H, params0=flat_params(net.critic).numpy(), target0=flat_params(net.target).numpy())
        if kind == "qmix":
            out["mixer0"], out["tmixer0"] = mixer_flat(net.mixer).numpy(), mixer_flat(net.target_mixer).numpy()
        losses = []
        for i in range(3):
            b = synthetic_batch(P, T, B, D, A, seed=2500 + i)
            b["rewards"][1:] = b["rewards"][0]  # CooperativeReward
            b["obss"] = b["obss"] * 0.25
            losses.append(net.update(rt.Batch(b["obss"], b["actions"], b["rewards"], b["dones"], b["filled"], None))["loss"])
            out[f"params{i + 1}"] = flat_params(net.critic).numpy()
            if kind == "qmix":
                out[f"mixer{i + 1}"] = mixer_flat(net.mixer).numpy()
            out[f"mean{i + 1}"], out[f"var{i + 1}"] = net.ret_ms.mean.numpy(), net.ret_ms.var.numpy()
            out[f"count{i + 1}"] = np.float64(net.ret_ms.count)
            assert out[f"mean{i + 1}"].shape == (B,), out[f"mean{i + 1}"].shape
            for k, v in b.items():
                out[f"batch{i}_{k}"] = v.numpy()
        out["losses"] = np.array(losses, np.float32)
        np.savez_compressed(os.path.join(OUT, f"learner_std_{kind}_gru_H64.npz"), **out)
        print(f"learner_std_{kind}_gru_H64", losses, out["mean3"][:3], out["var3"][:3], out["count3"])


if __name__ == "__main__":
    torch.set_num_threads(1)
    rm, rt = import_reference()
    from marlbase.ac import model as ram
    from marlbase.ac import train as rat
    idqn(rm, rt)
    a2c(ram, rat)
    vdn_qmix(rm, rt)
    vdn_qmix_rnn(rm, rt)
