"""Generate tests/golden/learner_qmix_*.npz from the REFERENCE's own QMixNetwork (marlbase/dqn/model.py:334-443).
Runs only in the build container (needs /root/reference); the vectors travel, the reference does not.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_qmix

  learner_qmix_H64.npz     : 2 agents x 15 obs (Foraging-8x8-2p-3f shapes), 25 x 32 batch: loss, critic and mixer
      gradients of QMixNetwork._compute_loss, then critic / target / mixer / target-mixer after 3 x update()
      (hard target update forced at update 2), Adam moments of the mixer.
  learner_qmix_p4_H64.npz  : 4 agents x 27 obs (15x15-4p-5f shapes), 25 x 24 batch: loss and gradients only.
  learner_qmix_e24_h16_H64.npz : 3 agents x 18 obs with mixing = {embed_dim 24, hypernet_layers 2, hypernet_embed 16}: loss, gradients, 3 updates.
  learner_qmix_L1_H64.npz      : 2 agents x 15 obs with mixing = {64, hypernet_layers 1, 32} (QMixer.__init__'s one-Linear hypernets, model.py:283-285).
  learner_qmix_L1_e40_p3_H64.npz : 3 agents x 18 obs with mixing = {40, 1, 7}; learner_qmix_e96_h48_H64.npz: 2 agents x 15 obs with {96, 2, 48}
      (wider than the fused kernels); all three: loss, gradients, 3 updates - they run on the generic mixer stage (csrc/qmix_gen.hip).
  learner_qmix_layers136.npz   : 2 agents x 15 obs, AGENT networks with layers = [136, 136] (wider than the fused agent kernels: the GEMM path,
      marlhip_wide_qmix_loss_grad) around qmix.yaml's mixer: loss, gradients, 2 updates (hard copy at update 2).
"""
import contextlib
import io
import os

import numpy as np
import torch

from .dqn_port import synthetic_batch
from .make_golden import OUT, Box, Cfg, Discrete, flat_params, import_reference


def mixer_flat(m):
    return torch.cat([p.detach().reshape(-1) for p in m.parameters()])


def mixer_grad(m):
    return torch.cat([p.grad.reshape(-1) for p in m.parameters()])


def build(ref_model, P, D, A, H, seed, mixing=None, layers=None):
    torch.manual_seed(seed)
    cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, target_update_interval_or_tau=2, double_q=True,
              standardise_returns=False)
    mixing = mixing or dict(embed_dim=64, hypernet_layers=2, hypernet_embed=32)  # configs/algorithm/qmix.yaml:14-17
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref_model.QMixNetwork([Box(D)] * P, [Discrete(A)] * P, cfg, list(layers) if layers else [H, H], False, False, True, mixing, "cpu")
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():  # critic != target, mixer != target mixer, biases non-zero
        for p in net.critic.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
        for p in net.target.parameters():
            p.add_(0.08 * torch.randn(p.shape, generator=g))
        for p in net.target_mixer.parameters():
            p.add_(0.03 * torch.randn(p.shape, generator=g))
    return net


def fixture(ref_model, ref_train, name, P, D, B, seed, updates, mixing=None, layers=None):
    T, A, H = 25, 6, 64
    net = build(ref_model, P, D, A, H, seed, mixing, layers)
    out = dict(P=P, T=T, B=B, D=D, A=A, H=H, layers=np.array(layers if layers else [H, H]), E=net.mixer.embed_dim, HE=net.mixer.hypernet_embed, L=net.mixer.hypernet_layers,
               mixer_keys=np.array([k for k in net.mixer.state_dict().keys()]), params0=flat_params(net.critic).numpy(),
               target0=flat_params(net.target).numpy(), mixer0=mixer_flat(net.mixer).numpy(),
               tmixer0=mixer_flat(net.target_mixer).numpy())
    batches = [synthetic_batch(P, T, B, D, A, seed=seed + 100 + i) for i in range(max(updates, 1))]
    for b in batches:
        b["rewards"][1:] = b["rewards"][0]  # CooperativeReward (utils/wrappers.py:106-108)
        b["obss"] = b["obss"] * 0.25        # keep the hypernet outputs in a range where elu' is exercised on both sides
    b0 = ref_train.Batch(batches[0]["obss"], batches[0]["actions"], batches[0]["rewards"], batches[0]["dones"],
                         batches[0]["filled"], None)
    loss = net._compute_loss(b0)
    net.optimizer.zero_grad()
    loss.backward()
    out["loss0"] = np.float32(loss.item())
    out["grad0"] = torch.stack([torch.cat([p.grad.reshape(-1) for p in m.parameters()]) for m in net.critic.independent]).numpy()
    out["mgrad0"] = mixer_grad(net.mixer).numpy()
    with torch.no_grad():  # the mixer outputs themselves, for a forward-only check
        q, _ = net.critic(b0.obss, None)
        q = torch.stack(q)
        out["chosen_tot0"] = net.mixer(q[:, :-1].gather(-1, b0.actions.unsqueeze(-1)).squeeze(-1),
                                       torch.concat(list(b0.obss[:, :-1]), dim=-1)).numpy()
    net.optimizer.zero_grad()
    losses = []
    for i in range(updates):
        b = batches[i]
        bb = ref_train.Batch(b["obss"], b["actions"], b["rewards"], b["dones"], b["filled"], None)
        losses.append(net.update(bb)["loss"])
        out[f"params{i + 1}"] = flat_params(net.critic).numpy()
        out[f"target{i + 1}"] = flat_params(net.target).numpy()
        out[f"mixer{i + 1}"] = mixer_flat(net.mixer).numpy()
        out[f"tmixer{i + 1}"] = mixer_flat(net.target_mixer).numpy()
    if updates:
        out["losses"] = np.array(losses, np.float32)
        st = net.optimizer.state
        out["mixer_exp_avg"] = torch.cat([st[p]["exp_avg"].reshape(-1) for p in net.mixer.parameters()]).numpy()
        out["mixer_exp_avg_sq"] = torch.cat([st[p]["exp_avg_sq"].reshape(-1) for p in net.mixer.parameters()]).numpy()
    for i, b in enumerate(batches):
        for k, v in b.items():
            out[f"batch{i}_{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(f"{name}: loss0={out['loss0']:.6f} |grad|={np.linalg.norm(out['grad0']):.4f} |mgrad|={np.linalg.norm(out['mgrad0']):.4f} "
          f"losses={losses}")


def init_fixture(ref_model):
    """init_qmix.npz: what QMixNetwork.__init__ leaves in mixer / target_mixer for torch.manual_seed(5), and the
    state_dict key order a checkpoint written by the reference carries (dqn/train.py:340-343)."""
    P, D, A, H = 2, 15, 6, 64
    torch.manual_seed(5)
    cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, target_update_interval_or_tau=200, double_q=True,
              standardise_returns=False)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref_model.QMixNetwork([Box(D)] * P, [Discrete(A)] * P, cfg, [H, H], False, False, True,
                                    dict(embed_dim=64, hypernet_layers=2, hypernet_embed=32), "cpu")
    keys = [k for k in net.state_dict().keys()]
    np.savez_compressed(os.path.join(OUT, "init_qmix.npz"), mixer=mixer_flat(net.mixer).numpy(),
                        tmixer=mixer_flat(net.target_mixer).numpy(), keys=np.array(keys))
    print(f"init_qmix: {len(keys)} state_dict keys, mixer {mixer_flat(net.mixer).numel()} params")


if __name__ == "__main__":
    torch.set_num_threads(1)
    rm, rt = import_reference()
    init_fixture(rm)
    fixture(rm, rt, "learner_qmix_H64.npz", P=2, D=15, B=32, seed=300, updates=3)
    fixture(rm, rt, "learner_qmix_p4_H64.npz", P=4, D=27, B=24, seed=400, updates=0)
    # a QMixer narrower than qmix.yaml's (QMixer.__init__, dqn/model.py:283-300, takes any widths): runs zero-padded on the 64 / 32 kernels
    fixture(rm, rt, "learner_qmix_e24_h16_H64.npz", P=3, D=18, B=24, seed=500, updates=3, mixing=dict(embed_dim=24, hypernet_layers=2, hypernet_embed=16))
    # the QMixer configurations outside the fused kernels (generic mixer stage): one-Linear hypernets, and widths beyond 64 / 32
    fixture(rm, rt, "learner_qmix_L1_H64.npz", P=2, D=15, B=32, seed=600, updates=3, mixing=dict(embed_dim=64, hypernet_layers=1, hypernet_embed=32))
    fixture(rm, rt, "learner_qmix_L1_e40_p3_H64.npz", P=3, D=18, B=24, seed=700, updates=3, mixing=dict(embed_dim=40, hypernet_layers=1, hypernet_embed=7))
    fixture(rm, rt, "learner_qmix_e96_h48_H64.npz", P=2, D=15, B=32, seed=800, updates=3, mixing=dict(embed_dim=96, hypernet_layers=2, hypernet_embed=48))
    # QMIX around agent networks wider than the fused kernels (QMixNetwork takes any `layers`, dqn/model.py:334-372)
    fixture(rm, rt, "learner_qmix_layers136.npz", P=2, D=15, B=32, seed=900, updates=2, layers=[136, 136])
