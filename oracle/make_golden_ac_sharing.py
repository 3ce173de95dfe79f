"""Generate tests/golden/learner_a2c_mixed_sharing_*.npz from the REFERENCE's own A2CNetwork / PPONetwork with actor.parameter_sharing
different from critic.parameter_sharing (marlbase/ac/model.py:45-97 builds the two families from their own settings).  Build container only.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_ac_sharing

  learner_a2c_mixed_sharing_H64.npz     A2C, 2 agents x 15 obs: actor.parameter_sharing = True (ONE actor), critic independent
  learner_ppo_mixed_sharing_p3_H64.npz  PPO, 3 agents x 18 obs: actor SePS [0, 0, 1], critic.parameter_sharing = True, critic.centralised
  learner_a2c_depths.npz                A2C, 2 agents x 15 obs: actor.layers [64, 64], critic.layers [48, 64, 32] (model.py:45-97: own lists)
  learner_ppo_depths_p3.npz             PPO, 3 agents x 18 obs: actor.layers [32, 48, 40], critic.layers [64, 64], critic.centralised
Blocks are [K][n] with K the family's own network count (`.independent` / `.networks` order); `keys` = state_dict key order; three
updates at env steps 0, 250, 400 as oracle/make_golden_ac.py.
"""
import contextlib
import io
import os

import numpy as np
import torch

from .ac_update_port import synthetic_batch
from .make_golden import OUT, Box, Cfg, Discrete, import_reference


def nets(family):
    return list(family.networks) if hasattr(family, "networks") else list(family.independent)


def flat(family):
    return torch.stack([torch.cat([p.detach().reshape(-1) for p in m.parameters()]) for m in nets(family)])


def fixture(ram, rat, name, cls, P, D, H, N, seed, actor_sharing, critic_sharing, centralised=False, actor_layers=None, critic_layers=None):
    T, A = 25, 6
    torch.manual_seed(seed)
    cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=0.5, n_steps=5, entropy_coef=0.001, value_loss_coef=0.5,
              standardise_returns=False, target_update_interval_or_tau=200, num_epochs=4, ppo_clip=0.2)
    base = dict(use_orthogonal_init=True, use_rnn=False)
    actor_layers, critic_layers = list(actor_layers or [H, H]), list(critic_layers or [H, H])
    with contextlib.redirect_stdout(io.StringIO()):
        net = cls([Box(D)] * P, [Discrete(A)] * P, cfg, Cfg(dict(base, layers=actor_layers, parameter_sharing=actor_sharing)),
                  Cfg(dict(base, layers=critic_layers, parameter_sharing=critic_sharing, centralised=centralised)), "cpu")
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in list(net.actor.parameters()) + list(net.critic.parameters()):
            p.add_(0.05 * torch.randn(p.shape, generator=g))
        for p in net.target_critic.parameters():
            p.add_(0.08 * torch.randn(p.shape, generator=g))
    out = dict(P=P, T=T, N=N, D=D, A=A, H=H, centralised=int(centralised), grad_clip=0.5, actor_layers=np.array(actor_layers), critic_layers=np.array(critic_layers),
               actor_sharing=np.array(getattr(net.actor, "sharing_indices", list(range(P)))),
               critic_sharing=np.array(getattr(net.critic, "sharing_indices", list(range(P)))),
               actor_is_shared=int(hasattr(net.actor, "networks")), critic_is_shared=int(hasattr(net.critic, "networks")),
               keys=np.array(list(net.state_dict().keys())), actor0=flat(net.actor).numpy(), critic0=flat(net.critic).numpy(),
               target0=flat(net.target_critic).numpy())
    steps = [0, 250, 400]
    batches = [synthetic_batch(P, T, N, D, A, seed=seed + 100 + i) for i in range(3)]
    mk = lambda b: rat.Batch(b["obss"], b["actions"], b["rewards"], b["dones"], b["filled"], None)  # noqa: E731
    metrics = []
    for i, (b, st) in enumerate(zip(batches, steps)):
        m = net.update(mk(b), st)
        metrics.append([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]])
        out[f"actor{i + 1}"] = flat(net.actor).numpy()
        out[f"critic{i + 1}"] = flat(net.critic).numpy()
        out[f"target{i + 1}"] = flat(net.target_critic).numpy()
    out["metrics"] = np.array(metrics, np.float64)
    out["steps"] = np.array(steps)
    for i, b in enumerate(batches):
        for k, v in b.items():
            out[f"batch{i}_{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "actor nets", out["actor0"].shape[0], "critic nets", out["critic0"].shape[0], "metrics", np.array(metrics).round(5).tolist())


if __name__ == "__main__":
    torch.set_num_threads(1)
    import_reference()
    from marlbase.ac import model as ram
    from marlbase.ac import train as rat

    fixture(ram, rat, "learner_a2c_depths.npz", ram.A2CNetwork, P=2, D=15, H=64, N=12, seed=2300, actor_sharing=False, critic_sharing=False,
            actor_layers=[64, 64], critic_layers=[48, 64, 32])
    fixture(ram, rat, "learner_ppo_depths_p3.npz", ram.PPONetwork, P=3, D=18, H=64, N=10, seed=2400, actor_sharing=False, critic_sharing=True,
            centralised=True, actor_layers=[32, 48, 40], critic_layers=[64, 64])
    fixture(ram, rat, "learner_a2c_mixed_sharing_H64.npz", ram.A2CNetwork, P=2, D=15, H=64, N=12, seed=2100, actor_sharing=True, critic_sharing=False)
    fixture(ram, rat, "learner_ppo_mixed_sharing_p3_H64.npz", ram.PPONetwork, P=3, D=18, H=64, N=10, seed=2200, actor_sharing=[0, 0, 1],
            critic_sharing=True, centralised=True)
