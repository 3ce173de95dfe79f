"""Generate tests/golden/learner_shared_*.npz from the REFERENCE's QNetwork / VDNetwork with `parameter_sharing`
(marlbase/utils/models.py:176-300 MultiAgentSharedNetwork; dqn/model.py:34-58).  Build container only.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_sharing

  learner_shared_H64.npz      parameter_sharing=True: 2 agents, ONE 15-64-64-6 network; IDQN loss, gradient, 3 updates
  learner_shared_seps_H64.npz parameter_sharing=[0, 0, 1] (SePS): 3 agents x 18 obs, 2 networks; VDN loss, gradient, 3 updates
Blocks are [K][n] with K = number of distinct networks, in `critic.networks` order; `keys` = state_dict key order.
"""
import contextlib
import io
import os

import numpy as np
import torch

from .dqn_port import synthetic_batch
from .make_golden import OUT, Box, Cfg, Discrete, import_reference


def flat_nets(net):
    return torch.stack([torch.cat([p.detach().reshape(-1) for p in m.parameters()]) for m in net.networks])


def fixture(ref_model, ref_train, name, cls, P, D, sharing, seed, B=32):
    T, A, H = 25, 6, 64
    torch.manual_seed(seed)
    cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, target_update_interval_or_tau=2, double_q=True,
              standardise_returns=False)
    with contextlib.redirect_stdout(io.StringIO()):
        net = cls([Box(D)] * P, [Discrete(A)] * P, cfg, [H, H], sharing, False, True, "cpu")
    init = flat_nets(net.critic).numpy().copy()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in net.critic.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
        for p in net.target.parameters():
            p.add_(0.08 * torch.randn(p.shape, generator=g))
    idx = net.critic.sharing_indices
    out = dict(P=P, T=T, B=B, D=D, A=A, H=H, sharing=np.array(idx), init=init, params0=flat_nets(net.critic).numpy(),
               target0=flat_nets(net.target).numpy(), keys=np.array(list(net.state_dict().keys())))
    batches = [synthetic_batch(P, T, B, D, A, seed=seed + 100 + i) for i in range(3)]
    if cls is ref_model.VDNetwork:
        for b in batches:
            b["rewards"][1:] = b["rewards"][0]
    mk = lambda b: ref_train.Batch(b["obss"], b["actions"], b["rewards"], b["dones"], b["filled"], None)  # noqa: E731
    loss = net._compute_loss(mk(batches[0]))
    net.optimizer.zero_grad()
    loss.backward()
    out["loss0"] = np.float32(loss.item())
    out["grad0"] = torch.stack([torch.cat([p.grad.reshape(-1) for p in m.parameters()]) for m in net.critic.networks]).numpy()
    net.optimizer.zero_grad()
    losses = []
    for i, b in enumerate(batches):
        losses.append(net.update(mk(b))["loss"])
        out[f"params{i + 1}"] = flat_nets(net.critic).numpy()
        out[f"target{i + 1}"] = flat_nets(net.target).numpy()
    out["losses"] = np.array(losses, np.float32)
    for i, b in enumerate(batches):
        for k, v in b.items():
            out[f"batch{i}_{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "sharing", idx, "loss0", out["loss0"], "losses", losses)


if __name__ == "__main__":
    torch.set_num_threads(1)
    rm, rt = import_reference()
    fixture(rm, rt, "learner_shared_H64.npz", rm.QNetwork, 2, 15, True, seed=800)
    fixture(rm, rt, "learner_shared_seps_H64.npz", rm.VDNetwork, 3, 18, [0, 0, 1], seed=900, B=20)
