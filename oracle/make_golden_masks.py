"""Generate tests/golden/learner_masks_*.npz from the REFERENCE's own QNetwork / VDNetwork (marlbase/dqn/model.py:118-163,
224-269) with `batch.action_mask` set, and the reference's masked `act` (dqn/model.py:94-116).
Runs only in the build container (needs /root/reference); the vectors travel, the reference does not.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_masks

Each file: critic / target blocks, one Batch with a random action mask ([P][T+1][B][A] f32, at least one allowed action per
row, the taken actions always allowed), loss and gradient of _compute_loss with double_q on and off, 2 x update().
  learner_masks_idqn_H64.npz   2 agents x 15 obs, QNetwork
  learner_masks_vdn_H128.npz   3 agents x 18 obs, VDNetwork, 128-128
plus `act_*`: observations, masks and the greedy actions QNetwork.act returns with epsilon 0.
"""
import contextlib
import io
import os

import numpy as np
import torch

from .dqn_port import synthetic_batch
from .make_golden import OUT, Box, Cfg, Discrete, flat_params, import_reference


def random_mask(P, T, B, A, actions, seed):
    g = torch.Generator().manual_seed(seed)
    m = (torch.rand(P, T + 1, B, A, generator=g) < 0.6).float()
    m[..., 0] = torch.maximum(m[..., 0], (m.sum(-1) == 0).float())  # never an empty row
    m[:, :-1].scatter_(-1, actions.unsqueeze(-1), 1.0)                 # the stored action was allowed when it was taken
    return m


def fixture(ref_model, ref_train, name, cls, P, D, H, B, seed):
    T, A = 12, 6
    out = dict(P=P, T=T, B=B, D=D, A=A, H=H)
    batch = synthetic_batch(P, T, B, D, A, seed=seed + 7)
    if cls is ref_model.VDNetwork:
        batch["rewards"][1:] = batch["rewards"][0]
    mask = random_mask(P, T, B, A, batch["actions"], seed + 9)
    for k, v in batch.items():
        out[f"batch_{k}"] = v.numpy()
    out["batch_action_mask"] = mask.numpy()
    bb = ref_train.Batch(batch["obss"], batch["actions"], batch["rewards"], batch["dones"], batch["filled"], mask)
    for dq in (True, False):
        torch.manual_seed(seed)
        cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, target_update_interval_or_tau=200, double_q=dq,
                  standardise_returns=False)
        with contextlib.redirect_stdout(io.StringIO()):
            net = cls([Box(D)] * P, [Discrete(A)] * P, cfg, [H, H], False, False, True, "cpu")
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for p in net.critic.parameters():
                p.add_(0.05 * torch.randn(p.shape, generator=g))
            for p in net.target.parameters():
                p.add_(0.08 * torch.randn(p.shape, generator=g))
        tag = "dq" if dq else "max"
        out["params0"], out["target0"] = flat_params(net.critic).numpy(), flat_params(net.target).numpy()
        loss = net._compute_loss(bb)
        net.optimizer.zero_grad()
        loss.backward()
        out[f"loss_{tag}"] = np.float32(loss.item())
        out[f"grad_{tag}"] = torch.stack([torch.cat([p.grad.reshape(-1) for p in m.parameters()]) for m in net.critic.independent]).numpy()
        # the same loss WITHOUT the mask must differ (the fixture exercises the masking)
        out[f"loss_nomask_{tag}"] = np.float32(net._compute_loss(bb._replace(action_mask=None)).item())
        net.optimizer.zero_grad()
        if dq:
            losses = [net.update(bb)["loss"] for _ in range(2)]
            out["losses"] = np.array(losses, np.float32)
            out["params2"] = flat_params(net.critic).numpy()
            # masked greedy act (epsilon 0): value * mask + (1 - mask) * -1e8 -> argmax
            ga = torch.Generator().manual_seed(seed + 3)
            obs = torch.randint(-1, 8, (20, P, D), generator=ga).float()
            am = (torch.rand(20, P, A, generator=ga) < 0.5).float()
            am[..., 5] = torch.maximum(am[..., 5], (am.sum(-1) == 0).float())
            acts = [net.act([o.numpy() for o in obs[i]], None, 0.0, [m for m in am[i]])[0] for i in range(20)]
            out["act_obs"], out["act_mask"], out["act_actions"] = obs.numpy(), am.numpy(), np.array(acts, np.int64)
            out["act_params"] = flat_params(net.critic).numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, {k: float(out[k]) for k in ("loss_dq", "loss_nomask_dq", "loss_max", "loss_nomask_max")})


if __name__ == "__main__":
    torch.set_num_threads(1)
    import_reference()
    from marlbase.dqn import model as rm
    from marlbase.dqn import train as rt

    fixture(rm, rt, "learner_masks_idqn_H64.npz", rm.QNetwork, P=2, D=15, H=64, B=37, seed=1100)
    fixture(rm, rt, "learner_masks_vdn_H128.npz", rm.VDNetwork, P=3, D=18, H=128, B=21, seed=1200)
