"""tests/golden/ref_checkpoint_*.pt: checkpoints written exactly as the reference writes them
(`torch.save(model.state_dict(), "checkpoints/model_s{step}.pt")`, marlbase/dqn/train.py:340-343, ac/train.py:205-208) by the
reference's own QNetwork / QMixNetwork / A2CNetwork, plus probe inputs and the reference's outputs on them.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_ckpt"""
import contextlib
import io
import os

import numpy as np
import torch

from .make_golden import OUT, Box, Cfg, Discrete, import_reference


def main():
    torch.set_num_threads(1)
    rm, _ = import_reference()
    from marlbase.ac import model as ram

    P, D, A, H = 2, 15, 6, 64
    cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, target_update_interval_or_tau=200, double_q=True,
              standardise_returns=False, n_steps=5, entropy_coef=0.001, value_loss_coef=0.5)
    g = torch.Generator().manual_seed(4)
    obs = torch.randint(-1, 8, (P, 32, D), generator=g).float()
    out = dict(obs=obs.numpy())
    spaces = ([Box(D)] * P, [Discrete(A)] * P)
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(21)
        q = rm.QNetwork(*spaces, cfg, [H, H], False, False, True, "cpu")
        torch.manual_seed(22)
        qm = rm.QMixNetwork(*spaces, cfg, [H, H], True, False, True, dict(embed_dim=64, hypernet_layers=2, hypernet_embed=32), "cpu")
        torch.manual_seed(23)
        net = Cfg(layers=[H, H], parameter_sharing=False, use_orthogonal_init=True, use_rnn=False)
        ac = ram.A2CNetwork(*spaces, cfg, net, Cfg(dict(net, centralised=False)), "cpu")
    for name, m in (("idqn", q), ("qmix_shared", qm), ("ia2c", ac)):
        with torch.no_grad():
            for p_ in m.parameters():
                p_.add_(0.03 * torch.randn(p_.shape, generator=g))
        torch.save(m.state_dict(), os.path.join(OUT, f"ref_checkpoint_{name}.pt"))
    with torch.no_grad():
        out["idqn_q"] = torch.stack(q.critic([obs[p].unsqueeze(0) for p in range(P)], None)[0]).squeeze(1).numpy()
        out["qmix_q"] = torch.stack(qm.critic([obs[p].unsqueeze(0) for p in range(P)], None)[0]).squeeze(1).numpy()
        out["qmix_mixer"] = torch.cat([p_.reshape(-1) for p_ in qm.mixer.parameters()]).numpy()
        out["ia2c_value"] = ac.get_value([obs[p] for p in range(P)], None)[0].numpy()
        out["ia2c_logits"] = torch.stack(ac.actor([obs[p].unsqueeze(0) for p in range(P)], None)[0]).squeeze(1).numpy()
    np.savez_compressed(os.path.join(OUT, "ref_checkpoint_probe.npz"), **out)
    print("checkpoints written:", [f for f in os.listdir(OUT) if f.startswith("ref_checkpoint")])


if __name__ == "__main__":
    main()
