"""Recurrent Q-networks (`use_rnn: True`, marlbase/utils/models.py:51-116): the oracle port against goldens produced by the
reference's own QNetwork / VDNetwork (CPU), the HIP path against the same goldens (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import gru_port as gp

G = os.path.join(os.path.dirname(__file__), "golden")
FILES = [("learner_gru_idqn_H64.npz", "idqn"), ("learner_gru_vdn_H64.npz", "vdn")]


def load(name):
    g = dict(np.load(os.path.join(G, name)))
    return g, {k[6:]: torch.tensor(v) for k, v in g.items() if k.startswith("batch_")}


@pytest.mark.parametrize("name,mode", FILES)
def test_oracle_port_matches_reference(name, mode):
    g, batch = load(name)
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    assert g["params0"].shape == (P, gp.nparams(D, H, A))
    assert list(g["keys"][:8]) == [f"critic.independent.0.{n}" for n in gp.NAMES]
    pr = torch.tensor(g["params0"]).requires_grad_(True)
    np.testing.assert_allclose(gp.q_values(pr.detach(), batch["obss"], D, H, A).numpy(), g["q0"], rtol=0, atol=2e-6)
    loss = gp.compute_loss(pr, torch.tensor(g["target0"]), batch, 0.99, True, D, H, A, mode=mode)
    loss.backward()
    assert abs(loss.item() - g["loss0"]) <= 1e-5 * abs(g["loss0"])
    np.testing.assert_allclose(pr.grad.numpy(), g["grad0"], rtol=1e-4, atol=1e-6)
    # act trace: greedy actions and the carried hidden states
    h = [torch.zeros(1, H) for _ in range(P)]
    for t in range(6):
        for p in range(P):
            q, h[p] = gp.cell(gp.split(torch.tensor(g["params0"][p]), D, H, A), torch.tensor(g["act_obs"][t, p])[None], h[p])
            top = torch.sort(q[0]).values
            if top[-1] - top[-2] > 1e-5:
                assert int(q.argmax()) == g["act_actions"][t, p]
            np.testing.assert_allclose(h[p][0].numpy(), g["act_hiddens"][t, p], rtol=0, atol=2e-6)
