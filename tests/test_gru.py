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


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode", FILES)
def test_hip_forward_matches_reference(name, mode):
    from codebase_amd import hip as h

    g, batch = load(name)
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    spec = h.NetSpec(P, D, H, A)
    params = torch.tensor(g["params0"]).cuda()
    q = h.gru_forward(spec, params, batch["obss"].cuda().contiguous())
    np.testing.assert_allclose(q.cpu().numpy(), g["q0"], rtol=0, atol=5e-6)
    # one step at a time with the hidden state carried by the caller == the whole sequence at once
    hid, outs = None, []
    for t in range(batch["obss"].shape[1]):
        qt, hid = h.gru_forward(spec, params, batch["obss"][:, t:t + 1].cuda().contiguous(), h_in=hid, want_h=True)
        outs.append(qt)
    np.testing.assert_allclose(torch.cat(outs, 1).cpu().numpy(), q.cpu().numpy(), rtol=0, atol=1e-6)
    # the reference's act trace: greedy actions + hidden states
    hid = None
    for t in range(6):
        qt, hid = h.gru_forward(spec, params, torch.tensor(g["act_obs"][t]).reshape(P, 1, 1, D).cuda(), h_in=hid, want_h=True)
        np.testing.assert_allclose(hid[:, 0].cpu().numpy(), g["act_hiddens"][t], rtol=0, atol=5e-6)
        qn = qt[:, 0, 0].cpu().numpy()
        for p in range(P):
            top = np.sort(qn[p])
            if top[-1] - top[-2] > 1e-5:
                assert int(qn[p].argmax()) == g["act_actions"][t, p]


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode", FILES)
def test_hip_loss_and_gradient_match_reference(name, mode):
    from codebase_amd import hip as h

    g, batch = load(name)
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    spec = h.NetSpec(P, D, H, A)
    assert h.gru_nparams(spec) == g["params0"].shape[1]
    hb = h.Batch(*(batch[k].cuda().contiguous() for k in ("obss", "actions", "rewards", "dones", "filled")), None)
    loss, grad = h.gru_loss_grad(spec, torch.tensor(g["params0"]).cuda(), torch.tensor(g["target0"]).cuda(), hb, mode=1 if mode == "vdn" else 0)
    assert abs(loss.cpu().numpy()[0] - g["loss0"]) <= 3e-5 * abs(g["loss0"])
    assert loss.cpu().numpy()[1] == batch["filled"].sum().item()
    gref = g["grad0"]
    np.testing.assert_allclose(grad.cpu().numpy(), gref, rtol=3e-4, atol=3e-5 * max(1e-2, np.abs(gref).max()))
    # bitwise reproducible
    g1 = grad.clone()
    _, g2 = h.gru_loss_grad(spec, torch.tensor(g["params0"]).cuda(), torch.tensor(g["target0"]).cuda(), hb, mode=1 if mode == "vdn" else 0)
    assert torch.equal(g1, g2)


@pytest.mark.gpu
@pytest.mark.parametrize("P,T,B,D,A,mode,dq", [(2, 25, 70, 15, 6, "idqn", True), (4, 6, 16, 27, 6, "vdn", False), (8, 5, 33, 39, 6, "idqn", True),
                                               (4, 12, 20, 71, 5, "idqn", False), (2, 1, 1, 12, 6, "vdn", True)])
def test_hip_loss_and_gradient_other_shapes_vs_port(P, T, B, D, A, mode, dq):
    from codebase_amd import hip as h
    from oracle import dqn_port as dp

    H = 64
    gen = torch.Generator().manual_seed(P * 100 + D)
    params = 0.15 * torch.randn(P, gp.nparams(D, H, A), generator=gen)
    target = params + 0.05 * torch.randn(P, gp.nparams(D, H, A), generator=gen)
    batch = dp.synthetic_batch(P, T, B, D, A, seed=5)
    batch["obss"] = batch["obss"] * 0.25
    if mode == "vdn":
        batch["rewards"][1:] = batch["rewards"][0]
    pr = params.clone().requires_grad_(True)
    ref = gp.compute_loss(pr, target, batch, 0.99, dq, D, H, A, mode=mode)
    ref.backward()
    spec = h.NetSpec(P, D, H, A)
    hb = h.Batch(*(batch[k].cuda().contiguous() for k in ("obss", "actions", "rewards", "dones", "filled")), None)
    loss, grad = h.gru_loss_grad(spec, params.cuda(), target.cuda(), hb, double_q=dq, mode=1 if mode == "vdn" else 0)
    assert abs(loss.cpu().numpy()[0] - ref.item()) <= 5e-5 * max(abs(ref.item()), 1e-3)
    gref = pr.grad.numpy()
    np.testing.assert_allclose(grad.cpu().numpy(), gref, rtol=5e-4, atol=5e-5 * max(1e-2, np.abs(gref).max()))
