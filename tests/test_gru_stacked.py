"""Stacked recurrent layers: `use_rnn` with layers = [h] * (L + 1) builds nn.GRU(num_layers = L) (marlbase/utils/models.py:74-90, any L).
Goldens from the reference's own QNetwork / VDNetwork / QMixNetwork (oracle/make_golden_gru.py: stacked()): the oracle port on the CPU,
the HIP path (csrc/gru_stack.h: the one-layer kernels run L times, chained through their activation records) on the GPU - through the
C-ABI at the exact width, through the driver classes (zero-padded onto the 64 / 128 kernels) at the others."""
import os

import numpy as np
import pytest
import torch

from oracle import gru_port as gp

G = os.path.join(os.path.dirname(__file__), "golden")
# (file, mode, L, kernel width)
FILES = [("learner_gru_idqn_L2_H64.npz", "idqn", 2, 64), ("learner_gru_vdn_L3_h40.npz", "vdn", 3, 64), ("learner_gru_idqn_L2_h72.npz", "idqn", 2, 128)]


def load(name):
    g = dict(np.load(os.path.join(G, name)))
    return g, {k[6:]: torch.tensor(v) for k, v in g.items() if k.startswith("batch_")}


@pytest.mark.parametrize("name,mode,L,Hk", FILES)
def test_oracle_port_matches_reference(name, mode, L, Hk):
    g, batch = load(name)
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    assert list(g["layers"]) == [H] * (L + 1)
    assert g["params0"].shape == (P, gp.nparams(D, H, A, L)) and gp.depth(torch.tensor(g["params0"][0]), D, H, A) == L
    assert list(g["keys"][:4 + 4 * L]) == [f"critic.independent.0.{n}" for n in gp.names(L)]
    pr = torch.tensor(g["params0"]).requires_grad_(True)
    np.testing.assert_allclose(gp.q_values(pr.detach(), batch["obss"], D, H, A).numpy(), g["q0"], rtol=0, atol=2e-6)
    loss = gp.compute_loss(pr, torch.tensor(g["target0"]), batch, 0.99, True, D, H, A, mode=mode)
    loss.backward()
    assert abs(loss.item() - g["loss0"]) <= 1e-5 * abs(g["loss0"])
    np.testing.assert_allclose(pr.grad.numpy(), g["grad0"], rtol=1e-4, atol=1e-6)
    # act trace: greedy actions and the carried hidden states ([num_layers, 1, H] per agent, utils/models.py:96-102)
    h = [torch.zeros(L, 1, H) for _ in range(P)]
    for t in range(6):
        for p in range(P):
            q, h[p] = gp.cell(gp.split(torch.tensor(g["params0"][p]), D, H, A), torch.tensor(g["act_obs"][t, p])[None], h[p])
            top = torch.sort(q[0]).values
            if top[-1] - top[-2] > 1e-5:
                assert int(q.argmax()) == g["act_actions"][t, p]
            np.testing.assert_allclose(h[p].reshape(-1).numpy(), g["act_hiddens"][t, p], rtol=0, atol=2e-6)


def test_oracle_qmix_port_with_stacked_recurrent_agents_matches_reference():
    g, batch = load("learner_gru_qmix_L2_H64.npz")
    D, H, A = int(g["D"]), int(g["H"]), int(g["A"])
    pr, mr = torch.tensor(g["params0"]).requires_grad_(True), torch.tensor(g["mixer0"]).requires_grad_(True)
    loss = gp.compute_qmix_loss(pr, torch.tensor(g["target0"]), mr, torch.tensor(g["tmixer0"]), batch, 0.99, True, D, H, A)
    loss.backward()
    assert abs(loss.item() - g["loss0"]) <= 1e-5 * abs(g["loss0"])
    np.testing.assert_allclose(pr.grad.numpy(), g["grad0"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(mr.grad.numpy(), g["mgrad0"], rtol=1e-4, atol=2e-6)


def test_host_layout_of_a_stack_is_the_references():
    """block views / padding / initial draws (dqn/model.py) for L layers: the reference's key order, and init_flat_gru_params consumes
    torch's RNG as RNNNetwork.__init__ does - checked against a torch module built the same way"""
    from codebase_amd.dqn import model as M

    D, h, A, L, Hk = 15, 40, 6, 3, 64
    torch.manual_seed(5)
    crit, targ = M.init_flat_gru_params([D, D], h, [A, A], True, None, num_layers=L)
    assert crit.shape == (2, gp.nparams(D, h, A, L)) and torch.equal(crit, targ)
    torch.manual_seed(5)
    first, rnn, final = torch.nn.Linear(D, h), torch.nn.GRU(h, h, num_layers=L), torch.nn.Linear(h, A)
    torch.nn.init.orthogonal_(final.weight.data, gain=np.sqrt(2))
    want = torch.cat([t.detach().reshape(-1) for t in [first.weight, first.bias] + list(rnn.parameters()) + [final.weight, final.bias]])
    assert torch.equal(crit[0, :want.numel() - A], want[:want.numel() - A])  # (the final bias is zeroed by the orthogonal init)
    names = [n for n, _ in M._gru_layout(D, h, A, L)]
    assert names == list(gp.names(L)) and [n for n, _, _ in M.gru_block_views(crit[0], D, h, A, h, L)] == names
    padded = M.pad_gru_blocks(crit, D, h, A, Hk, L)
    assert padded.shape == (2, gp.nparams(D, Hk, A, L))
    live = torch.cat([v.reshape(-1) for _, v, _ in M.gru_block_views(padded[1], D, h, A, Hk, L)])
    assert torch.equal(live, crit[1]) and float(padded.abs().sum()) == float(crit.abs().sum())
    assert M.recurrent_depth([h] * 4) == 3 and M.recurrent_width([h] * 4) == (h, 64)
    with pytest.raises(NotImplementedError):
        M.recurrent_width([64] * 6)  # five stacked layers: beyond csrc/gru_stack.h's GRU_MAX_LAYERS
    with pytest.raises(NotImplementedError):
        M.recurrent_width([64, 32, 64])


# ---- GPU --------------------------------------------------------------------------------------------------------------------------------
def _net(g, mode, cfg_extra=None):
    from codebase_amd.dqn import model as M
    from tests.test_gpu_layers import spaces

    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    cfg = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False, target_update_interval_or_tau=200)
    cfg.update(cfg_extra or {})
    obs_space, act_space = spaces(P, D, A)
    net = (M.VDNetwork if mode == "vdn" else M.QNetwork)(obs_space, act_space, cfg, [int(x) for x in g["layers"]], False, True, True, "cuda")
    L = len(g["layers"]) - 1
    net.params.copy_(M.pad_gru_blocks(torch.tensor(g["params0"]), D, H, A, net.spec.hidden, L))
    net.target_params.copy_(M.pad_gru_blocks(torch.tensor(g["target0"]), D, H, A, net.spec.hidden, L))
    return net, M


def _live(M, blocks, D, h, A, Hk, L):
    return torch.stack([torch.cat([v.reshape(-1) for _, v, _ in M.gru_block_views(blocks[p], D, h, A, Hk, L)]) for p in range(blocks.shape[0])])


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode,L,Hk", FILES)
def test_hip_stack_matches_reference(name, mode, L, Hk):
    """values of the whole batch, the sequence one step at a time with the caller carrying [L][P][B][H], loss + gradient (the live parts;
    the padding's gradient exactly zero), two update() calls, and the reference's act trace through QNetwork.act"""
    from codebase_amd import hip as h

    g, batch = load(name)
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    net, M = _net(g, mode)
    assert net.spec.hidden == Hk and net.rnn_layers == L and net.spec.n_hidden == L + 1 and h.gru_nparams(net.spec) == gp.nparams(D, Hk, A, L)
    assert list(net.state_dict().keys()) == list(g["keys"])
    obs = batch["obss"].cuda().contiguous()
    q = h.gru_forward(net.spec, net.params, obs)
    np.testing.assert_allclose(q.cpu().numpy(), g["q0"], rtol=0, atol=5e-6)
    hid, outs = None, []
    for t in range(obs.shape[1]):
        qt, hid = h.gru_forward(net.spec, net.params, obs[:, t:t + 1].contiguous(), h_in=hid, want_h=True)
        assert hid.shape == (L, P, obs.shape[2], Hk)
        outs.append(qt)
    np.testing.assert_allclose(torch.cat(outs, 1).cpu().numpy(), q.cpu().numpy(), rtol=0, atol=1e-6)
    # loss / gradient
    hb = h.Batch(*(batch[k].cuda().contiguous() for k in ("obss", "actions", "rewards", "dones", "filled")), None)
    loss, grad = h.gru_loss_grad(net.spec, net.params, net.target_params, hb, mode=1 if mode == "vdn" else 0)
    assert abs(loss.cpu().numpy()[0] - g["loss0"]) <= 3e-5 * abs(g["loss0"]) and loss.cpu().numpy()[1] == batch["filled"].sum().item()
    gref = g["grad0"]
    np.testing.assert_allclose(_live(M, grad.cpu(), D, H, A, Hk, L).numpy(), gref, rtol=3e-4, atol=3e-5 * max(1e-2, np.abs(gref).max()))
    pad = torch.ones_like(grad, dtype=torch.bool)
    for p in range(P):
        for _, view, _ in M.gru_block_views(pad[p], D, H, A, Hk, L):
            view.fill_(False)
    assert H == Hk or (int(pad.sum()) > 0 and float(grad[pad].abs().max()) == 0.0)
    _, g2 = h.gru_loss_grad(net.spec, net.params, net.target_params, hb, mode=1 if mode == "vdn" else 0)
    assert torch.equal(grad, g2)  # bitwise reproducible
    # the reference's act trace (dqn/model.py:94-116 with the hidden states of dqn/train.py:210-216)
    hid = net.init_hiddens(1)
    assert hid[0].shape == (L, 1, Hk)
    hp = [torch.zeros(L, 1, H) for _ in range(P)]  # the port beside it: where its two best values are closer than 1e-5 the argmax may differ
    for t in range(6):
        acts, hid = net.act([o for o in g["act_obs"][t]], hid, 0.0)
        got = torch.stack([x.reshape(L, Hk)[:, :H].reshape(-1) for x in hid]).cpu().numpy()
        np.testing.assert_allclose(got, g["act_hiddens"][t], rtol=0, atol=5e-6)
        for p in range(P):
            qp, hp[p] = gp.cell(gp.split(torch.tensor(g["params0"][p]), D, H, A), torch.tensor(g["act_obs"][t, p])[None], hp[p])
            top = torch.sort(qp[0]).values
            if top[-1] - top[-2] > 1e-5:
                assert acts[p] == g["act_actions"][t, p]
    # two updates
    b = h.Batch(*(batch[k] for k in ("obss", "actions", "rewards", "dones", "filled")), None)
    losses = [net.update(b)["loss"] for _ in range(2)]
    np.testing.assert_allclose(losses, g["losses"], rtol=5e-5)
    # (two Adam steps of lr 3e-4: an entry whose gradient sits at the rounding floor moves by up to lr either way - 2 of 129,324 entries of
    # the [72] * 3 golden are 8e-6 off, every other within 5e-6)
    np.testing.assert_allclose(_live(M, net.params.cpu(), D, H, A, Hk, L).numpy(), g["params2"], rtol=0, atol=1e-5)
    assert H == Hk or float(net.params[pad].abs().max()) == 0.0


@pytest.mark.gpu
def test_stacked_recurrent_qmix_matches_reference():
    """QMixNetwork(use_rnn=True, layers [64] * 3): loss, agent and mixer gradients, two update() calls"""
    from codebase_amd.dqn.model import QMixNetwork
    from codebase_amd.hip import Batch
    from codebase_amd.spaces import Box, Discrete, Tuple

    g, batch = load("learner_gru_qmix_L2_H64.npz")
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False,
                 target_update_interval_or_tau=200)
    net = QMixNetwork(Tuple([Box(-1, 8, (D,))] * P), Tuple([Discrete(A)] * P), hyper, [H, H, H], False, True, True,
                      dict(embed_dim=64, hypernet_layers=2, hypernet_embed=32), "cuda")
    assert list(net.state_dict().keys()) == list(g["keys"])
    net.params.copy_(torch.tensor(g["params0"]))
    net.target_params.copy_(torch.tensor(g["target0"]))
    net.mixer_params.copy_(torch.tensor(g["mixer0"]))
    net.target_mixer_params.copy_(torch.tensor(g["tmixer0"]))
    hb = Batch(*(batch[k].cuda().contiguous() for k in ("obss", "actions", "rewards", "dones", "filled")), None)
    loss, grad = net.updater.loss_grad(hb)
    assert abs(loss.cpu().numpy()[0] - g["loss0"]) <= 3e-5 * abs(g["loss0"])
    np.testing.assert_allclose(grad.cpu().numpy(), g["grad0"], rtol=3e-4, atol=3e-5 * max(1e-2, np.abs(g["grad0"]).max()))
    np.testing.assert_allclose(net.updater.mixer_grad.cpu().numpy(), g["mgrad0"], rtol=3e-4, atol=3e-5 * max(1e-2, np.abs(g["mgrad0"]).max()))
    b = Batch(*(batch[k] for k in ("obss", "actions", "rewards", "dones", "filled")), None)
    losses = [net.update(b)["loss"] for _ in range(2)]
    np.testing.assert_allclose(losses, g["losses"], rtol=5e-5)
    np.testing.assert_allclose(net.params.cpu().numpy(), g["params2"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(net.mixer_params.cpu().numpy(), g["mixer2"], rtol=0, atol=5e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("P,T,B,D,A,mode,H,L", [(2, 25, 70, 15, 6, "idqn", 64, 2), (4, 6, 16, 27, 6, "vdn", 64, 4), (3, 7, 130, 18, 6, "vdn", 128, 3),
                                                (4, 5, 20, 71, 5, "idqn", 128, 2), (8, 3, 17, 39, 6, "idqn", 64, 3)])
def test_hip_stack_other_shapes_vs_port(P, T, B, D, A, mode, H, L):
    """depths up to the maximum, ragged batches (B not a multiple of 16 / 64), both kernel widths - loss and gradient against the port"""
    from codebase_amd import hip as h
    from oracle import dqn_port as dp

    gen = torch.Generator().manual_seed(100 * L + P)
    params = 0.12 * torch.randn(P, gp.nparams(D, H, A, L), generator=gen)
    target = params + 0.05 * torch.randn(P, gp.nparams(D, H, A, L), generator=gen)
    batch = dp.synthetic_batch(P, T, B, D, A, seed=7 + L)
    batch["obss"] = batch["obss"] * 0.25
    if mode == "vdn":
        batch["rewards"][1:] = batch["rewards"][0]
    pr = params.clone().requires_grad_(True)
    ref = gp.compute_loss(pr, target, batch, 0.99, True, D, H, A, mode=mode)
    ref.backward()
    spec = h.NetSpec(P, D, H, A, n_hidden=L + 1)
    hb = h.Batch(*(batch[k].cuda().contiguous() for k in ("obss", "actions", "rewards", "dones", "filled")), None)
    loss, grad = h.gru_loss_grad(spec, params.cuda(), target.cuda(), hb, mode=1 if mode == "vdn" else 0)
    assert abs(loss.cpu().numpy()[0] - ref.item()) <= 3e-5 * abs(ref.item())
    gref = pr.grad.numpy()
    np.testing.assert_allclose(grad.cpu().numpy(), gref, rtol=3e-4, atol=3e-5 * max(1e-2, np.abs(gref).max()))


@pytest.mark.gpu
def test_stacked_recurrent_idqn_trains_through_the_entry_point(tmp_path, monkeypatch):
    """`+algorithm=idqn algorithm.model.use_rnn=True algorithm.model.layers=[64,64,64]` through codebase_amd.run: the modular collection loop
    carries [L][P][N][H] between the steps, the learner runs the stack, evaluation and the checkpoint go through the same classes"""
    from codebase_amd import run

    monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path))
    df = run.main(["+algorithm=idqn", "env.name=lbforaging:Foraging-8x8-2p-3f-v3", "env.time_limit=25", "env.parallel_envs=64", "seed=1",
                   "algorithm.total_steps=30000", "algorithm.eval_interval=10000", "algorithm.model.use_rnn=True", "algorithm.model.layers=[64,64,64]"])
    assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all() and np.isfinite(df["mean_episode_returns"]).all()
