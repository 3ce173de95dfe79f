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


# ---- actor-critic learners: both families recurrent at one depth (ac/model.py:45-97 with use_rnn and layers [h] * (L + 1)) ---------------
AC_FILES = [("learner_a2c_gru_L2_h24.npz", 2), ("learner_mappo_gru_L3_p3_h40.npz", 3)]
AC_TWO_DEPTHS = "learner_a2c_gru_L1_L3_h24.npz"  # actor.layers [24, 24], critic.layers [24] * 4: each family from its own list (ac/model.py:45-97)


def _ac_batch(g, i):
    return {k: torch.tensor(g[f"batch{i}_{k}"]) for k in ("obss", "actions", "rewards", "dones", "filled")}


@pytest.mark.parametrize("name,L", AC_FILES)
def test_ac_oracle_port_with_stacked_recurrent_networks_matches_reference(name, L):
    from oracle import ac_update_port as ap

    g = dict(np.load(os.path.join(G, name)))
    D, H, A = int(g["D"]), int(g["H"]), int(g["A"])
    assert list(g["layers"]) == [H] * (L + 1) and g["actor0"].shape[1] == gp.nparams(D, H, A, L)
    with gp.recurrent_ac(L):
        lr = ap.Learner(torch.tensor(g["actor0"]), torch.tensor(g["critic0"]), D, H, A, gamma=float(g["gamma"]), n_steps=int(g["n_steps"]),
                        entropy_coef=float(g["entropy_coef"]), value_loss_coef=float(g["value_loss_coef"]),
                        num_epochs=int(g["num_epochs"]) if "ppo" in name else 0, ppo_clip=float(g["ppo_clip"]))
        lr.target = torch.tensor(g["target0"])
        for i in range(3):
            m = lr.update(_ac_batch(g, i), int(g["steps"][i]))
            np.testing.assert_allclose([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]], g["metrics"][i], rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(lr.actor().detach().numpy(), g[f"actor{i + 1}"], rtol=0, atol=3e-6)
            np.testing.assert_allclose(lr.critic().detach().numpy(), g[f"critic{i + 1}"], rtol=0, atol=3e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name,L", AC_FILES)
def test_hip_stacked_recurrent_actor_critic_matches_reference(name, L):
    """A2CNetwork / PPONetwork (centralised critics in the second) built with use_rnn and L + 1 layer sizes: the reference's state_dict
    keys and shapes, acting / values with [num_layers, N, h]-shaped hidden states, metrics and the live parts of all three blocks after 3
    updates (the blocks are zero-padded onto the 64-wide kernels: the padding stays exactly zero)"""
    from collections import namedtuple

    from codebase_amd.ac.model import A2CNetwork, PPONetwork
    from codebase_amd.dqn import model as M
    from codebase_amd.spaces import Box, Discrete, Tuple

    Batch = namedtuple("Batch", ["obss", "actions", "rewards", "dones", "filled", "action_masks"])
    g = dict(np.load(os.path.join(G, name)))
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    ppo, cen = "ppo" in name, "mappo" in name or "maa2c" in name
    cfg = dict(optimizer="Adam", lr=3e-4, gamma=float(g["gamma"]), grad_clip=False, n_steps=int(g["n_steps"]), entropy_coef=float(g["entropy_coef"]),
               value_loss_coef=float(g["value_loss_coef"]), standardise_returns=False, target_update_interval_or_tau=200,
               num_epochs=int(g["num_epochs"]), ppo_clip=float(g["ppo_clip"]))
    net_cfg = dict(layers=[H] * (L + 1), parameter_sharing=False, use_orthogonal_init=True, use_rnn=True)
    net = (PPONetwork if ppo else A2CNetwork)(Tuple([Box(-1, 8, (D,))] * P), Tuple([Discrete(A)] * P), cfg, dict(net_cfg), dict(net_cfg, centralised=cen), "cuda")
    Hk = net.spec.hidden
    assert net.recurrent and net.rnn_layers == {"actor": L, "critic": L, "target_critic": L} and Hk == 64 and net.spec.n_hidden == L + 1
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["state_dict_keys"]]
    assert sd[f"actor.independent.0.rnn.weight_hh_l{L - 1}"].shape == (3 * H, H) and sd[f"critic.independent.1.rnn.bias_ih_l{L - 1}"].shape == (3 * H,)
    dc = P * D if cen else D
    net.actor_params.copy_(M.pad_gru_blocks(torch.tensor(g["actor0"]), D, H, A, Hk, L))
    net.critic_params.copy_(M.pad_gru_blocks(torch.tensor(g["critic0"]), dc, H, 1, Hk, L))
    net.target_critic_params.copy_(M.pad_gru_blocks(torch.tensor(g["target0"]), dc, H, 1, Hk, L))
    # acting / values one step at a time == the port's sequence (zero initial hidden states), hidden states in the reference's shape
    obs = torch.tensor(g["batch0_obss"][:4])  # [4][N][P*D]
    N = obs.shape[1]
    ah, ch = net.init_actor_hiddens(N), net.init_critic_hiddens(N)
    assert ah[0].shape == (L, N, Hk) and ch[0].shape == (L, N, Hk)
    vals = []
    for t in range(4):
        per_agent = [obs[t, :, p * D:(p + 1) * D] for p in range(P)]
        acts, ah = net.act(per_agent, ah)
        v, ch = net.get_value(per_agent, ch)
        assert acts.shape == (P, N, 1) and v.shape == (N, P) and ah[1].shape == (L, N, Hk)
        vals.append(v.cpu())
    for p in range(P):
        x = obs[:4] if cen else obs[:4, :, p * D:(p + 1) * D]
        want, hT = gp.sequence(torch.tensor(g["critic0"][p]), x, dc, H, 1)
        np.testing.assert_allclose(torch.stack(vals)[:, :, p].numpy(), want[..., 0].numpy(), rtol=0, atol=5e-6)
        np.testing.assert_allclose(ch[p][:, :, :H].cpu().numpy(), hT.numpy(), rtol=0, atol=5e-6)
        assert float(ch[p][:, :, H:].abs().max()) == 0.0
    for i in range(3):
        b = Batch(*(x.cuda() for x in _ac_batch(g, i).values()), None)
        m = net.update(b._replace(dones=b.dones.float()), int(g["steps"][i]))
        np.testing.assert_allclose([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]], g["metrics"][i], rtol=1e-4, atol=1e-5)
        for got, want, d, a in ((net.actor_params, g[f"actor{i + 1}"], D, A), (net.critic_params, g[f"critic{i + 1}"], dc, 1),
                                (net.target_critic_params, g[f"target{i + 1}"], dc, 1)):
            diff = np.abs(_live(M, got.cpu(), d, H, a, Hk, L).numpy() - want)
            assert diff.max() <= 5e-5 and (diff > 5e-6).mean() <= 1e-4, (i, diff.max(), (diff > 5e-6).sum())
    pad = torch.ones_like(net.actor_params, dtype=torch.bool)
    for p in range(P):
        for _, view, _ in M.gru_block_views(pad[p], D, H, A, Hk, L):
            view.fill_(False)
    assert int(pad.sum()) > 0 and float(net.actor_params[pad].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("P,T,N,D,H,L,cen,ppo", [(2, 9, 20, 15, 64, 2, False, False), (3, 6, 33, 18, 128, 2, True, True), (4, 7, 16, 27, 64, 4, True, False),
                                                  (2, 5, 40, 71, 128, 3, False, True)])
def test_hip_stacked_recurrent_actor_critic_other_shapes_vs_port(P, T, N, D, H, L, cen, ppo):
    """exact kernel widths (64: LDS-resident, 128: streamed gate matrices), depths up to 4, independent and centralised critics, A2C and PPO:
    metrics and both gradients of one step against the port"""
    from collections import namedtuple

    from codebase_amd import hip as h
    from oracle import ac_update_port as ap

    Batch = namedtuple("Batch", ["obss", "actions", "rewards", "dones", "filled", "action_masks"])
    A = 5 if D == 71 else 6
    gen = torch.Generator().manual_seed(31 * L + P)
    dc = P * D if cen else D
    actor = 0.1 * torch.randn(P, gp.nparams(D, H, A, L), generator=gen)
    critic = 0.1 * torch.randn(P, gp.nparams(dc, H, 1, L), generator=gen)
    target = critic + 0.05 * torch.randn(P, gp.nparams(dc, H, 1, L), generator=gen)
    batch = ap.synthetic_batch(P, T, N, D, A, seed=5 + L)
    with gp.recurrent_ac(L):
        lr = ap.Learner(actor, critic, D, H, A, gamma=0.99, n_steps=5, entropy_coef=0.001, value_loss_coef=0.5, num_epochs=1 if ppo else 0, ppo_clip=0.2)
        lr.target = target.clone()
        m_ref = lr.update(batch, 1)
    spec = h.NetSpec(P, D, H, A, n_hidden=L + 1)
    block = torch.cat([actor.reshape(-1), critic.reshape(-1)]).cuda()
    up = h.AcUpdater(spec, block, target.cuda().contiguous(), lr=3e-4, gamma=0.99, n_steps=5, entropy_coef=0.001, value_loss_coef=0.5, grad_clip=False,
                     ppo_clip=0.2, recurrent=True, centralised_critic=cen)
    assert up.n_actor == actor.shape[1] and up.n_critic == critic.shape[1]
    b = Batch(*(batch[k].cuda() for k in ("obss", "actions", "rewards", "dones", "filled")), None)
    if ppo:
        up.ppo_prepare(b)
        m = up.ppo_loss_grad(b).cpu().numpy()[:4]
    else:
        m = up.a2c_loss_grad(b).cpu().numpy()[:4]
    np.testing.assert_allclose(m, [m_ref["loss"], m_ref["actor_loss"], m_ref["value_loss"], m_ref["entropy"]], rtol=1e-4, atol=1e-5)
    up.apply()
    for got, want in ((up.block[:P * up.n_actor], lr.actor().detach()), (up.block[P * up.n_actor:], lr.critic().detach())):
        diff = np.abs(got.cpu().numpy().reshape(P, -1) - want.numpy())
        assert diff.max() <= 3.1e-4 and (diff > 5e-6).mean() <= 2e-3, (diff.max(), (diff > 5e-6).mean())  # one Adam step: |step| <= lr


@pytest.mark.gpu
def test_stacked_recurrent_ia2c_and_ippo_end_to_end(tmp_path, monkeypatch):
    """`+algorithm=ia2c|ippo` with actor / critic use_rnn and layers [64,64,64] through codebase_amd.run (the modular rollout carries the
    actors' [L][P][N][H] between the steps)"""
    from codebase_amd import run

    for algo in ("ia2c", "ippo"):
        monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / algo))
        df = run.main([f"+algorithm={algo}", "env.name=lbforaging:Foraging-8x8-2p-3f-v3", "env.time_limit=25", "env.parallel_envs=64", "seed=1",
                       "algorithm.model.actor.layers=[64,64,64]", "algorithm.model.critic.layers=[64,64,64]", "algorithm.model.actor.use_rnn=True",
                       "algorithm.model.critic.use_rnn=True", "algorithm.total_steps=30000", "algorithm.eval_interval=10000"])
        assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all() and np.isfinite(df["mean_episode_returns"]).all()


# ---- a stack together with parameter sharing / standardise_returns (the reference's own classes again) ------------------------------------
@pytest.mark.gpu
def test_stacked_recurrent_vdn_with_seps_sharing_matches_reference():
    """VDNetwork(use_rnn, layers [24] * 3, parameter_sharing=[0, 0, 1]): two networks for three agents, each agent with its own hidden
    state through the network of its group; the shared network's gradient is the sum over its agents (MultiAgentSharedNetwork)"""
    from codebase_amd import hip as h
    from codebase_amd.dqn import model as M
    from codebase_amd.spaces import Box, Discrete, Tuple

    g, batch = load("learner_gru_seps_vdn_L2_h24.npz")
    P, D, H, A, L = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"]), 2
    sharing = [int(x) for x in g["sharing"]]
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False, target_update_interval_or_tau=200)
    net = M.VDNetwork(Tuple([Box(-1, 8, (D,))] * P), Tuple([Discrete(A)] * P), hyper, [H] * (L + 1), sharing, True, True, "cuda")
    Hk = net.spec.hidden
    assert list(net.state_dict().keys()) == list(g["keys"]) and net.params.shape[0] == g["params0"].shape[0] == 2
    net.params.copy_(M.pad_gru_blocks(torch.tensor(g["params0"]), D, H, A, Hk, L))
    net.target_params.copy_(M.pad_gru_blocks(torch.tensor(g["target0"]), D, H, A, Hk, L))
    hb = h.Batch(*(batch[k].cuda().contiguous() for k in ("obss", "actions", "rewards", "dones", "filled")), None)
    loss, grad = net.updater.loss_grad(hb, mode=net.mode)
    assert abs(loss.cpu().numpy()[0] - g["loss0"]) <= 3e-5 * abs(g["loss0"])
    np.testing.assert_allclose(_live(M, grad.cpu(), D, H, A, Hk, L).numpy(), g["grad0"], rtol=3e-4, atol=3e-5 * max(1e-2, np.abs(g["grad0"]).max()))
    b = h.Batch(*(batch[k] for k in ("obss", "actions", "rewards", "dones", "filled")), None)
    np.testing.assert_allclose([net.update(b)["loss"] for _ in range(2)], g["losses"], rtol=5e-5)
    np.testing.assert_allclose(_live(M, net.params.cpu(), D, H, A, Hk, L).numpy(), g["params2"], rtol=0, atol=1e-5)


@pytest.mark.gpu
def test_stacked_recurrent_idqn_with_standardise_returns_matches_reference():
    """QNetwork(use_rnn, layers [24] * 3, standardise_returns=True): 3 updates on 3 batches - losses, running statistics, parameters"""
    from codebase_amd import hip as h
    from codebase_amd.dqn import model as M
    from codebase_amd.spaces import Box, Discrete, Tuple

    g = dict(np.load(os.path.join(G, "learner_gru_std_L2_h24.npz")))
    P, D, H, A, L = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"]), 2
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=True, target_update_interval_or_tau=200)
    net = M.QNetwork(Tuple([Box(-1, 8, (D,))] * P), Tuple([Discrete(A)] * P), hyper, [H] * (L + 1), False, True, True, "cuda")
    Hk = net.spec.hidden
    net.params.copy_(M.pad_gru_blocks(torch.tensor(g["params0"]), D, H, A, Hk, L))
    net.target_params.copy_(M.pad_gru_blocks(torch.tensor(g["target0"]), D, H, A, Hk, L))
    for i in range(3):
        b = h.Batch(*(torch.tensor(g[f"batch{i}_{k}"]) for k in ("obss", "actions", "rewards", "dones", "filled")), None)
        loss = net.update(b)["loss"]
        assert abs(loss - g["losses"][i]) <= 5e-5 * abs(g["losses"][i])
        np.testing.assert_allclose(net.ret_ms.mean.cpu().numpy(), g[f"ret_mean{i + 1}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(net.ret_ms.var.cpu().numpy(), g[f"ret_var{i + 1}"], rtol=1e-5, atol=1e-6)
        assert abs(net.ret_ms.count - float(g[f"ret_count{i + 1}"])) < 1e-6
        np.testing.assert_allclose(_live(M, net.params.cpu(), D, H, A, Hk, L).numpy(), g[f"params{i + 1}"], rtol=0, atol=1e-5)


@pytest.mark.gpu
def test_stacked_recurrent_actor_critic_with_sharing_vs_port_free_run(tmp_path, monkeypatch):
    """ia2c with both families recurrent, three layer sizes and parameter sharing in both, through codebase_amd.run on the warehouse's
    71-wide rows (the recurrent centralised / shared maps travel through the same AgentMap that now carries the depth)"""
    from codebase_amd import run

    monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path))
    df = run.main(["+algorithm=ia2c", "env.name=rware:rware-tiny-2ag-v2", "env.time_limit=40", "env.parallel_envs=64", "seed=1",
                   "algorithm.model.actor.layers=[64,64,64]", "algorithm.model.critic.layers=[64,64,64]", "algorithm.model.actor.use_rnn=True",
                   "algorithm.model.critic.use_rnn=True", "algorithm.model.actor.parameter_sharing=True", "algorithm.model.critic.parameter_sharing=True",
                   "algorithm.total_steps=30000", "algorithm.eval_interval=10000"])
    assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all() and np.isfinite(df["mean_episode_returns"]).all()


def test_ac_oracle_port_with_recurrent_families_of_two_depths_matches_reference():
    from oracle import ac_update_port as ap

    g = dict(np.load(os.path.join(G, AC_TWO_DEPTHS)))
    D, H, A = int(g["D"]), int(g["H"]), int(g["A"])
    La, Lc = len(g["layers"]) - 1, len(g["critic_layers"]) - 1
    assert (La, Lc) == (1, 3) and g["actor0"].shape[1] == gp.nparams(D, H, A, La) and g["critic0"].shape[1] == gp.nparams(D, H, 1, Lc)
    with gp.recurrent_ac(La, Lc):
        lr = ap.Learner(torch.tensor(g["actor0"]), torch.tensor(g["critic0"]), D, H, A, gamma=float(g["gamma"]), n_steps=int(g["n_steps"]),
                        entropy_coef=float(g["entropy_coef"]), value_loss_coef=float(g["value_loss_coef"]), num_epochs=0, ppo_clip=float(g["ppo_clip"]))
        lr.target = torch.tensor(g["target0"])
        for i in range(3):
            m = lr.update(_ac_batch(g, i), int(g["steps"][i]))
            np.testing.assert_allclose([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]], g["metrics"][i], rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(lr.actor().detach().numpy(), g[f"actor{i + 1}"], rtol=0, atol=3e-6)
            np.testing.assert_allclose(lr.critic().detach().numpy(), g[f"critic{i + 1}"], rtol=0, atol=3e-6)


@pytest.mark.gpu
def test_hip_recurrent_families_of_two_depths_match_reference():
    """A2CNetwork with actor.layers [24, 24] and critic.layers [24] * 4, both use_rnn (C-ABI 219: marlhip_ac_config.critic_n_hidden carries the
    critics' depth into the recurrent step): state_dict keys, hidden-state shapes per family, metrics and all three blocks after 3 updates"""
    from collections import namedtuple

    from codebase_amd.ac.model import A2CNetwork
    from codebase_amd.dqn import model as M
    from codebase_amd.spaces import Box, Discrete, Tuple

    Batch = namedtuple("Batch", ["obss", "actions", "rewards", "dones", "filled", "action_masks"])
    g = dict(np.load(os.path.join(G, AC_TWO_DEPTHS)))
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    la, lc = [int(x) for x in g["layers"]], [int(x) for x in g["critic_layers"]]
    La, Lc = len(la) - 1, len(lc) - 1
    cfg = dict(optimizer="Adam", lr=3e-4, gamma=float(g["gamma"]), grad_clip=False, n_steps=int(g["n_steps"]), entropy_coef=float(g["entropy_coef"]),
               value_loss_coef=float(g["value_loss_coef"]), standardise_returns=False, target_update_interval_or_tau=200)
    base = dict(parameter_sharing=False, use_orthogonal_init=True, use_rnn=True)
    net = A2CNetwork(Tuple([Box(-1, 8, (D,))] * P), Tuple([Discrete(A)] * P), cfg, dict(base, layers=la), dict(base, layers=lc, centralised=False), "cuda")
    Hk = net.spec.hidden
    assert net.recurrent and net.rnn_layers["actor"] == La and net.rnn_layers["critic"] == Lc and not net.spec.wide
    assert list(net.state_dict().keys()) == [str(k) for k in g["state_dict_keys"]]
    assert net.init_actor_hiddens(7)[0].shape == (La, 7, Hk) and net.init_critic_hiddens(7)[1].shape == (Lc, 7, Hk)
    net.actor_params.copy_(M.pad_gru_blocks(torch.tensor(g["actor0"]), D, H, A, Hk, La))
    net.critic_params.copy_(M.pad_gru_blocks(torch.tensor(g["critic0"]), D, H, 1, Hk, Lc))
    net.target_critic_params.copy_(M.pad_gru_blocks(torch.tensor(g["target0"]), D, H, 1, Hk, Lc))
    obs = [torch.rand(7, D) for _ in range(P)]
    acts, ah = net.act(obs, net.init_actor_hiddens(7))
    v, ch = net.get_value(obs, net.init_critic_hiddens(7))
    assert acts.shape == (P, 7, 1) and v.shape == (7, P) and ah[0].shape == (La, 7, Hk) and ch[0].shape == (Lc, 7, Hk)
    for i in range(3):
        b = Batch(*(x.cuda() for x in _ac_batch(g, i).values()), None)
        m = net.update(b._replace(dones=b.dones.float()), int(g["steps"][i]))
        np.testing.assert_allclose([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]], g["metrics"][i], rtol=1e-4, atol=1e-5)
        for got, want, a, L in ((net.actor_params, g[f"actor{i + 1}"], A, La), (net.critic_params, g[f"critic{i + 1}"], 1, Lc),
                                (net.target_critic_params, g[f"target{i + 1}"], 1, Lc)):
            diff = np.abs(_live(M, got.cpu(), D, H, a, Hk, L).numpy() - want)
            assert diff.max() <= 5e-5 and (diff > 5e-6).mean() <= 1e-4, (i, diff.max(), (diff > 5e-6).sum())


# ---- a stack next to a feed-forward family (actor.use_rnn != critic.use_rnn with three layer sizes on the recurrent side) ---------------------
AC_MIXED_STACK = [("learner_a2c_rnn_actor_L2_ff_critic_h24.npz", "actor"), ("learner_ppo_ff_actor_rnn_critic_L2_h24.npz", "critic")]


@pytest.mark.parametrize("name,which", AC_MIXED_STACK)
def test_ac_oracle_port_with_one_stacked_recurrent_family_matches_reference(name, which):
    from oracle import ac_update_port as ap

    g = dict(np.load(os.path.join(G, name)))
    D, H, A = int(g["D"]), int(g["H"]), int(g["A"])
    rec = g["actor0"] if which == "actor" else g["critic0"]
    assert rec.shape[1] == gp.nparams(D, H, A if which == "actor" else 1, 2)
    with gp.mixed_ac():
        lr = ap.Learner(torch.tensor(g["actor0"]), torch.tensor(g["critic0"]), D, H, A, gamma=float(g["gamma"]), n_steps=int(g["n_steps"]),
                        entropy_coef=float(g["entropy_coef"]), value_loss_coef=float(g["value_loss_coef"]),
                        num_epochs=int(g["num_epochs"]) if "ppo" in name else 0, ppo_clip=float(g["ppo_clip"]))
        lr.target = torch.tensor(g["target0"])
        for i in range(3):
            m = lr.update(_ac_batch(g, i), int(g["steps"][i]))
            np.testing.assert_allclose([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]], g["metrics"][i], rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(lr.actor().detach().numpy(), g[f"actor{i + 1}"], rtol=0, atol=3e-6)
            np.testing.assert_allclose(lr.critic().detach().numpy(), g[f"critic{i + 1}"], rtol=0, atol=3e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name,which", AC_MIXED_STACK)
def test_hip_one_stacked_recurrent_family_next_to_a_feed_forward_one_matches_reference(name, which):
    """marlhip_mixed_* with the recurrent family as a stack (C-ABI 219): recurrent actors [24] * 3 + feed-forward critics [24, 24] (A2CNetwork),
    feed-forward actors [24, 24] + recurrent critics [24] * 3 (PPONetwork) - state_dict keys, metrics, the live parts of all blocks"""
    from collections import namedtuple

    from codebase_amd.ac.model import A2CNetwork, PPONetwork
    from codebase_amd.dqn import model as M
    from codebase_amd.spaces import Box, Discrete, Tuple

    Batch = namedtuple("Batch", ["obss", "actions", "rewards", "dones", "filled", "action_masks"])
    g = dict(np.load(os.path.join(G, name)))
    P, D, H, A, L = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"]), 2
    ppo = "ppo" in name
    cfg = dict(optimizer="Adam", lr=3e-4, gamma=float(g["gamma"]), grad_clip=False, n_steps=int(g["n_steps"]), entropy_coef=float(g["entropy_coef"]),
               value_loss_coef=float(g["value_loss_coef"]), standardise_returns=False, target_update_interval_or_tau=200,
               num_epochs=int(g["num_epochs"]), ppo_clip=float(g["ppo_clip"]))
    base = dict(parameter_sharing=False, use_orthogonal_init=True)
    la = [H] * (L + 1) if which == "actor" else [H, H]
    lc = [H] * (L + 1) if which == "critic" else [H, H]
    net = (PPONetwork if ppo else A2CNetwork)(Tuple([Box(-1, 8, (D,))] * P), Tuple([Discrete(A)] * P), cfg, dict(base, layers=la, use_rnn=which == "actor"),
                                             dict(base, layers=lc, use_rnn=which == "critic", centralised=False), "cuda")
    Hk = net.spec.hidden
    assert net.updater.mixed_rnn == which and net.rnn_layers[which] == L and net.rnn_layers["critic" if which == "actor" else "actor"] == 1
    assert list(net.state_dict().keys()) == [str(k) for k in g["state_dict_keys"]]

    def put(dst, src, d, a, recurrent):
        dst.copy_(M.pad_gru_blocks(torch.tensor(src), d, H, a, Hk, L) if recurrent else M.pad_blocks(torch.tensor(src), d, [H, H], a, Hk))

    def live(blocks, d, a, recurrent):
        if recurrent:
            return _live(M, blocks.cpu(), d, H, a, Hk, L).numpy()
        return torch.stack([torch.cat([v.reshape(-1) for _, v in M.block_views(blocks[p].cpu(), d, (H, H), a, Hk)]) for p in range(P)]).numpy()

    put(net.actor_params, g["actor0"], D, A, which == "actor")
    put(net.critic_params, g["critic0"], D, 1, which == "critic")
    put(net.target_critic_params, g["target0"], D, 1, which == "critic")
    obs = [torch.rand(5, D) for _ in range(P)]
    acts, hid = net.act(obs, net.init_actor_hiddens(5))
    v, ch = net.get_value(obs, net.init_critic_hiddens(5))
    assert acts.shape == (P, 5, 1) and v.shape == (5, P)
    assert (hid[0] is not None and hid[0].shape == (L, 5, Hk)) == (which == "actor") and (ch[0] is not None and ch[0].shape == (L, 5, Hk)) == (which == "critic")
    for i in range(3):
        b = Batch(*(x.cuda() for x in _ac_batch(g, i).values()), None)
        m = net.update(b._replace(dones=b.dones.float()), int(g["steps"][i]))
        np.testing.assert_allclose([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]], g["metrics"][i], rtol=1e-4, atol=1e-5)
        for got, want, a, rec in ((net.actor_params, g[f"actor{i + 1}"], A, which == "actor"), (net.critic_params, g[f"critic{i + 1}"], 1, which == "critic"),
                                  (net.target_critic_params, g[f"target{i + 1}"], 1, which == "critic")):
            diff = np.abs(live(got, D, a, rec) - want)
            assert diff.max() <= 5e-5 and (diff > 5e-6).mean() <= 1e-4, (i, diff.max(), (diff > 5e-6).sum())


# ---- at the bench batch: a stack against the float64 port ---------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("H,L,mode", [(128, 2, "idqn"), (64, 3, "vdn")])
def test_stack_at_the_bench_batch_vs_float64_port(H, L, mode):
    """B = 4096 sequences x 26 steps x 2 agents (the headline's batch, 212,992 rows through every layer), reference-default width 128 with two
    stacked layers and width 64 with three: loss within north_star's 1e-5 of the float64 port, every gradient entry within 3e-4 of the
    largest (tests/test_gpu_at_size_vs_oracle.py's bound for the feed-forward learners at this size)"""
    from codebase_amd import hip as h
    from oracle import dqn_port as dp
    from tests.test_gpu_at_size_vs_oracle import assert_grad_at_size

    P, D, A, T, B = 2, 15, 6, 25, 4096
    gen = torch.Generator().manual_seed(900 + L)
    params = 0.12 * torch.randn(P, gp.nparams(D, H, A, L), generator=gen)
    target = params + 0.05 * torch.randn(P, gp.nparams(D, H, A, L), generator=gen)
    batch = dp.synthetic_batch(P, T, B, D, A, seed=70 + L)
    batch["obss"] = batch["obss"] * 0.25
    if mode == "vdn":
        batch["rewards"][1:] = batch["rewards"][0]
    pr = params.double().requires_grad_(True)
    ref = gp.compute_loss(pr, target.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}, 0.99, True, D, H, A, mode=mode)
    ref.backward()
    spec = h.NetSpec(P, D, H, A, n_hidden=L + 1)
    hb = h.Batch(*(batch[k].cuda().contiguous() for k in ("obss", "actions", "rewards", "dones", "filled")), None)
    loss, grad = h.gru_loss_grad(spec, params.cuda(), target.cuda(), hb, mode=1 if mode == "vdn" else 0)
    rel = abs(float(loss.cpu()[0]) - ref.item()) / abs(ref.item())
    print(f"[at-size] stacked GRU {mode} H{H} L{L} B{B}: loss {float(loss.cpu()[0]):.7f} vs float64 {ref.item():.7f} (relative {rel:.2e})")
    assert rel <= 1e-5
    assert_grad_at_size(grad.cpu().numpy(), pr.grad.numpy(), f"stacked GRU {mode} H{H} L{L} gradient")
