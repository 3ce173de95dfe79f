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
@pytest.mark.parametrize("P,T,B,D,A,mode,dq,H", [(2, 25, 70, 15, 6, "idqn", True, 64), (4, 6, 16, 27, 6, "vdn", False, 64),
                                                 (8, 5, 33, 39, 6, "idqn", True, 64), (4, 12, 20, 71, 5, "idqn", False, 64),
                                                 (2, 1, 1, 12, 6, "vdn", True, 64), (2, 25, 70, 15, 6, "idqn", True, 128),
                                                 (3, 7, 130, 18, 6, "vdn", False, 128), (4, 5, 20, 71, 5, "idqn", True, 128),
                                                 (8, 3, 17, 39, 6, "idqn", False, 128)])
def test_hip_loss_and_gradient_other_shapes_vs_port(P, T, B, D, A, mode, dq, H):
    """hidden 64 (the whole network LDS-resident) and hidden 128 (the reference default: gate matrices streamed through LDS)"""
    from codebase_amd import hip as h
    from oracle import dqn_port as dp

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


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode", FILES)
def test_recurrent_qnetwork_interface_matches_reference(name, mode):
    """QNetwork / VDNetwork(use_rnn=True): the reference's state_dict keys, its act() trace with carried hidden states, and two
    update() calls (loss.backward + clip_grad_norm_ + Adam) against the reference's own"""
    from codebase_amd.dqn.model import QNetwork, VDNetwork
    from codebase_amd.hip import Batch
    from codebase_amd.spaces import Box, Discrete, Tuple

    g, batch = load(name)
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False,
                 target_update_interval_or_tau=200)
    net = (VDNetwork if mode == "vdn" else QNetwork)(Tuple([Box(-1, 8, (D,))] * P), Tuple([Discrete(A)] * P), hyper, [H, H], False, True, True,
                                                     "cuda")
    assert list(net.state_dict().keys()) == list(g["keys"])
    # default init statistics: nn.Linear / nn.GRU uniform(-1/sqrt(fan), 1/sqrt(fan)), orthogonal output layer with zero bias
    sd = net.state_dict()
    assert float(sd["critic.independent.0.rnn.weight_hh_l0"].abs().max()) <= 1 / 8 + 1e-6
    assert float(sd["critic.independent.0.final_layer.bias"].abs().max()) == 0.0
    assert torch.equal(sd["critic.independent.1.rnn.weight_ih_l0"], sd["target.independent.1.rnn.weight_ih_l0"])
    net.params.copy_(torch.tensor(g["params0"]))
    net.target_params.copy_(torch.tensor(g["target0"]))
    hid = net.init_hiddens(1)
    assert hid[0].shape == (1, 1, H)
    for t in range(6):
        acts, hid = net.act([o for o in g["act_obs"][t]], hid, 0.0)
        np.testing.assert_allclose(torch.stack([h.reshape(-1) for h in hid]).cpu().numpy(), g["act_hiddens"][t], rtol=0, atol=5e-6)
        assert len(acts) == P and all(isinstance(a, int) and 0 <= a < A for a in acts)
    acts, hid2 = net.act([o for o in g["act_obs"][0]], hid, 1.0)  # a random step still advances the hidden state
    assert not torch.equal(hid2[0], hid[0])
    b = Batch(*(batch[k] for k in ("obss", "actions", "rewards", "dones", "filled")), None)
    losses = [net.update(b)["loss"] for _ in range(2)]
    np.testing.assert_allclose(losses, g["losses"], rtol=5e-5)
    np.testing.assert_allclose(net.params.cpu().numpy(), g["params2"], rtol=0, atol=5e-6)
    with pytest.raises(NotImplementedError):
        QNetwork(Tuple([Box(-1, 8, (D,))] * P), Tuple([Discrete(A)] * P), hyper, [32] * 6, False, True, True, "cuda")  # five stacked GRU layers (up to four: tests/test_gru_stacked.py)
    wide = QNetwork(Tuple([Box(-1, 8, (D,))] * P), Tuple([Discrete(A)] * P), hyper, [128, 128], False, True, True, "cuda")  # reference default
    assert wide.state_dict()["critic.independent.0.rnn.weight_ih_l0"].shape == (384, 128)
    acts, hid = wide.act([o for o in g["act_obs"][0]], wide.init_hiddens(1), 0.0)
    assert hid[0].shape == (1, 1, 128) and len(acts) == P


@pytest.mark.gpu
def test_recurrent_idqn_end_to_end(tmp_path, monkeypatch):
    """+algorithm=idqn algorithm.model.use_rnn=True through the drop-in command line: vectorised (modular collection loop with the
    hidden state carried on the device) and scalar (the reference-shaped python loop), then eval of the saved checkpoint"""
    from codebase_amd import eval as ev
    from codebase_amd import run

    NAME = "lbforaging:Foraging-8x8-2p-3f-v3"
    monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / "vec"))
    df = run.main(["+algorithm=idqn", f"env.name={NAME}", "env.time_limit=25", "env.parallel_envs=128", "algorithm.model.layers=[64,64]",
                   "algorithm.model.use_rnn=True", "seed=1", "algorithm.total_steps=120000", "algorithm.eval_interval=40000",
                   "algorithm.save_interval=100000"])
    assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all() and np.isfinite(df["mean_episode_returns"]).all()
    ev.main([f"path={tmp_path / 'vec'}", "episodes=64"])
    monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / "scalar"))
    df = run.main(["+algorithm=vdn", f"env.name={NAME}", "env.time_limit=25", "algorithm.model.layers=[64,64]", "algorithm.model.use_rnn=True",
                   "seed=1", "algorithm.total_steps=400", "algorithm.training_start=100", "algorithm.batch_size=4",
                   "algorithm.eval_interval=200", "algorithm.eval_episodes=3"])
    assert df.shape[0] >= 1 and np.isfinite(df["loss"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("P,S,B,D,A", [(2, 26, 70, 15, 6), (4, 3, 16, 71, 5)])
def test_hip_forward_hidden128_vs_port(P, S, B, D, A):
    from codebase_amd import hip as h

    H = 128
    gen = torch.Generator().manual_seed(7)
    params = 0.1 * torch.randn(P, gp.nparams(D, H, A), generator=gen)
    obs = torch.randint(-1, 8, (P, S, B, D), generator=gen).float() * 0.25
    spec = h.NetSpec(P, D, H, A)
    q = h.gru_forward(spec, params.cuda(), obs.cuda())
    np.testing.assert_allclose(q.cpu().numpy(), gp.q_values(params, obs, D, H, A).numpy(), rtol=0, atol=2e-5)
    h0 = 0.3 * torch.randn(P, B, H, generator=gen)
    q1, h1 = h.gru_forward(spec, params.cuda(), obs[:, :1].contiguous().cuda(), h_in=h0.cuda(), want_h=True)
    for p in range(P):
        qr, hr = gp.cell(gp.split(params[p], D, H, A), obs[p, 0], h0[p])
        np.testing.assert_allclose(q1[p, 0].cpu().numpy(), qr.numpy(), rtol=0, atol=2e-5)
        np.testing.assert_allclose(h1[p].cpu().numpy(), hr.numpy(), rtol=0, atol=2e-5)


def test_oracle_qmix_port_with_recurrent_agents_matches_reference():
    g, batch = load("learner_gru_qmix_H64.npz")
    D, H, A = int(g["D"]), int(g["H"]), int(g["A"])
    pr, mr = torch.tensor(g["params0"]).requires_grad_(True), torch.tensor(g["mixer0"]).requires_grad_(True)
    loss = gp.compute_qmix_loss(pr, torch.tensor(g["target0"]), mr, torch.tensor(g["tmixer0"]), batch, 0.99, True, D, H, A)
    loss.backward()
    assert abs(loss.item() - g["loss0"]) <= 1e-5 * abs(g["loss0"])
    np.testing.assert_allclose(pr.grad.numpy(), g["grad0"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(mr.grad.numpy(), g["mgrad0"], rtol=1e-4, atol=2e-6)


@pytest.mark.gpu
def test_recurrent_qmix_matches_reference():
    """QMixNetwork(use_rnn=True): loss, agent and mixer gradients, two update() calls, vs the reference's own"""
    from codebase_amd.dqn.model import QMixNetwork
    from codebase_amd.hip import Batch
    from codebase_amd.spaces import Box, Discrete, Tuple

    g, batch = load("learner_gru_qmix_H64.npz")
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False,
                 target_update_interval_or_tau=200)
    net = QMixNetwork(Tuple([Box(-1, 8, (D,))] * P), Tuple([Discrete(A)] * P), hyper, [H, H], False, True, True,
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


# ---- actor-critic learners with recurrent actors and critics (ac/model.py:189-352 with use_rnn)
AC_GRU = ["learner_a2c_gru_H64.npz", "learner_ppo_gru_H128.npz", "learner_maa2c_gru_H64.npz", "learner_mappo_gru_p3_H64.npz"]


def _ac_batch(g, i):
    return {k: torch.tensor(g[f"batch{i}_{k}"]) for k in ("obss", "actions", "rewards", "dones", "filled")}


@pytest.mark.parametrize("name", AC_GRU)
def test_ac_oracle_port_with_recurrent_networks_matches_reference(name):
    from oracle import ac_update_port as ap

    g = dict(np.load(os.path.join(G, name)))
    D, H, A = int(g["D"]), int(g["H"]), int(g["A"])
    with gp.recurrent_ac():
        lr = ap.Learner(torch.tensor(g["actor0"]), torch.tensor(g["critic0"]), D, H, A, gamma=float(g["gamma"]), n_steps=int(g["n_steps"]),
                        entropy_coef=float(g["entropy_coef"]), value_loss_coef=float(g["value_loss_coef"]),
                        num_epochs=int(g["num_epochs"]) if "ppo" in name else 0, ppo_clip=float(g["ppo_clip"]))
        lr.target = torch.tensor(g["target0"])
        for i in range(3):
            m = lr.update(_ac_batch(g, i), int(g["steps"][i]))
            np.testing.assert_allclose([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]], g["metrics"][i], rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(lr.actor().detach().numpy(), g[f"actor{i + 1}"], rtol=0, atol=3e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", AC_GRU)
def test_hip_recurrent_actor_critic_matches_reference(name):
    from collections import namedtuple

    from codebase_amd import hip as h

    Batch = namedtuple("Batch", ["obss", "actions", "rewards", "dones", "filled", "action_masks"])
    g = dict(np.load(os.path.join(G, name)))
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    spec = h.NetSpec(P, D, H, A)
    block = torch.cat([torch.tensor(g["actor0"]).reshape(-1), torch.tensor(g["critic0"]).reshape(-1)]).cuda()
    up = h.AcUpdater(spec, block, torch.tensor(g["target0"]).cuda().contiguous(), lr=3e-4, gamma=float(g["gamma"]), n_steps=int(g["n_steps"]),
                     entropy_coef=float(g["entropy_coef"]), value_loss_coef=float(g["value_loss_coef"]), grad_clip=False,
                     ppo_clip=float(g["ppo_clip"]), recurrent=True, centralised_critic="maa2c" in name or "mappo" in name)
    assert up.n_actor == g["actor0"].shape[1] and up.n_critic == g["critic0"].shape[1]
    ppo = "ppo" in name
    for i in range(3):
        b = Batch(*(v.cuda() for v in _ac_batch(g, i).values()), None)
        if ppo:
            up.ppo_prepare(b)
            acc = np.zeros(4)
            for _ in range(int(g["num_epochs"])):
                acc += up.ppo_loss_grad(b).cpu().numpy()[:4]
                up.apply()
            m = acc / int(g["num_epochs"])
        else:
            m = up.a2c_loss_grad(b).cpu().numpy()[:4]
            up.apply()
        np.testing.assert_allclose(m, g["metrics"][i], rtol=1e-4, atol=1e-5)
        if int(g["steps"][i]) % 200 == 0:
            up.target_critic.copy_(up.critic)
        # Adam divides by sqrt(v): an element whose gradient is fp32 noise can move by a fraction of lr (3e-4) in either direction;
        # all but a handful of the ~3e5 parameters agree to 5e-6
        for got, want in ((up.block[:P * up.n_actor], g[f"actor{i + 1}"]), (up.block[P * up.n_actor:], g[f"critic{i + 1}"])):
            diff = np.abs(got.cpu().numpy().reshape(P, -1) - want)
            assert diff.max() <= 5e-5 and (diff > 5e-6).mean() <= 1e-4, (diff.max(), (diff > 5e-6).sum())


# ---- actor.use_rnn != critic.use_rnn (ac/model.py:45-97 builds each family from its own flag; csrc/mixed_ac.hip) ----------------------------
AC_MIXED = [("learner_a2c_rnn_actor_ff_critic_H64.npz", "actor"), ("learner_ppo_ff_actor_rnn_critic_H64.npz", "critic")]


@pytest.mark.parametrize("name,which", AC_MIXED)
def test_ac_oracle_port_with_one_recurrent_family_matches_reference(name, which):
    from oracle import ac_update_port as ap

    g = dict(np.load(os.path.join(G, name)))
    assert (int(g["actor_rnn"]), int(g["critic_rnn"])) == ((1, 0) if which == "actor" else (0, 1))
    D, H, A = int(g["D"]), int(g["H"]), int(g["A"])
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
@pytest.mark.parametrize("name,which", AC_MIXED)
def test_hip_actor_critic_with_one_recurrent_family_matches_reference(name, which):
    """marlhip_mixed_a2c_loss_grad / marlhip_mixed_ppo_* through A2CNetwork / PPONetwork built with actor.use_rnn != critic.use_rnn: the
    reference's state_dict keys (recurrent names on one family, network.N on the other), metrics and both blocks after 3 updates"""
    from collections import namedtuple

    from codebase_amd.ac.model import A2CNetwork, PPONetwork
    from codebase_amd.spaces import Box, Discrete, Tuple

    Batch = namedtuple("Batch", ["obss", "actions", "rewards", "dones", "filled", "action_masks"])
    g = dict(np.load(os.path.join(G, name)))
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    ppo = "ppo" in name
    cfg = dict(optimizer="Adam", lr=3e-4, gamma=float(g["gamma"]), grad_clip=False, n_steps=int(g["n_steps"]), entropy_coef=float(g["entropy_coef"]),
               value_loss_coef=float(g["value_loss_coef"]), standardise_returns=False, target_update_interval_or_tau=200,
               num_epochs=int(g["num_epochs"]), ppo_clip=float(g["ppo_clip"]))
    net_cfg = dict(layers=[H, H], parameter_sharing=False, use_orthogonal_init=True)
    net = (PPONetwork if ppo else A2CNetwork)(Tuple([Box(-1, 8, (D,))] * P), Tuple([Discrete(A)] * P), cfg, dict(net_cfg, use_rnn=which == "actor"),
                                             dict(net_cfg, use_rnn=which == "critic", centralised=False), "cuda")
    assert net.updater.mixed_rnn == which and not net.keeps_actor_forward
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["state_dict_keys"]]
    rec, ff = ("actor", "critic") if which == "actor" else ("critic", "actor")
    assert f"{rec}.independent.0.rnn.weight_hh_l0" in sd and f"{ff}.independent.1.network.2.weight" in sd
    assert net.actor_params.shape == g["actor0"].shape and net.critic_params.shape == g["critic0"].shape
    net.actor_params.copy_(torch.tensor(g["actor0"]))
    net.critic_params.copy_(torch.tensor(g["critic0"]))
    net.target_critic_params.copy_(torch.tensor(g["target0"]))
    # reference-shaped acting / values with the hidden state of the recurrent family only
    obs = [torch.rand(5, D) for _ in range(P)]
    acts, hid = net.act(obs, net.init_actor_hiddens(5))
    v, ch = net.get_value(obs, net.init_critic_hiddens(5))
    assert acts.shape == (P, 5, 1) and v.shape == (5, P)
    assert (hid[0] is not None and hid[0].shape == (1, 5, H)) == (which == "actor") and (ch[0] is not None) == (which == "critic")
    for i in range(3):
        b = Batch(*(x.cuda() for x in _ac_batch(g, i).values()), None)
        m = net.update(b._replace(dones=b.dones.float()), int(g["steps"][i]))
        np.testing.assert_allclose([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]], g["metrics"][i], rtol=1e-4, atol=1e-5)
        for got, want in ((net.actor_params, g[f"actor{i + 1}"]), (net.critic_params, g[f"critic{i + 1}"]), (net.target_critic_params, g[f"target{i + 1}"])):
            diff = np.abs(got.cpu().numpy() - want)
            assert diff.max() <= 5e-5 and (diff > 5e-6).mean() <= 1e-4, (i, diff.max(), (diff > 5e-6).sum())


@pytest.mark.gpu
def test_ia2c_with_recurrent_actors_and_feed_forward_critics_end_to_end(tmp_path, monkeypatch):
    """+algorithm=ia2c algorithm.model.actor.use_rnn=True algorithm.model.critic.use_rnn=False (and the reverse, whose rollout is the fused
    collector's) through run.main"""
    from codebase_amd import run

    NAME = "lbforaging:Foraging-8x8-2p-3f-v3"
    for a_rnn, c_rnn in ((True, False), (False, True)):
        monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / f"a{int(a_rnn)}c{int(c_rnn)}"))
        df = run.main(["+algorithm=ia2c", f"env.name={NAME}", "env.time_limit=25", "env.parallel_envs=128", "algorithm.model.actor.layers=[64,64]",
                       "algorithm.model.critic.layers=[64,64]", f"algorithm.model.actor.use_rnn={a_rnn}", f"algorithm.model.critic.use_rnn={c_rnn}",
                       "seed=1", "algorithm.total_steps=40000", "algorithm.eval_interval=15000"])
        assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all() and np.isfinite(df["mean_episode_returns"]).all()
    # the warehouse's 71-wide rows, 128-128, PPO: recurrent critics next to feed-forward actors
    monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / "rware"))
    df = run.main(["+algorithm=ippo", "env.name=rware:rware-tiny-2ag-v2", "env.time_limit=50", "env.parallel_envs=64", "algorithm.model.actor.layers=[128,128]",
                   "algorithm.model.critic.layers=[128,128]", "algorithm.model.actor.use_rnn=False", "algorithm.model.critic.use_rnn=True", "seed=2",
                   "algorithm.total_steps=20000", "algorithm.eval_interval=8000"])
    assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all()


@pytest.mark.gpu
def test_recurrent_ia2c_and_ippo_end_to_end(tmp_path, monkeypatch):
    """+algorithm=ia2c / ippo with use_rnn for actor and critic: recurrent rollout collection (hidden state carried on the device,
    Philox sampling), recurrent update, reference-shaped act / get_value with hidden states"""
    from codebase_amd import run
    from codebase_amd.ac.model import A2CNetwork
    from codebase_amd.spaces import Box, Discrete, Tuple

    net_cfg = dict(layers=[64, 64], parameter_sharing=False, use_orthogonal_init=True, use_rnn=True)
    cfg = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=False, n_steps=5, entropy_coef=0.001, value_loss_coef=0.5,
               standardise_returns=False, target_update_interval_or_tau=200)
    net = A2CNetwork(Tuple([Box(-1, 8, (15,))] * 2), Tuple([Discrete(6)] * 2), cfg, net_cfg, dict(net_cfg, centralised=False), "cuda")
    sd = net.state_dict()
    assert "actor.independent.0.rnn.weight_hh_l0" in sd and sd["critic.independent.1.final_layer.weight"].shape == (1, 64)
    obs = [torch.rand(5, 15) for _ in range(2)]
    hid = net.init_actor_hiddens(5)
    acts, hid = net.act(obs, hid)
    assert acts.shape == (2, 5, 1) and hid[0].shape == (1, 5, 64)
    v, ch = net.get_value(obs, net.init_critic_hiddens(5))
    assert v.shape == (5, 2) and ch[1].shape == (1, 5, 64)
    ref = torch.cat([gp.cell(gp.split(net.critic_params[p].cpu(), 15, 64, 1), obs[p], torch.zeros(5, 64))[0] for p in range(2)], dim=-1)
    np.testing.assert_allclose(v.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    NAME = "lbforaging:Foraging-8x8-2p-3f-v3"
    for algo in ("ia2c", "ippo"):
        monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / algo))
        df = run.main([f"+algorithm={algo}", f"env.name={NAME}", "env.time_limit=25", "env.parallel_envs=128",
                       "algorithm.model.actor.layers=[64,64]", "algorithm.model.critic.layers=[64,64]", "algorithm.model.actor.use_rnn=True",
                       "algorithm.model.critic.use_rnn=True", "seed=1", "algorithm.total_steps=40000", "algorithm.eval_interval=15000"])
        assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all() and np.isfinite(df["mean_episode_returns"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode", [("learner_gru_shared_H64.npz", "idqn"), ("learner_gru_seps_vdn_H128.npz", "vdn")])
def test_recurrent_networks_with_parameter_sharing_match_reference(name, mode):
    """parameter_sharing=True / SePS [0, 0, 1] with use_rnn=True (MultiAgentSharedNetwork over RNNNetworks): every agent carries its
    own hidden state through the network of its group; the gradient of a shared network is the sum over its agents"""
    from codebase_amd.dqn.model import QNetwork, VDNetwork
    from codebase_amd.hip import Batch
    from codebase_amd.spaces import Box, Discrete, Tuple

    g, batch = load(name)
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    sharing = [int(x) for x in g["sharing"]]
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False,
                 target_update_interval_or_tau=200)
    net = (VDNetwork if mode == "vdn" else QNetwork)(Tuple([Box(-1, 8, (D,))] * P), Tuple([Discrete(A)] * P), hyper, [H, H],
                                                     True if sharing == [0] * P else sharing, True, True, "cuda")
    assert list(net.state_dict().keys()) == list(g["keys"]) and net.params.shape == g["params0"].shape
    net.params.copy_(torch.tensor(g["params0"]))
    net.target_params.copy_(torch.tensor(g["target0"]))
    hb = Batch(*(batch[k].cuda().contiguous() for k in ("obss", "actions", "rewards", "dones", "filled")), None)
    loss, grad = net.updater.loss_grad(hb, mode=net.mode)
    assert abs(loss.cpu().numpy()[0] - g["loss0"]) <= 3e-5 * abs(g["loss0"])
    np.testing.assert_allclose(grad.cpu().numpy(), g["grad0"], rtol=3e-4, atol=3e-5 * max(1e-2, np.abs(g["grad0"]).max()))
    b = Batch(*(batch[k] for k in ("obss", "actions", "rewards", "dones", "filled")), None)
    np.testing.assert_allclose([net.update(b)["loss"] for _ in range(2)], g["losses"], rtol=5e-5)
    np.testing.assert_allclose(net.params.cpu().numpy(), g["params2"], rtol=0, atol=5e-6)


@pytest.mark.gpu
def test_recurrent_idqn_with_standardise_returns_matches_reference():
    from codebase_amd.dqn.model import QNetwork
    from codebase_amd.hip import Batch
    from codebase_amd.spaces import Box, Discrete, Tuple

    g = dict(np.load(os.path.join(G, "learner_gru_std_H64.npz")))
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=True,
                 target_update_interval_or_tau=200)
    net = QNetwork(Tuple([Box(-1, 8, (D,))] * P), Tuple([Discrete(A)] * P), hyper, [H, H], False, True, True, "cuda")
    net.params.copy_(torch.tensor(g["params0"]))
    net.target_params.copy_(torch.tensor(g["target0"]))
    for i in range(3):
        b = Batch(*(torch.tensor(g[f"batch{i}_{k}"]) for k in ("obss", "actions", "rewards", "dones", "filled")), None)
        loss = net.update(b)["loss"]
        assert abs(loss - g["losses"][i]) <= 5e-5 * abs(g["losses"][i])
        np.testing.assert_allclose(net.ret_ms.mean.cpu().numpy(), g[f"ret_mean{i + 1}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(net.ret_ms.var.cpu().numpy(), g[f"ret_var{i + 1}"], rtol=1e-5, atol=1e-6)
        assert abs(net.ret_ms.count - float(g[f"ret_count{i + 1}"])) < 1e-6
        np.testing.assert_allclose(net.params.cpu().numpy(), g[f"params{i + 1}"], rtol=0, atol=5e-6)


@pytest.mark.gpu
def test_recurrent_networks_on_the_warehouse_end_to_end(tmp_path, monkeypatch):
    """use_rnn with the 71-wide / 5-action warehouse shapes: IDQN (modular collection loop) and IA2C (recurrent rollout loop)"""
    from codebase_amd import run

    NAME = "rware:rware-tiny-2ag-v2"
    for algo, extra in (("idqn", ["algorithm.model.layers=[64,64]", "algorithm.model.use_rnn=True"]),
                        ("ia2c", ["algorithm.model.actor.use_rnn=True", "algorithm.model.critic.use_rnn=True"])):
        monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / algo))
        df = run.main([f"+algorithm={algo}", f"env.name={NAME}", "env.time_limit=40", "env.parallel_envs=64", "seed=1",
                       "algorithm.total_steps=30000", "algorithm.eval_interval=10000"] + extra)
        assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all() and np.isfinite(df["mean_episode_returns"]).all()


@pytest.mark.gpu
def test_recurrent_networks_with_observe_id_and_sharing_end_to_end(tmp_path, monkeypatch):
    """the usual companions: env.observe_id + parameter_sharing + use_rnn (17-wide observations), IDQN and IA2C"""
    from codebase_amd import run

    NAME = "lbforaging:Foraging-8x8-2p-3f-v3"
    for algo, extra in (("idqn", ["algorithm.model.layers=[64,64]", "algorithm.model.use_rnn=True", "algorithm.model.parameter_sharing=True"]),
                        ("ia2c", ["algorithm.model.actor.use_rnn=True", "algorithm.model.critic.use_rnn=True",
                                  "algorithm.model.actor.parameter_sharing=True", "algorithm.model.critic.parameter_sharing=True"])):
        monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / algo))
        df = run.main([f"+algorithm={algo}", f"env.name={NAME}", "env.time_limit=25", "env.parallel_envs=64", "env.observe_id=True", "seed=1",
                       "algorithm.total_steps=30000", "algorithm.eval_interval=10000"] + extra)
        assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all() and np.isfinite(df["mean_episode_returns"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("h,mode,P,D", [(40, "idqn", 2, 15), (96, "vdn", 3, 18), (100, "idqn", 2, 15), (17, "idqn", 4, 27)])
def test_recurrent_networks_of_other_widths_match_the_port_at_the_true_width(h, mode, P, D):
    """`use_rnn` with layers [h, h], h not 64 / 128 (RNNNetwork builds any width, utils/models.py:51-116): zero-padded onto the
    recurrent kernels (dqn/model.py: recurrent_width, pad_gru_blocks).  The driver class against oracle/gru_port at the TRUE width:
    loss, the live parts of the parameters after 3 updates, the padding still exactly zero, state_dict in the reference's shapes."""
    from codebase_amd import hip as hh
    from codebase_amd.dqn import model as M
    from oracle import dqn_port as dp
    from tests.test_gpu_layers import spaces

    A, T, B = 6, 9, 24
    cfg = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False, target_update_interval_or_tau=2)
    torch.manual_seed(11)
    cls = M.VDNetwork if mode == "vdn" else M.QNetwork
    obs_space, act_space = spaces(P, D, A)
    net = cls(obs_space, act_space, cfg, [h, h], False, True, True, "cuda")
    Hk = net.spec.hidden
    assert Hk == (64 if h <= 64 else 128) and net.recurrent and not net.spec.wide
    sd = net.state_dict()
    assert sd["critic.independent.0.rnn.weight_ih_l0"].shape == (3 * h, h) and sd["critic.independent.1.first_layer.weight"].shape == (h, D)
    live0 = torch.stack([torch.cat([sd[f"critic.independent.{p}.{n}"].reshape(-1).cpu() for n in gp.NAMES]) for p in range(P)])
    assert live0.shape == (P, gp.nparams(D, h, A))
    pad_mask = torch.ones_like(net.params, dtype=torch.bool)
    for p in range(P):
        for _, view, _ in M.gru_block_views(pad_mask[p], D, h, A, Hk):
            view.fill_(False)
    assert int(pad_mask.sum()) > 0 and float(net.params[pad_mask].abs().max()) == 0.0
    # the port at the true width, driven by dqn_port.Learner through the recurrent network hooks
    ref_params, ref_target = live0.clone(), live0.clone()
    opt_tensors = [torch.nn.Parameter(ref_params[p].clone()) for p in range(P)]
    opt = torch.optim.Adam(opt_tensors, lr=3e-4)
    updates = last = 0
    for i in range(3):
        batch = dp.synthetic_batch(P, T, B, D, A, seed=40 + i)
        batch["obss"] = batch["obss"] * 0.25
        if mode == "vdn":
            batch["rewards"][1:] = batch["rewards"][0]
        loss_ref = gp.compute_loss(torch.stack(list(opt_tensors)), ref_target, batch, 0.99, True, D, h, A, mode=mode)
        opt.zero_grad()
        loss_ref.backward()
        torch.nn.utils.clip_grad_norm_(opt_tensors, 1.0)
        opt.step()
        updates += 1
        if updates - last >= 2:
            ref_target = torch.stack([t.detach().clone() for t in opt_tensors])
            last = updates
        hb = hh.Batch(*(batch[k] for k in ("obss", "actions", "rewards", "dones", "filled")), None)
        got = net.update(hb)["loss"]
        assert abs(got - loss_ref.item()) <= 5e-5 * max(abs(loss_ref.item()), 1e-3), (i, got, loss_ref.item())
    sd = net.state_dict()
    for prefix, ref in (("critic", torch.stack([t.detach() for t in opt_tensors])), ("target", ref_target)):
        live = torch.stack([torch.cat([sd[f"{prefix}.independent.{p}.{n}"].reshape(-1).cpu() for n in gp.NAMES]) for p in range(P)])
        np.testing.assert_allclose(live.numpy(), ref.numpy(), rtol=0, atol=5e-6, err_msg=prefix)
    assert float(net.params[pad_mask].abs().max()) == 0.0 and float(net.updater.exp_avg[pad_mask].abs().max()) == 0.0  # the padding never moves
    with pytest.raises(NotImplementedError):
        M.QNetwork(obs_space, act_space, cfg, [64] * 6, False, True, True, "cuda")  # five stacked GRU layers (tests/test_gru_stacked.py: up to four)
    with pytest.raises(NotImplementedError):
        M.QNetwork(obs_space, act_space, cfg, [256, 256], False, True, True, "cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("P,T,N,D,H", [(3, 9, 20, 24, 64), (3, 6, 33, 24, 128), (4, 7, 16, 27, 64), (2, 12, 40, 15, 128)])
def test_recurrent_centralised_critics_other_shapes_vs_port(P, T, N, D, H):
    """maa2c with use_rnn on the compiled (agents, observation) pairs the goldens do not cover (3 agents x 24: Foraging-10x10-3p-5f):
    recurrent actors, recurrent critics over the concatenated row - loss pieces and both gradients against the port"""
    from codebase_amd import hip as h
    from oracle import ac_update_port as ap

    A = 6
    g = torch.Generator().manual_seed(3)
    actor = 0.1 * torch.randn(P, gp.nparams(D, H, A), generator=g)
    critic = 0.1 * torch.randn(P, gp.nparams(P * D, H, 1), generator=g)
    target = 0.1 * torch.randn(P, gp.nparams(P * D, H, 1), generator=g)
    batch = ap.synthetic_batch(P, T, N, D, A, seed=7)
    with gp.recurrent_ac():
        a, c = actor.clone().requires_grad_(True), critic.clone().requires_grad_(True)
        loss, m = ap.a2c_loss(a, c, target, batch, D, H, A, n_steps=5, gamma=0.97, entropy_coef=0.01, value_loss_coef=0.5)
        loss.backward()
    spec = h.NetSpec(P, D, H, A)
    up = h.AcUpdater(spec, torch.cat([actor.reshape(-1), critic.reshape(-1)]).cuda(), target.cuda().contiguous(), gamma=0.97, n_steps=5,
                     entropy_coef=0.01, value_loss_coef=0.5, recurrent=True, centralised_critic=True)
    assert up.n_critic == gp.nparams(P * D, H, 1)
    from collections import namedtuple
    Batch = namedtuple("Batch", ["obss", "actions", "rewards", "dones", "filled", "action_masks"])
    b = Batch(*(batch[k].cuda() for k in ("obss", "actions", "rewards", "dones", "filled")), None)
    got = up.a2c_loss_grad(b).cpu().numpy()
    ref = [m["loss"].item(), m["actor_loss"].item(), m["value_loss"].item(), m["entropy"].item()]
    np.testing.assert_allclose(got[:4], ref, rtol=1e-4, atol=1e-5)
    for gg, rr in ((up.actor_grad, a.grad), (up.critic_grad, c.grad)):
        np.testing.assert_allclose(gg.cpu().numpy(), rr.numpy(), rtol=3e-4, atol=3e-4 * max(1e-3, float(rr.abs().max())))
