"""CPU: the oracle's learner restatement (oracle/dqn_port.py) against golden vectors produced by
the REFERENCE's own classes (oracle/make_golden.py -> tests/golden/).  This is what pins the
learner half of the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import dqn_port as dp

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name)))


def batch_of(g, i):
    return {k: torch.tensor(g[f"batch{i}_{k}"]) for k in ("obss", "actions", "rewards", "dones", "filled")}


@pytest.mark.parametrize("H", [64, 128])
def test_loss_grad_and_updates_match_reference(H):
    g = load(f"learner_H{H}.npz")
    P, D, A = int(g["P"]), int(g["D"]), int(g["A"])
    params = torch.tensor(g["params0"]).requires_grad_(True)
    target = torch.tensor(g["target0"])
    loss = dp.compute_loss(params, target, batch_of(g, 0), 0.99, True, D, H, A)
    loss.backward()
    assert abs(loss.item() - float(g["loss0"])) <= 1e-5 * abs(float(g["loss0"]))
    np.testing.assert_allclose(params.grad.numpy(), g["grad0"], rtol=1e-4, atol=1e-5)
    lr = dp.Learner(torch.tensor(g["params0"]), D, H, A, target_update_interval_or_tau=2)
    lr.target = torch.tensor(g["target0"])
    for i in range(3):
        m = lr.update(batch_of(g, i))
        assert abs(m["loss"] - float(g["losses"][i])) <= 2e-5 * abs(float(g["losses"][i]))
        np.testing.assert_allclose(lr.flat().detach().numpy(), g[f"params{i + 1}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(lr.target.numpy(), g[f"target{i + 1}"], rtol=0, atol=2e-6)
    # hard update happened exactly at update 2 (interval 2): target2 == params2, target3 == params2
    np.testing.assert_array_equal(g["target2"], g["params2"])
    np.testing.assert_array_equal(g["target3"], g["params2"])
    assert float(g["gnorm0"]) > 1.0  # clipping is active in this fixture


@pytest.mark.parametrize("H", [64, 128])
def test_act_greedy_matches_reference(H):
    g = load(f"learner_H{H}.npz")
    D, A = int(g["D"]), int(g["A"])
    obs = torch.tensor(g["act_obs"])
    N = obs.shape[1]
    acts, q = dp.act(torch.tensor(g["params0"]), obs, 0.0, torch.ones(N), torch.zeros(2, N, dtype=torch.int64), D, H, A)
    np.testing.assert_allclose(q.numpy(), g["act_q"], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(acts.numpy(), g["act_greedy"])


def test_replay_trace_matches_reference():
    g = load("replay.npz")
    P, D, T, CAP = int(g["P"]), int(g["D"]), int(g["T"]), int(g["CAP"])
    rb = dp.ReplayBuffer(CAP, P, D, T)
    for kind, o, a, r, d in zip(g["kind"], g["obs"], g["acts"], g["rews"], g["done"]):
        if kind == 0:
            rb.init_episode(list(o))
        else:
            rb.add(list(o), a, r, bool(d))
    assert rb.pos == int(g["pos"]) and len(rb) == int(g["length"])
    b = rb.sample_idx(g["idx"])
    for k in ("obss", "actions", "rewards", "dones", "filled"):
        np.testing.assert_array_equal(b[k].numpy(), g[k])
    # the stale tail of the re-used slot 0 (5-step episode overwritten by a 2-step one) survives
    assert g["filled"][:, 0].sum() == 5


def test_epsilon_schedule_matches_reference():
    g = load("eps.npz")
    lin = dp.epsilon_schedule("linear", 0.5, 1.0, 0.05, 6.5, 100000)
    ex = dp.epsilon_schedule("exponential", 0.5, 1.0, 0.05, 6.5, 100000)
    np.testing.assert_array_equal(np.array([lin(s) for s in g["steps"]]), g["linear"])
    np.testing.assert_array_equal(np.array([ex(s) for s in g["steps"]]), g["exponential"])


def test_vdn_loss_grad_and_updates_match_reference():
    g = load("learner_vdn_H64.npz")
    D, H, A = int(g["D"]), 64, int(g["A"])
    params = torch.tensor(g["params0"]).requires_grad_(True)
    loss = dp.compute_loss(params, torch.tensor(g["target0"]), batch_of(g, 0), 0.99, True, D, H, A, mode="vdn")
    loss.backward()
    assert abs(loss.item() - float(g["loss0"])) <= 1e-5 * abs(float(g["loss0"]))
    np.testing.assert_allclose(params.grad.numpy(), g["grad0"], rtol=1e-4, atol=1e-5)
    lr = dp.Learner(torch.tensor(g["params0"]), D, H, A, target_update_interval_or_tau=2, mode="vdn")
    lr.target = torch.tensor(g["target0"])
    for i in range(3):
        m = lr.update(batch_of(g, i))
        assert abs(m["loss"] - float(g["losses"][i])) <= 2e-5 * abs(float(g["losses"][i]))
        np.testing.assert_allclose(lr.flat().detach().numpy(), g[f"params{i + 1}"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", ["learner_qmix_H64.npz", "learner_qmix_p4_H64.npz"])
def test_qmix_loss_grad_and_updates_match_reference(name):
    """oracle/qmix_port.py against the reference's QMixNetwork (dqn/model.py:272-443): loss, critic and mixer
    gradients, then 3 updates incl. the clip-critic-only rule and the hard target / target-mixer copy."""
    from oracle import qmix_port as qp

    g = load(name)
    P, D, H, A = int(g["P"]), int(g["D"]), 64, int(g["A"])
    params = torch.tensor(g["params0"]).requires_grad_(True)
    mixer = torch.tensor(g["mixer0"]).requires_grad_(True)
    assert mixer.numel() == qp.mixer_nparams(P, P * D)
    loss = qp.compute_loss(params, torch.tensor(g["target0"]), mixer, torch.tensor(g["tmixer0"]), batch_of(g, 0), 0.99, True, D, H, A)
    loss.backward()
    assert abs(loss.item() - float(g["loss0"])) <= 1e-5 * abs(float(g["loss0"]))
    np.testing.assert_allclose(params.grad.numpy(), g["grad0"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(mixer.grad.numpy(), g["mgrad0"], rtol=1e-4, atol=1e-4)
    if "losses" not in g:
        return
    lr = qp.Learner(torch.tensor(g["params0"]), torch.tensor(g["mixer0"]), D, H, A, target_update_interval_or_tau=2)
    lr.target, lr.tmixer = torch.tensor(g["target0"]), torch.tensor(g["tmixer0"])
    for i in range(3):
        m = lr.update(batch_of(g, i))
        assert abs(m["loss"] - float(g["losses"][i])) <= 2e-5 * abs(float(g["losses"][i]))
        np.testing.assert_allclose(lr.flat().detach().numpy(), g[f"params{i + 1}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(lr.mflat().detach().numpy(), g[f"mixer{i + 1}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(lr.tmixer.numpy(), g[f"tmixer{i + 1}"], rtol=0, atol=2e-6)


def ac_batch_of(g, i):
    return {k: torch.tensor(g[f"batch{i}_{k}"]) for k in ("obss", "actions", "rewards", "dones", "filled")}


@pytest.mark.parametrize("name", ["learner_a2c_H64.npz", "learner_a2c_clip_H128.npz", "learner_ppo_H64.npz",
                                  "learner_maa2c_H64.npz", "learner_mappo_p3_H128.npz"])  # the last two: critic.centralised
def test_actor_critic_update_matches_reference(name):
    """oracle/ac_update_port.py against the reference's A2CNetwork / PPONetwork (marlbase/ac/model.py:189-352):
    n-step returns, gradient, metrics and the parameter blocks after 3 updates incl. the step-keyed target copy."""
    from oracle import ac_update_port as ap

    g = load(name)
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    ppo = "ppo" in name
    kw = dict(n_steps=int(g["n_steps"]), gamma=float(g["gamma"]), entropy_coef=float(g["entropy_coef"]),
              value_loss_coef=float(g["value_loss_coef"]))
    actor, critic, target = (torch.tensor(g[k]) for k in ("actor0", "critic0", "target0"))
    if not ppo:
        b = ac_batch_of(g, 0)
        with torch.no_grad():
            nv = ap.values(target, b["obss"], D, H)
            np.testing.assert_allclose(nv.numpy(), g["next_value0"], rtol=1e-5, atol=1e-5)
            done = b["dones"].float().unsqueeze(-1).repeat(1, 1, P)
            np.testing.assert_allclose(ap.nstep_returns(b["rewards"], done, nv, kw["n_steps"], kw["gamma"]).numpy(),
                                       g["returns0"], rtol=1e-5, atol=1e-5)
        a, c = actor.clone().requires_grad_(True), critic.clone().requires_grad_(True)
        loss, m = ap.a2c_loss(a, c, target, b, D, H, A, **kw)
        loss.backward()
        assert abs(loss.item() - g["metrics"][0][0]) <= 1e-5 * abs(g["metrics"][0][0])
        np.testing.assert_allclose(a.grad.numpy(), g["actor_grad0"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(c.grad.numpy(), g["critic_grad0"], rtol=1e-4, atol=1e-6)
    lr = ap.Learner(actor, critic, D, H, A, grad_clip=float(g["grad_clip"]) or False, num_epochs=int(g["num_epochs"]) if ppo else 0,
                    ppo_clip=float(g["ppo_clip"]), **kw)
    lr.target = target
    for i in range(3):
        m = lr.update(ac_batch_of(g, i), int(g["steps"][i]))
        got = [m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]]
        np.testing.assert_allclose(got, g["metrics"][i], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(lr.actor().detach().numpy(), g[f"actor{i + 1}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(lr.critic().detach().numpy(), g[f"critic{i + 1}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(lr.target.numpy(), g[f"target{i + 1}"], rtol=0, atol=2e-6)


def test_actor_critic_port_with_different_depths_matches_reference():
    """actor.layers [64, 64] next to critic.layers [48, 64, 32] (marlbase/ac/model.py:45-97: each family from its own list): the port with the
    critics' own layer list (Hc) against the reference's A2CNetwork - metrics and every block after 3 clipped updates"""
    from oracle import ac_update_port as ap

    g = load("learner_a2c_depths.npz")
    D, A = int(g["D"]), int(g["A"])
    H, Hc = tuple(int(h) for h in g["actor_layers"]), tuple(int(h) for h in g["critic_layers"])
    assert len(H) != len(Hc) and g["critic0"].shape[1] == dp.nparams(D, Hc, 1) and g["actor0"].shape[1] == dp.nparams(D, H, A)
    lr = ap.Learner(torch.tensor(g["actor0"]), torch.tensor(g["critic0"]), D, H, A, grad_clip=float(g["grad_clip"]), Hc=Hc)
    lr.target = torch.tensor(g["target0"])
    for i in range(3):
        m = lr.update(ac_batch_of(g, i), int(g["steps"][i]))
        np.testing.assert_allclose([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]], g["metrics"][i], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(lr.actor().detach().numpy(), g[f"actor{i + 1}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(lr.critic().detach().numpy(), g[f"critic{i + 1}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(lr.target.numpy(), g[f"target{i + 1}"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name,mode", [("learner_shared_H64.npz", "idqn"), ("learner_shared_seps_H64.npz", "vdn")])
def test_parameter_sharing_matches_reference(name, mode):
    """MultiAgentSharedNetwork (utils/models.py:176-300): agents mapped onto K shared networks, gradients tied"""
    g = load(name)
    D, H, A = int(g["D"]), 64, int(g["A"])
    sharing = [int(i) for i in g["sharing"]]
    params = torch.tensor(g["params0"]).requires_grad_(True)
    loss = dp.compute_loss(params, torch.tensor(g["target0"]), batch_of(g, 0), 0.99, True, D, H, A, mode=mode, sharing=sharing)
    loss.backward()
    assert params.shape[0] == len(set(sharing))
    assert abs(loss.item() - float(g["loss0"])) <= 1e-5 * abs(float(g["loss0"]))
    np.testing.assert_allclose(params.grad.numpy(), g["grad0"], rtol=1e-4, atol=1e-5)
    lr = dp.Learner(torch.tensor(g["params0"]), D, H, A, target_update_interval_or_tau=2, mode=mode, sharing=sharing)
    lr.target = torch.tensor(g["target0"])
    for i in range(3):
        m = lr.update(batch_of(g, i))
        assert abs(m["loss"] - float(g["losses"][i])) <= 2e-5 * abs(float(g["losses"][i]))
        np.testing.assert_allclose(lr.flat().detach().numpy(), g[f"params{i + 1}"], rtol=0, atol=2e-6)


def test_standardise_returns_matches_reference():
    """RunningMeanStd (utils/standardise_stream.py) inside QNetwork._compute_loss and A2CNetwork.update: losses / metrics,
    parameters and the running statistics after each of 3 updates"""
    from oracle import ac_update_port as ap

    g = load("learner_std_idqn_H64.npz")
    D, H, A = int(g["D"]), 64, int(g["A"])
    lr = dp.Learner(torch.tensor(g["params0"]), D, H, A, target_update_interval_or_tau=2, standardise_returns=True)
    lr.target = torch.tensor(g["target0"])
    for i in range(3):
        m = lr.update(batch_of(g, i))
        assert abs(m["loss"] - float(g["losses"][i])) <= 2e-5 * abs(float(g["losses"][i]))
        np.testing.assert_allclose(lr.flat().detach().numpy(), g[f"params{i + 1}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(lr.ret_ms.mean.numpy(), g[f"mean{i + 1}"], rtol=1e-6)
        np.testing.assert_allclose(lr.ret_ms.var.numpy(), g[f"var{i + 1}"], rtol=1e-6)
        assert abs(lr.ret_ms.count - float(g[f"count{i + 1}"])) < 1e-6
    g = load("learner_std_a2c_H64.npz")
    al = ap.Learner(torch.tensor(g["actor0"]), torch.tensor(g["critic0"]), D, H, A, standardise_returns=True)
    al.target = torch.tensor(g["target0"])
    for i in range(3):
        m = al.update(ac_batch_of(g, i), int(g["steps"][i]))
        np.testing.assert_allclose([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]], g["metrics"][i], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(al.actor().detach().numpy(), g[f"actor{i + 1}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(al.ret_ms.mean.numpy(), g[f"mean{i + 1}"], rtol=1e-6)
        np.testing.assert_allclose(al.ret_ms.var.numpy(), g[f"var{i + 1}"], rtol=1e-6)
