"""batch.action_mask (dqn/model.py:100-111,133-143; dqn/train.py:55-124): the oracle port against goldens produced by the
reference's own QNetwork / VDNetwork with masks (CPU), and the HIP learner kernels + the masked act / ReplayBuffer adapter
against the same goldens (GPU).  No env on this path emits masks; they enter through Batch arguments and the scalar loop."""
import os

import numpy as np
import pytest
import torch

from oracle import dqn_port as dp

G = os.path.join(os.path.dirname(__file__), "golden")
FILES = [("learner_masks_idqn_H64.npz", "idqn"), ("learner_masks_vdn_H128.npz", "vdn")]


def load(name):
    g = dict(np.load(os.path.join(G, name)))
    batch = {k[6:]: torch.tensor(v) for k, v in g.items() if k.startswith("batch_")}
    return g, batch


@pytest.mark.parametrize("name,mode", FILES)
def test_oracle_port_matches_reference_with_masks(name, mode):
    g, batch = load(name)
    D, H, A = int(g["D"]), int(g["H"]), int(g["A"])
    for tag, dq in (("dq", True), ("max", False)):
        pr = torch.tensor(g["params0"]).requires_grad_(True)
        loss = dp.compute_loss(pr, torch.tensor(g["target0"]), batch, 0.99, dq, D, H, A, mode=mode)
        loss.backward()
        assert abs(loss.item() - g[f"loss_{tag}"]) <= 1e-5 * abs(g[f"loss_{tag}"])
        np.testing.assert_allclose(pr.grad.numpy(), g[f"grad_{tag}"], rtol=1e-4, atol=1e-5)
        nomask = dp.compute_loss(pr.detach(), torch.tensor(g["target0"]), {k: v for k, v in batch.items() if k != "action_mask"},
                                 0.99, dq, D, H, A, mode=mode)
        assert abs(nomask.item() - g[f"loss_nomask_{tag}"]) <= 1e-5 * abs(g[f"loss_nomask_{tag}"])
        assert abs(g[f"loss_{tag}"] - g[f"loss_nomask_{tag}"]) > 1.0  # the fixture exercises the mask


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode", FILES)
def test_hip_learner_with_masks_matches_reference(name, mode):
    from codebase_amd import hip as h

    g, batch = load(name)
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    spec = h.NetSpec(P, D, H, A)
    dev = lambda b: h.Batch(*(b[k].cuda().contiguous() for k in ("obss", "actions", "rewards", "dones", "filled")),  # noqa: E731
                            b["action_mask"].cuda().contiguous())
    for tag, dq in (("dq", True), ("max", False)):
        up = h.DqnUpdater(spec, torch.tensor(g["params0"]).cuda(), torch.tensor(g["target0"]).cuda(), double_q=dq)
        loss, grad = up.loss_grad(dev(batch), mode=1 if mode == "vdn" else 0)
        assert abs(loss.cpu().numpy()[0] - g[f"loss_{tag}"]) <= 2e-5 * abs(g[f"loss_{tag}"])
        gref = g[f"grad_{tag}"]
        np.testing.assert_allclose(grad.cpu().numpy(), gref, rtol=3e-4, atol=3e-5 * max(1.0, np.abs(gref).max()))
    # two updates through the model class (mask carried by the Batch), then the masked greedy act
    from codebase_amd.dqn.model import QNetwork, VDNetwork
    from codebase_amd.spaces import Box, Discrete, Tuple

    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False,
                 target_update_interval_or_tau=200)
    net = (VDNetwork if mode == "vdn" else QNetwork)(Tuple([Box(-1, 8, (D,))] * P), Tuple([Discrete(A)] * P), hyper, [H, H], False,
                                                     False, True, "cuda")
    net.params.copy_(torch.tensor(g["params0"]))
    net.target_params.copy_(torch.tensor(g["target0"]))
    b = h.Batch(*(batch[k] for k in ("obss", "actions", "rewards", "dones", "filled")), batch["action_mask"])
    losses = [net.update(b)["loss"] for _ in range(2)]
    np.testing.assert_allclose(losses, g["losses"], rtol=3e-5)
    np.testing.assert_allclose(net.params.cpu().numpy(), g["params2"], rtol=0, atol=3e-6)
    net.params.copy_(torch.tensor(g["act_params"]))
    for obs, am, want in zip(g["act_obs"], g["act_mask"], g["act_actions"]):
        acts, _ = net.act([o for o in obs], None, 0.0, [m for m in am])
        q = net.q_values(torch.tensor(obs).cuda().reshape(P, 1, D))[:, 0].cpu().numpy()
        for p in range(P):  # ties within fp32 roundoff aside, the same allowed action as the reference
            allowed = np.where(am[p] == 1)[0]
            top = np.sort(q[p][allowed])
            if len(top) == 1 or top[-1] - top[-2] > 1e-4:
                assert acts[p] == want[p]
            assert am[p][acts[p]] == 1


@pytest.mark.gpu
def test_replay_adapter_stores_and_samples_masks():
    from codebase_amd import spaces
    from codebase_amd.dqn.train import ReplayBuffer

    P, D, A, T, CAP = 2, 15, 6, 5, 4
    rb = ReplayBuffer(CAP, P, spaces.Tuple([spaces.Box(-1.0, 8.0, shape=(D,))] * P), spaces.Tuple([spaces.Discrete(A)] * P), T, "cuda",
                      store_action_masks=True)
    rng = np.random.default_rng(0)
    ref = np.zeros((P, T + 1, CAP, A), np.float32)
    for ep in range(6):  # wraps the ring
        slot = ep % CAP
        m = (rng.random((P, A)) < 0.5).astype(np.float32)
        rb.init_episode([rng.random(D).astype(np.float32) for _ in range(P)], m)
        ref[:, 0, slot] = m
        for t in range(3):
            m = (rng.random((P, A)) < 0.5).astype(np.float32)
            rb.add([rng.random(D).astype(np.float32) for _ in range(P)], [1, 2], [0.0, 1.0], t == 2, m)
            ref[:, t + 1, slot] = m
    orig = np.random.randint
    np.random.randint = lambda lo, hi, size: np.array([3, 0, 0, 2])[:size]
    try:
        b = rb.sample(4)
    finally:
        np.random.randint = orig
    np.testing.assert_array_equal(b.action_mask.cpu().numpy(), ref[:, :, [3, 0, 0, 2]])
    with pytest.raises(AssertionError):
        ReplayBuffer(CAP, P, spaces.Tuple([spaces.Box(-1.0, 8.0, shape=(D,))] * P), spaces.Tuple([spaces.Discrete(A)] * P), T,
                     "cuda").init_episode([np.zeros(D, np.float32)] * P, np.ones((P, A), np.float32))


# ---- actor-critic learners: batch.action_masks (ac/train.py:53-63, ac/model.py:135-145,205-214,283-306)
AC_FILES = ["learner_a2c_masks_H64.npz", "learner_ppo_masks_H128.npz"]


def ac_batch(g, i):
    return {k: torch.tensor(g[f"batch{i}_{k}"]) for k in ("obss", "actions", "rewards", "dones", "filled", "action_masks")}


@pytest.mark.parametrize("name", AC_FILES)
def test_ac_oracle_port_matches_reference_with_masks(name):
    from oracle import ac_update_port as ap

    g = dict(np.load(os.path.join(G, name)))
    D, H, A = int(g["D"]), int(g["H"]), int(g["A"])
    lr = ap.Learner(torch.tensor(g["actor0"]), torch.tensor(g["critic0"]), D, H, A, gamma=float(g["gamma"]), n_steps=int(g["n_steps"]),
                    entropy_coef=float(g["entropy_coef"]), value_loss_coef=float(g["value_loss_coef"]), grad_clip=float(g["grad_clip"]) or False,
                    num_epochs=int(g["num_epochs"]) if "ppo" in name else 0, ppo_clip=float(g["ppo_clip"]))
    lr.target = torch.tensor(g["target0"])
    for i in range(3):
        m = lr.update(ac_batch(g, i), int(g["steps"][i]))
        np.testing.assert_allclose([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]], g["metrics"][i], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(lr.actor().detach().numpy(), g[f"actor{i + 1}"], rtol=0, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", AC_FILES)
def test_hip_ac_learner_with_masks_matches_reference(name):
    from collections import namedtuple

    from codebase_amd import hip as h

    Batch = namedtuple("Batch", ["obss", "actions", "rewards", "dones", "filled", "action_masks"])
    g = dict(np.load(os.path.join(G, name)))
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    spec = h.NetSpec(P, D, H, A)
    block = torch.cat([torch.tensor(g["actor0"]).reshape(-1), torch.tensor(g["critic0"]).reshape(-1)]).cuda()
    up = h.AcUpdater(spec, block, torch.tensor(g["target0"]).cuda().contiguous(), lr=3e-4, gamma=float(g["gamma"]), n_steps=int(g["n_steps"]),
                     entropy_coef=float(g["entropy_coef"]), value_loss_coef=float(g["value_loss_coef"]), grad_clip=False,
                     ppo_clip=float(g["ppo_clip"]))
    dev = lambda b: Batch(*(b[k].cuda() for k in ("obss", "actions", "rewards", "dones", "filled", "action_masks")))  # noqa: E731
    ppo = "ppo" in name
    for i in range(3):
        b = dev(ac_batch(g, i))
        if ppo:
            up.ppo_prepare(b)
            acc = np.zeros(4)
            for _ in range(int(g["num_epochs"])):
                acc += up.ppo_loss_grad(b).cpu().numpy()[:4]
                up.apply()
            m = acc / int(g["num_epochs"])
        else:
            m = up.a2c_loss_grad(b).cpu().numpy()[:4]
            if i == 0:
                np.testing.assert_allclose(up.actor_grad.cpu().numpy(), g["actor_grad0"], rtol=1e-4, atol=1e-4 * np.abs(g["actor_grad0"]).max())
            up.apply()
        np.testing.assert_allclose(m, g["metrics"][i], rtol=5e-5, atol=5e-6)
        if int(g["steps"][i]) % 200 == 0:  # step-keyed hard target copy (model.py:233-239)
            up.target_critic.copy_(up.critic)
        np.testing.assert_allclose(up.block[:P * up.n_actor].cpu().numpy().reshape(P, -1), g[f"actor{i + 1}"], rtol=0, atol=3e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,H,dq", [("vdn", 64, True), ("idqn", 128, True), ("idqn", 128, False), ("vdn", 64, False)])
def test_hip_learner_with_masks_other_kernel_paths_vs_port(mode, H, dq):
    """the remaining (kernel family, mode) pairs against the oracle port that the goldens above pin: the fused kernel's
    forward-only pass (VDN / QMIX at hidden 64) and the tensor-parallel pass F on independent learners"""
    from codebase_amd import hip as h
    from oracle.make_golden_masks import random_mask

    P, T, B, D, A = 3, 9, 29, 18, 6
    spec = h.NetSpec(P, D, H, A)
    params = dp.init_params(P, D, H, A, seed=1) + 0.04
    target = dp.init_params(P, D, H, A, seed=3)
    batch = dp.synthetic_batch(P, T, B, D, A, seed=5)
    if mode == "vdn":
        batch["rewards"][1:] = batch["rewards"][0]
    batch["action_mask"] = random_mask(P, T, B, A, batch["actions"], 77)
    pr = params.clone().requires_grad_(True)
    ref = dp.compute_loss(pr, target, batch, 0.99, dq, D, H, A, mode=mode)
    ref.backward()
    up = h.DqnUpdater(spec, params.cuda(), target.cuda(), double_q=dq)
    hb = h.Batch(*(batch[k].cuda().contiguous() for k in ("obss", "actions", "rewards", "dones", "filled", "action_mask")))
    loss, grad = up.loss_grad(hb, mode=1 if mode == "vdn" else 0)
    assert abs(loss.cpu().numpy()[0] - ref.item()) <= 3e-5 * abs(ref.item())
    gref = pr.grad.numpy()
    np.testing.assert_allclose(grad.cpu().numpy(), gref, rtol=3e-4, atol=3e-5 * max(1.0, np.abs(gref).max()))
