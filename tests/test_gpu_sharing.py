"""GPU parity of parameter sharing (MultiAgentSharedNetwork, marlbase/utils/models.py:176-300) through the C-ABI's
agent -> network map (marlhip_net_shape.net_of): learner vs the reference's own shared QNetwork / SePS VDNetwork
goldens; act / fused collector with a shared block == the same kernels fed the expanded per-agent blocks."""
import numpy as np
import pytest
import torch

from oracle import dqn_port as dp
from tests.test_gpu_parity import DEV, dev_batch, golden_batch, hip, load, make_env

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,mode,H", [("learner_shared_H64.npz", 0, 64), ("learner_shared_seps_H64.npz", 1, 64)])
def test_shared_learner_matches_reference_golden(name, mode, H):
    h = hip()
    g = load(name)
    P, D, A = int(g["P"]), int(g["D"]), int(g["A"])
    sharing = tuple(int(i) for i in g["sharing"])
    spec = h.NetSpec(P, D, H, A, sharing)
    params, target = torch.tensor(g["params0"], device=DEV), torch.tensor(g["target0"], device=DEV)
    assert params.shape[0] == spec.n_blocks
    up = h.DqnUpdater(spec, params, target, lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True)
    loss, grad = up.loss_grad(dev_batch(h, golden_batch(g, 0)), mode=mode)
    assert abs(loss.cpu().numpy()[0] - g["loss0"]) <= 1e-5 * abs(g["loss0"])
    np.testing.assert_allclose(grad.cpu().numpy(), g["grad0"], rtol=1e-4, atol=2e-5)
    last = 0
    for i in range(3):
        loss, _ = up.loss_grad(dev_batch(h, golden_batch(g, i)), mode=mode)
        hard = (i + 1 - last) >= 2
        up.apply(hard_update=hard)
        if hard:
            last = i + 1
        assert abs(loss.cpu().numpy()[0] - g["losses"][i]) <= 2e-5 * abs(g["losses"][i])
        np.testing.assert_allclose(params.cpu().numpy(), g[f"params{i + 1}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(target.cpu().numpy(), g[f"target{i + 1}"], rtol=0, atol=3e-6)


@pytest.mark.parametrize("H", [64, 128])
def test_shared_blocks_equal_expanded_independent_blocks(H):
    """act, fused collector and the learner's loss with a [K][n] shared block == the same calls on the [P][n] expansion
    (bitwise for act / collector / loss; gradient = per-network sum of the per-agent gradients)"""
    h = hip()
    P, D, A, T, N = 3, 18, 6, 25, 96
    sharing = (0, 1, 0)
    shared = (dp.init_params(2, D, H, A, seed=4) + 0.02).to(DEV)
    tshared = dp.init_params(2, D, H, A, seed=5).to(DEV)
    expanded, texpanded = shared[list(sharing)].contiguous(), tshared[list(sharing)].contiguous()
    s_sh, s_in = h.NetSpec(P, D, H, A, sharing), h.NetSpec(P, D, H, A)
    obs = torch.randint(-1, 8, (P, N, D), device=DEV).float()
    u = torch.rand(N, device=DEV)
    ra = torch.randint(0, A, (P, N), dtype=torch.int32, device=DEV)
    q1, q2 = torch.empty(P, N, A, device=DEV), torch.empty(P, N, A, device=DEV)
    a1 = h.dqn_act(s_sh, shared, obs, 0.3, u=u, rand_actions=ra, q_out=q1)
    a2 = h.dqn_act(s_in, expanded, obs, 0.3, u=u, rand_actions=ra, q_out=q2)
    assert torch.equal(a1, a2) and torch.equal(q1, q2)
    # fused collector
    outs = []
    for spec, blk in ((s_sh, shared), (s_in, expanded)):
        cfg = h.lbf_config("lbforaging:Foraging-8x8-3p-3f-v3", N, T, seed=11)
        rb = h.DeviceReplay(N, P, D, T)
        fr, fl = torch.zeros(P, N, device=DEV), torch.zeros(N, dtype=torch.int32, device=DEV)
        h.idqn_collect(cfg, spec, blk, 0.2, 3, rb, 0, fr, fl)
        outs.append((rb.obs.clone(), rb.act.clone(), rb.rew.clone(), fr, fl))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    # learner
    batch = dp.synthetic_batch(P, T, 40, D, A, seed=9)
    up_sh = h.DqnUpdater(s_sh, shared.clone(), tshared.clone())
    up_in = h.DqnUpdater(s_in, expanded.clone(), texpanded.clone())
    l1, g1 = up_sh.loss_grad(dev_batch(h, batch))
    l2, g2 = up_in.loss_grad(dev_batch(h, batch))
    assert torch.equal(l1, l2)
    want = torch.zeros_like(g1).index_add_(0, torch.tensor(sharing, device=DEV), g2)
    np.testing.assert_allclose(g1.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_shared_qnetwork_host_interface():
    """parameter_sharing=True / SePS list through the reference-shaped QNetwork: key names, init RNG order, update"""
    from codebase_amd.dqn.model import QNetwork, VDNetwork
    from codebase_amd.spaces import Box, Discrete, Tuple

    for name, cls, P, D in (("learner_shared_H64.npz", QNetwork, 2, 15), ("learner_shared_seps_H64.npz", VDNetwork, 3, 18)):
        g = load(name)
        sharing = [int(i) for i in g["sharing"]]
        obs_space = Tuple([Box(-1, 8, (D,)) for _ in range(P)])
        act_space = Tuple([Discrete(6) for _ in range(P)])
        hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False,
                     target_update_interval_or_tau=2)
        nt = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            torch.manual_seed(800 if P == 2 else 900)
            net = cls(obs_space, act_space, hyper, [64, 64], True if P == 2 else sharing, False, True, "cuda")
        finally:
            torch.set_num_threads(nt)
        assert list(net.state_dict().keys()) == [str(k) for k in g["keys"]]
        np.testing.assert_allclose(net.params.cpu().numpy(), g["init"], rtol=0, atol=2e-5)  # orthogonal init: LAPACK QR rounding
        net.params.copy_(torch.tensor(g["params0"]))
        net.target_params.copy_(torch.tensor(g["target0"]))
        for i in range(3):
            b = golden_batch(g, i)
            m = net.update(h_batch(b))
            assert abs(m["loss"] - g["losses"][i]) <= 2e-5 * abs(g["losses"][i])
        np.testing.assert_allclose(net.params.cpu().numpy(), g["params3"], rtol=0, atol=3e-6)
        acts, _ = net.act([np.zeros(D, np.float32)] * P, net.init_hiddens(1), 0.0)
        assert len(acts) == P


def h_batch(b):
    from codebase_amd import hip as _h
    return _h.Batch(b["obss"], b["actions"], b["rewards"], b["dones"], b["filled"], None)
