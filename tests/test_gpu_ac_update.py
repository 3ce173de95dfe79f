"""GPU parity of the actor-critic learner step (csrc/a2c.hip: marlhip_a2c_loss_grad, marlhip_ppo_*, marlhip_ac_forward_rows)
through the C-ABI: against the reference's own A2CNetwork / PPONetwork (tests/golden/learner_a2c_*.npz, learner_ppo_H64.npz)
and against the oracle port (oracle/ac_update_port.py) on other shapes.  fp32: metrics to 2e-5 relative, gradients to 1e-4
of their largest entry, parameter blocks after 3 updates to 3e-6 absolute."""
from collections import namedtuple

import numpy as np
import pytest
import torch

from oracle import ac_update_port as ap
from oracle import dqn_port as dp
from tests.test_gpu_parity import DEV, hip, load

pytestmark = pytest.mark.gpu
Batch = namedtuple("Batch", ["obss", "actions", "rewards", "dones", "filled", "action_masks"])


def dev_ac_batch(b):
    return Batch(b["obss"].to(DEV), b["actions"].to(DEV), b["rewards"].to(DEV), b["dones"].to(DEV), b["filled"].to(DEV), None)


def golden_ac_batch(g, i):
    return {k: torch.tensor(g[f"batch{i}_{k}"]) for k in ("obss", "actions", "rewards", "dones", "filled")}


def make(h, g, actor, critic, target, **kw):
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    spec = h.NetSpec(P, D, H, A)
    block = torch.cat([actor.reshape(-1), critic.reshape(-1)]).to(DEV)
    kw["centralised_critic"] = P > 1 and critic.shape[1] == dp.nparams(P * D, H, 1)  # maa2c / mappo goldens
    return h.AcUpdater(spec, block, target.to(DEV).contiguous(), lr=3e-4, gamma=float(g["gamma"]), n_steps=int(g["n_steps"]),
                       entropy_coef=float(g["entropy_coef"]), value_loss_coef=float(g["value_loss_coef"]),
                       grad_clip=float(g["grad_clip"]) or False, ppo_clip=float(g["ppo_clip"]), **kw)


def assert_grad_close(got, ref, rel=1e-4):
    np.testing.assert_allclose(got, ref, rtol=rel, atol=rel * max(1e-4, float(np.abs(ref).max())))


@pytest.mark.parametrize("name", ["learner_a2c_H64.npz", "learner_a2c_clip_H128.npz", "learner_maa2c_H64.npz"])
def test_a2c_gradient_metrics_and_updates_match_reference_golden(name):
    h = hip()
    g = load(name)
    P, D, H = int(g["P"]), int(g["D"]), int(g["H"])
    t = lambda k: torch.tensor(g[k])  # noqa: E731
    up = make(h, g, t("actor0"), t("critic0"), t("target0"))
    # forward pieces: target-critic values on every observation (A2CNetwork.get_value(target=True))
    b0 = golden_ac_batch(g, 0)
    T, N = b0["filled"].shape
    nv = h.ac_forward_rows(up.spec, up.target_critic, b0["obss"].to(DEV), D, P * D, (T + 1) * N, value_net=2 if up.centralised else 1)
    np.testing.assert_allclose(nv.reshape(P, T + 1, N).permute(1, 2, 0).cpu().numpy(), g["next_value0"], rtol=1e-5, atol=1e-5)
    m = up.a2c_loss_grad(dev_ac_batch(b0)).cpu().numpy()
    np.testing.assert_allclose(m[:4], g["metrics"][0], rtol=2e-5, atol=2e-6)
    assert m[4] == b0["filled"].sum().item()
    assert_grad_close(up.actor_grad.cpu().numpy(), g["actor_grad0"])
    assert_grad_close(up.critic_grad.cpu().numpy(), g["critic_grad0"])
    tui = 200
    for i in range(3):
        step = int(g["steps"][i])
        m = up.a2c_loss_grad(dev_ac_batch(golden_ac_batch(g, i))).cpu().numpy()
        up.apply()
        if step % tui == 0:
            up.target_critic.copy_(up.critic)
        np.testing.assert_allclose(m[:4], g["metrics"][i], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(up.actor.cpu().numpy(), g[f"actor{i + 1}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(up.critic.cpu().numpy(), g[f"critic{i + 1}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(up.target_critic.cpu().numpy(), g[f"target{i + 1}"], rtol=0, atol=3e-6)


@pytest.mark.parametrize("name", ["learner_ppo_H64.npz", "learner_mappo_p3_H128.npz"])
def test_ppo_updates_match_reference_golden(name):
    """PPONetwork.update: returns + old log-probs once, 4 epochs of the clipped surrogate, metrics averaged over epochs
    (mappo: centralised critics of 54 inputs on the tensor-parallel hidden-128 path)"""
    h = hip()
    g = load(name)
    t = lambda k: torch.tensor(g[k])  # noqa: E731
    up = make(h, g, t("actor0"), t("critic0"), t("target0"))
    for i in range(3):
        b = dev_ac_batch(golden_ac_batch(g, i))
        up.ppo_prepare(b)
        acc = np.zeros(4)
        for _ in range(int(g["num_epochs"])):
            acc += up.ppo_loss_grad(b).cpu().numpy()[:4]
            up.apply()
        if int(g["steps"][i]) % 200 == 0:
            up.target_critic.copy_(up.critic)
        np.testing.assert_allclose(acc / int(g["num_epochs"]), g["metrics"][i], rtol=5e-5, atol=5e-6)
        np.testing.assert_allclose(up.actor.cpu().numpy(), g[f"actor{i + 1}"], rtol=0, atol=5e-6)
        np.testing.assert_allclose(up.critic.cpu().numpy(), g[f"critic{i + 1}"], rtol=0, atol=5e-6)


@pytest.mark.parametrize("P,T,N,D,H,n", [(2, 25, 33, 15, 64, 5), (4, 7, 16, 27, 64, 1), (8, 25, 20, 39, 128, 10), (3, 25, 130, 24, 128, 5),
                                         (2, 2, 1, 12, 64, 3), (4, 25, 600, 21, 64, 5),
                                         # rollouts of whole 16-row blocks on the tensor-parallel kernels (hidden 128; the 71-wide warehouse rows
                                         # at either width): the forward-rows pass stores the second hidden layer and the backward pass reads it
                                         # back (tp_bwd_kernel<FULL, STORED>) - one block, an odd block count, several steps per workgroup
                                         (2, 25, 32, 15, 128, 5), (3, 25, 144, 24, 128, 5), (2, 25, 2048, 15, 128, 5), (4, 12, 48, 71, 128, 5),
                                         (2, 6, 304, 71, 64, 2), (8, 9, 64, 39, 128, 3),
                                         # hidden 64 with whole row blocks: the forward-rows pass keeps h1 | h2 for dqn_lossgrad_kernel<MODE 4, STORED>
                                         (2, 25, 2048, 15, 64, 5), (3, 12, 48, 24, 64, 5), (8, 5, 32, 39, 64, 2)])
def test_a2c_other_shapes_vs_oracle_port(P, T, N, D, H, n):
    h = hip()
    A = 5 if D == 71 else 6
    actor = dp.init_params(P, D, H, A, seed=1) + 0.03
    critic = torch.stack([dp.init_params(1, D, H, 1, seed=20 + p)[0] for p in range(P)]) + 0.02
    target = torch.stack([dp.init_params(1, D, H, 1, seed=40 + p)[0] for p in range(P)])
    batch = ap.synthetic_batch(P, T, N, D, A, seed=7)
    a, c = actor.clone().requires_grad_(True), critic.clone().requires_grad_(True)
    loss, m = ap.a2c_loss(a, c, target, batch, D, H, A, n_steps=n, gamma=0.97, entropy_coef=0.01, value_loss_coef=0.5)
    loss.backward()
    spec = h.NetSpec(P, D, H, A)
    up = h.AcUpdater(spec, torch.cat([actor.reshape(-1), critic.reshape(-1)]).to(DEV), target.to(DEV).contiguous(), gamma=0.97,
                     n_steps=n, entropy_coef=0.01, value_loss_coef=0.5)
    got = up.a2c_loss_grad(dev_ac_batch(batch)).cpu().numpy()
    ref = [m["loss"].item(), m["actor_loss"].item(), m["value_loss"].item(), m["entropy"].item()]
    np.testing.assert_allclose(got[:4], ref, rtol=5e-5, atol=5e-6)
    assert_grad_close(up.actor_grad.cpu().numpy(), a.grad.numpy(), 3e-4)
    assert_grad_close(up.critic_grad.cpu().numpy(), c.grad.numpy(), 3e-4)


@pytest.mark.parametrize("P,T,N,D,H", [(2, 25, 32, 15, 128), (3, 12, 144, 24, 128), (2, 25, 33, 15, 128), (4, 10, 64, 71, 64), (2, 25, 256, 15, 64)])
def test_ppo_epoch_after_a_parameter_change_vs_oracle_port(P, T, N, D, H):
    """PPONetwork.update's epoch (ac/model.py:300-352) with ratios off 1: returns and old log-probs from the parameters at prepare time,
    the clipped surrogate and its gradients at perturbed ones - on rollouts of whole 16-row blocks (the epoch's forward-rows pass stores
    the second hidden layer for its backward pass at hidden 128 / 71-wide rows) and on a ragged one (the recomputing form)"""
    h = hip()
    A = 5 if D == 71 else 6
    actor0 = dp.init_params(P, D, H, A, seed=1) + 0.03
    critic0 = torch.stack([dp.init_params(1, D, H, 1, seed=20 + p)[0] for p in range(P)]) + 0.02
    target = torch.stack([dp.init_params(1, D, H, 1, seed=40 + p)[0] for p in range(P)])
    g = torch.Generator().manual_seed(9)
    actor1 = actor0 + 0.02 * torch.randn(actor0.shape, generator=g)
    critic1 = critic0 + 0.02 * torch.randn(critic0.shape, generator=g)
    batch = ap.synthetic_batch(P, T, N, D, A, seed=7)
    with torch.no_grad():
        returns, _, old_logp, _ = ap.evaluate(actor0, critic0, target, batch, D, H, A, 5, 0.97)
    a, c = actor1.clone().requires_grad_(True), critic1.clone().requires_grad_(True)
    loss, m = ap.ppo_loss(a, c, returns, old_logp, batch, D, H, A, 0.01, 0.5, 0.2)
    loss.backward()
    spec = h.NetSpec(P, D, H, A)
    up = h.AcUpdater(spec, torch.cat([actor0.reshape(-1), critic0.reshape(-1)]).to(DEV), target.to(DEV).contiguous(), gamma=0.97,
                     n_steps=5, entropy_coef=0.01, value_loss_coef=0.5, ppo_clip=0.2)
    b = dev_ac_batch(batch)
    up.ppo_prepare(b)
    up.actor.copy_(actor1)
    up.critic.copy_(critic1)
    got = up.ppo_loss_grad(b).cpu().numpy()
    ref = [m["loss"].item(), m["actor_loss"].item(), m["value_loss"].item(), m["entropy"].item()]
    np.testing.assert_allclose(got[:4], ref, rtol=5e-5, atol=5e-6)
    assert_grad_close(up.actor_grad.cpu().numpy(), a.grad.numpy(), 3e-4)
    assert_grad_close(up.critic_grad.cpu().numpy(), c.grad.numpy(), 3e-4)


def test_a2c_network_interface_and_algorithm_end_to_end(tmp_path, monkeypatch):
    """A2CNetwork / PPONetwork: reference constructor, state_dict keys, act / get_value shapes, then +algorithm=ia2c and
    +algorithm=ippo through the reference-shaped driver (fused collector + one update per rollout)"""
    from codebase_amd import run
    from codebase_amd.ac.model import A2CNetwork
    from codebase_amd.spaces import Box, Discrete, Tuple

    obs_space = Tuple([Box(-1, 8, (15,)) for _ in range(2)])
    act_space = Tuple([Discrete(6) for _ in range(2)])
    cfg = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=False, n_steps=5, entropy_coef=0.001, value_loss_coef=0.5,
               standardise_returns=False, target_update_interval_or_tau=200)
    net_cfg = dict(layers=[64, 64], parameter_sharing=False, use_orthogonal_init=True, use_rnn=False)
    torch.manual_seed(3)
    net = A2CNetwork(obs_space, act_space, cfg, net_cfg, dict(net_cfg, centralised=False), "cuda")
    sd = net.state_dict()
    want = [f"{pre}.independent.{i}.network.{l}.{wb}" for pre in ("actor", "critic", "target_critic") for i in range(2)
            for l in (0, 2, 4) for wb in ("weight", "bias")]
    assert list(sd.keys()) == want
    assert sd["critic.independent.1.network.4.weight"].shape == (1, 64) and sd["actor.independent.0.network.4.bias"].shape == (6,)
    assert torch.equal(sd["critic.independent.0.network.0.weight"], sd["target_critic.independent.0.network.0.weight"])
    obs = [torch.rand(5, 15) for _ in range(2)]
    acts, _ = net.act(obs, net.init_actor_hiddens(5))
    assert acts.shape == (2, 5, 1) and acts.dtype == torch.int64 and int(acts.max()) < 6
    v, _ = net.get_value(obs, None)
    assert v.shape == (5, 2)
    # value matches the oracle MLP on the same block
    ref = torch.cat([dp.mlp(net.critic_params[p].cpu(), obs[p], 15, 64, 1) for p in range(2)], dim=-1)
    np.testing.assert_allclose(v.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    NAME = "lbforaging:Foraging-8x8-2p-3f-v3"
    for algo in ("ia2c", "ippo"):
        monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / algo))
        df = run.main([f"+algorithm={algo}", f"env.name={NAME}", "env.time_limit=25", "env.parallel_envs=256",
                       "algorithm.model.actor.layers=[64,64]", "algorithm.model.critic.layers=[64,64]", "seed=1",
                       "algorithm.total_steps=60000", "algorithm.eval_interval=20000"])
        assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all() and np.isfinite(df["mean_episode_returns"]).all()
        assert {"actor_loss", "value_loss", "entropy"} <= set(df.columns)


def test_ia2c_learns_on_the_device_path(tmp_path, monkeypatch):
    """end to end: fused rollout collector + device update raise the mean episode return on Foraging-8x8-2p-3f"""
    from codebase_amd import run

    monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / "learn"))
    df = run.main(["+algorithm=ia2c", "env.name=lbforaging:Foraging-8x8-2p-3f-v3", "env.time_limit=25", "env.parallel_envs=1024",
                   "algorithm.model.actor.layers=[64,64]", "algorithm.model.critic.layers=[64,64]", "seed=0",
                   "algorithm.total_steps=30000000", "algorithm.eval_interval=5000000", "algorithm.entropy_coef=0.01"])
    r = df["mean_episode_returns"].to_numpy()
    assert r[-1] > r[0] + 0.05, r


def test_centralised_critic_algorithms_end_to_end(tmp_path, monkeypatch):
    """+algorithm=maa2c / mappo (critic.centralised: True, configs/algorithm/maa2c.yaml, mappo.yaml): state_dict shapes, get_value on
    the concatenated observations vs the oracle MLP, and the drivers"""
    from codebase_amd import run
    from codebase_amd.ac.model import A2CNetwork
    from codebase_amd.spaces import Box, Discrete, Tuple

    obs_space = Tuple([Box(-1, 8, (15,)) for _ in range(2)])
    act_space = Tuple([Discrete(6) for _ in range(2)])
    cfg = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=False, n_steps=5, entropy_coef=0.001, value_loss_coef=0.5,
               standardise_returns=False, target_update_interval_or_tau=200)
    net_cfg = dict(layers=[128, 128], parameter_sharing=False, use_orthogonal_init=True, use_rnn=False)
    net = A2CNetwork(obs_space, act_space, cfg, net_cfg, dict(net_cfg, centralised=True), "cuda")
    sd = net.state_dict()
    assert sd["critic.independent.0.network.0.weight"].shape == (128, 30) and sd["actor.independent.0.network.0.weight"].shape == (128, 15)
    obs = [torch.rand(7, 15) for _ in range(2)]
    v, _ = net.get_value(obs, None, target=True)
    ref = torch.cat([dp.mlp(net.target_critic_params[p].cpu(), torch.cat(obs, dim=-1), 30, 128, 1) for p in range(2)], dim=-1)
    np.testing.assert_allclose(v.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    for algo in ("maa2c", "mappo"):
        monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / algo))
        df = run.main([f"+algorithm={algo}", "env.name=lbforaging:Foraging-8x8-2p-3f-v3", "env.time_limit=25", "env.parallel_envs=256",
                       "seed=1", "algorithm.total_steps=60000", "algorithm.eval_interval=20000"])
        assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all()
    # 8 agents (312-wide critics) and the warehouse (4 x 71): the critics take the GEMM path of csrc/wide_mlp.h
    for algo, env, tl in (("maa2c", "lbforaging:Foraging-15x15-8p-5f-v3", 25), ("mappo", "rware:rware-tiny-4ag-v2", 50)):
        monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / (algo + "_wide")))
        df = run.main([f"+algorithm={algo}", f"env.name={env}", f"env.time_limit={tl}", "env.parallel_envs=128", "seed=1",
                       "algorithm.total_steps=60000", "algorithm.eval_interval=20000"])
        assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all()


@pytest.mark.parametrize("P,T,N,D,H,A,n", [(8, 25, 20, 39, 128, 6, 5), (8, 9, 70, 39, 64, 6, 3), (4, 12, 33, 71, 128, 5, 5), (2, 6, 300, 71, 64, 5, 2),
                                           (5, 7, 130, 27, 64, 6, 5),
                                           # more 64-row tiles than the persistent backward kernel has workgroups (512 / P per agent): every
                                           # workgroup accumulates dW2 and its column sums over several tiles
                                           (8, 10, 1000, 39, 128, 6, 5), (4, 40, 520, 71, 64, 5, 3)])
def test_wide_centralised_critics_vs_oracle_port(P, T, N, D, H, A, n):
    """centralised critics the register-resident kernels do not cover (8 LBF agents: 312 inputs; the warehouse: 71 per agent; 5 agents:
    135-wide rows that are not 16-byte aligned): csrc/wide_critic.h (layer 1 streamed through LDS, the three layers fused, hidden layers
    kept for the backward pass) - loss pieces, actor and critic gradients vs the port, and the value rows through marlhip_ac_forward_rows"""
    h = hip()
    actor = dp.init_params(P, D, H, A, seed=1) + 0.03
    critic = torch.stack([dp.init_params(1, P * D, H, 1, seed=20 + p)[0] for p in range(P)]) + 0.02
    target = torch.stack([dp.init_params(1, P * D, H, 1, seed=40 + p)[0] for p in range(P)])
    batch = ap.synthetic_batch(P, T, N, D, A, seed=7)
    a, c = actor.clone().requires_grad_(True), critic.clone().requires_grad_(True)
    loss, m = ap.a2c_loss(a, c, target, batch, D, H, A, n_steps=n, gamma=0.97, entropy_coef=0.01, value_loss_coef=0.5)
    loss.backward()
    spec = h.NetSpec(P, D, H, A)
    up = h.AcUpdater(spec, torch.cat([actor.reshape(-1), critic.reshape(-1)]).to(DEV), target.to(DEV).contiguous(), gamma=0.97,
                     n_steps=n, entropy_coef=0.01, value_loss_coef=0.5, centralised_critic=True)
    assert up.n_critic == dp.nparams(P * D, H, 1)
    got = up.a2c_loss_grad(dev_ac_batch(batch)).cpu().numpy()
    ref = [m["loss"].item(), m["actor_loss"].item(), m["value_loss"].item(), m["entropy"].item()]
    np.testing.assert_allclose(got[:4], ref, rtol=5e-5, atol=5e-6)
    assert_grad_close(up.actor_grad.cpu().numpy(), a.grad.numpy(), 3e-4)
    assert_grad_close(up.critic_grad.cpu().numpy(), c.grad.numpy(), 3e-4)
    # get_value's rows: every critic on the concatenated observations
    rows = (T + 1) * N
    v = h.ac_forward_rows(spec, up.target_critic, batch["obss"].to(DEV), D, P * D, rows, value_net=2).cpu()
    want = ap.values(target, batch["obss"], D, H).reshape(rows, P).T
    np.testing.assert_allclose(v.reshape(P, rows).numpy(), want.detach().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("P,T,N,D,H,A,sharing", [(8, 10, 100, 39, 128, 6, (0,) * 8), (4, 9, 150, 71, 64, 5, (0, 0, 1, 1))])
def test_wide_centralised_critics_with_shared_networks_vs_oracle_port(P, T, N, D, H, A, sharing):
    """parameter_sharing True / a SePS index list next to critic.centralised (ac/model.py:62-66 builds the shared MultiAgentFCNetwork over the
    concatenated observations): csrc/wide_critic.h reads network net_of[p] for agent p and the per-agent gradients of a network are added in
    agent order - against the port run on the EXPANDED per-agent parameters (autograd sums the shared blocks)"""
    h = hip()
    nb = max(sharing) + 1
    actor = dp.init_params(nb, D, H, A, seed=3) + 0.03
    critic = torch.stack([dp.init_params(1, P * D, H, 1, seed=60 + k)[0] for k in range(nb)]) + 0.02
    target = torch.stack([dp.init_params(1, P * D, H, 1, seed=80 + k)[0] for k in range(nb)])
    batch = ap.synthetic_batch(P, T, N, D, A, seed=9)
    a, c = actor.clone().requires_grad_(True), critic.clone().requires_grad_(True)
    idx = torch.tensor(sharing)
    loss, m = ap.a2c_loss(a[idx], c[idx], target[idx], batch, D, H, A, n_steps=5, gamma=0.97, entropy_coef=0.01, value_loss_coef=0.5)
    loss.backward()
    spec = h.NetSpec(P, D, H, A, sharing=tuple(sharing))
    up = h.AcUpdater(spec, torch.cat([actor.reshape(-1), critic.reshape(-1)]).to(DEV), target.to(DEV).contiguous(), gamma=0.97,
                     n_steps=5, entropy_coef=0.01, value_loss_coef=0.5, centralised_critic=True)
    got = up.a2c_loss_grad(dev_ac_batch(batch)).cpu().numpy()
    ref = [m["loss"].item(), m["actor_loss"].item(), m["value_loss"].item(), m["entropy"].item()]
    np.testing.assert_allclose(got[:4], ref, rtol=5e-5, atol=5e-6)
    assert_grad_close(up.actor_grad.cpu().numpy(), a.grad.numpy(), 3e-4)
    assert_grad_close(up.critic_grad.cpu().numpy(), c.grad.numpy(), 3e-4)


# ---- round 5: actor.parameter_sharing != critic.parameter_sharing (ac/model.py:45-97: two agent -> network maps) -------------------------
@pytest.mark.parametrize("name,cls_name", [("learner_a2c_mixed_sharing_H64.npz", "A2CNetwork"), ("learner_ppo_mixed_sharing_p3_H64.npz", "PPONetwork")])
def test_actor_and_critic_with_their_own_parameter_sharing_match_reference_golden(name, cls_name):
    """one shared actor next to independent critics (A2C), SePS actors [0, 0, 1] next to ONE shared centralised critic (PPO): the reference's
    state_dict keys (`actor.networks.*` beside `critic.independent.*` and the other way round), metrics and blocks of 3 x update() with
    clipping over actor AND critic and the step-keyed target copy"""
    from codebase_amd.ac import model as acm
    from codebase_amd.spaces import Box, Discrete, Tuple

    g = load(name)
    P, D, H, A = int(g["P"]), int(g["D"]), int(g["H"]), int(g["A"])
    a_sh = True if (int(g["actor_is_shared"]) and len(set(g["actor_sharing"].tolist())) == 1) else ([int(i) for i in g["actor_sharing"]] if int(g["actor_is_shared"]) else False)
    c_sh = True if (int(g["critic_is_shared"]) and len(set(g["critic_sharing"].tolist())) == 1) else ([int(i) for i in g["critic_sharing"]] if int(g["critic_is_shared"]) else False)
    cfg = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=float(g["grad_clip"]), n_steps=5, entropy_coef=0.001, value_loss_coef=0.5,
               standardise_returns=False, target_update_interval_or_tau=200, num_epochs=4, ppo_clip=0.2)
    base = dict(layers=[H, H], use_orthogonal_init=True, use_rnn=False)
    net = getattr(acm, cls_name)(Tuple([Box(-1, 8, (D,)) for _ in range(P)]), Tuple([Discrete(A) for _ in range(P)]), cfg,
                                dict(base, parameter_sharing=a_sh), dict(base, parameter_sharing=c_sh, centralised=bool(int(g["centralised"]))), "cuda")
    assert list(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    assert net.actor_params.shape[0] == g["actor0"].shape[0] and net.critic_params.shape[0] == g["critic0"].shape[0]

    # the reference's flat [K][n] blocks <-> the live tensors (views into blocks that may be zero-padded to a wider kernel: the 3-agent
    # centralised critic runs at width 128), family by family in parameters() order
    def family(prefix):
        return [v for k, v in net._views().items() if k.startswith(prefix + ".")]

    def put(prefix, flat):
        o, flat = 0, torch.tensor(flat).reshape(-1)
        for v in family(prefix):
            v.copy_(flat[o:o + v.numel()].reshape(v.shape))
            o += v.numel()
        assert o == flat.numel(), (prefix, o, flat.numel())

    def get(prefix):
        return torch.cat([v.detach().reshape(-1) for v in family(prefix)]).cpu().numpy()

    put("actor", g["actor0"])
    put("critic", g["critic0"])
    put("target_critic", g["target0"])
    # A2CNetwork.get_value through the critics' own map
    b0 = golden_ac_batch(g, 0)
    for i in range(3):
        b = dev_ac_batch(golden_ac_batch(g, i))
        m = net.update(b, int(g["steps"][i]))
        np.testing.assert_allclose([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]], g["metrics"][i], rtol=5e-5, atol=5e-6)
        np.testing.assert_allclose(get("actor"), g[f"actor{i + 1}"].reshape(-1), rtol=0, atol=5e-6)
        np.testing.assert_allclose(get("critic"), g[f"critic{i + 1}"].reshape(-1), rtol=0, atol=5e-6)
        np.testing.assert_allclose(get("target_critic"), g[f"target{i + 1}"].reshape(-1), rtol=0, atol=5e-6)
    obs = [b0["obss"][0, :, p * D:(p + 1) * D].to(DEV) for p in range(P)]
    v, _ = net.get_value(obs, net.init_critic_hiddens(obs[0].shape[0]))
    assert tuple(v.shape) == (obs[0].shape[0], P)


# ---- round 5: actor.layers and critic.layers of different LENGTHS (ac/model.py:45-97: each family from its own list) ---------------------
@pytest.mark.parametrize("name,cls_name", [("learner_a2c_depths.npz", "A2CNetwork"), ("learner_ppo_depths_p3.npz", "PPONetwork")])
def test_actor_and_critic_of_different_depths_match_reference_golden(name, cls_name):
    """two-layer actors next to three-layer critics (A2C: [64, 64] | [48, 64, 32]) and the other way round (PPO, centralised, one shared
    critic: [32, 48, 40] | [64, 64]): both families on the GEMM path, the critics with their own layer count
    (marlhip_ac_config.critic_n_hidden) - state_dict keys and shapes, metrics and every block of 3 x update() against the reference's"""
    from codebase_amd.ac import model as acm
    from codebase_amd.spaces import Box, Discrete, Tuple

    g = load(name)
    P, D, A = int(g["P"]), int(g["D"]), int(g["A"])
    la, lc = [int(h) for h in g["actor_layers"]], [int(h) for h in g["critic_layers"]]
    assert len(la) != len(lc)
    c_sh = bool(int(g["critic_is_shared"]))
    cfg = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=float(g["grad_clip"]), n_steps=5, entropy_coef=0.001, value_loss_coef=0.5,
               standardise_returns=False, target_update_interval_or_tau=200, num_epochs=4, ppo_clip=0.2)
    base = dict(use_orthogonal_init=True, use_rnn=False)
    net = getattr(acm, cls_name)(Tuple([Box(-1, 8, (D,)) for _ in range(P)]), Tuple([Discrete(A) for _ in range(P)]), cfg,
                                dict(base, layers=la, parameter_sharing=False), dict(base, layers=lc, parameter_sharing=c_sh, centralised=bool(int(g["centralised"]))), "cuda")
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert net.spec.wide and net.spec.n_hidden == len(la) and net.critic_spec.n_hidden == len(lc) and net.updater.cfg.critic_n_hidden == len(lc)
    cin = P * D if int(g["centralised"]) else D
    grp = "networks" if c_sh else "independent"
    assert tuple(sd[f"critic.{grp}.0.network.0.weight"].shape) == (lc[0], cin) and tuple(sd[f"critic.{grp}.0.network.{2 * len(lc)}.weight"].shape) == (1, lc[-1])
    assert tuple(sd["actor.independent.0.network.2.weight"].shape) == (la[1], la[0])

    def family(prefix):
        return [v for k, v in net._views().items() if k.startswith(prefix + ".")]

    def put(prefix, flat):
        o, flat = 0, torch.tensor(flat).reshape(-1)
        for v in family(prefix):
            v.copy_(flat[o:o + v.numel()].reshape(v.shape))
            o += v.numel()
        assert o == flat.numel(), (prefix, o, flat.numel())

    def get(prefix):
        return torch.cat([v.detach().reshape(-1) for v in family(prefix)]).cpu().numpy()

    put("actor", g["actor0"])
    put("critic", g["critic0"])
    put("target_critic", g["target0"])
    for i in range(3):
        b = dev_ac_batch(golden_ac_batch(g, i))
        m = net.update(b, int(g["steps"][i]))
        np.testing.assert_allclose([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]], g["metrics"][i], rtol=5e-5, atol=5e-6)
        np.testing.assert_allclose(get("actor"), g[f"actor{i + 1}"].reshape(-1), rtol=0, atol=5e-6)
        np.testing.assert_allclose(get("critic"), g[f"critic{i + 1}"].reshape(-1), rtol=0, atol=5e-6)
        np.testing.assert_allclose(get("target_critic"), g[f"target{i + 1}"].reshape(-1), rtol=0, atol=5e-6)
    b0 = golden_ac_batch(g, 0)
    obs = [b0["obss"][0, :, p * D:(p + 1) * D].to(DEV) for p in range(P)]
    v, _ = net.get_value(obs, net.init_critic_hiddens(obs[0].shape[0]))
    assert tuple(v.shape) == (obs[0].shape[0], P) and bool(torch.isfinite(v).all())
