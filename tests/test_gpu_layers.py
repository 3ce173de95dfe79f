"""`algorithm.model.layers` other than [64, 64] / [128, 128]: any two hidden widths up to 128 run on the compiled kernels by zero
padding (codebase_amd/dqn/model.py pad_blocks).  The oracle here is the port at the TRUE widths (FCNetwork([h1, h2]),
utils/models.py:34-48): same initial tensors, same batches -> same losses and parameters, state_dict in the reference's shapes,
padding exactly zero after the updates.
"""
import numpy as np
import pytest
import torch

from oracle import ac_update_port as ap
from oracle import dqn_port as dp

pytestmark = pytest.mark.gpu
DEV = "cuda"


def spaces(P, D, A):
    from codebase_amd.spaces import Box, Discrete, Tuple
    return Tuple([Box(-1, 8, (D,)) for _ in range(P)]), Tuple([Discrete(A) for _ in range(P)])


def live_blocks(sd, prefix, P):
    """state_dict -> [P][n(h1, h2)] flat blocks in parameters() order"""
    return torch.stack([torch.cat([v.reshape(-1).cpu() for k, v in sd.items() if k.startswith(f"{prefix}.independent.{p}.")]) for p in range(P)])


@pytest.mark.parametrize("cls_name,mode", [("QNetwork", "idqn"), ("VDNetwork", "vdn")])
@pytest.mark.parametrize("layers", [[32, 32], [48, 24], [96, 128], [128, 40], [256, 256], [200, 96], [64], [64, 64, 64], [100, 50, 30, 20], [48, 40, 32, 24, 16, 8]])  # > 128 / not two layers (any depth up to 16): the GEMM path
def test_dqn_family_with_other_widths_matches_the_port_at_the_true_widths(cls_name, mode, layers):
    from codebase_amd import hip as h
    from codebase_amd.dqn import model as M

    P, D, A, T, B = 2, 15, 6, 25, 37
    obs_space, act_space = spaces(P, D, A)
    hyper = dict(optimizer="Adam", lr=1e-3, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False,
                 target_update_interval_or_tau=2)
    torch.manual_seed(5)
    net = getattr(M, cls_name)(obs_space, act_space, hyper, layers, False, False, True, DEV)
    hid = tuple(layers)
    sd = net.state_dict()
    dims = [D] + list(layers) + [A]
    for k in range(len(dims) - 1):  # the reference's keys and shapes: network.{0, 2, 4, ...}.{weight, bias}
        assert sd[f"critic.independent.0.network.{2 * k}.weight"].shape == (dims[k + 1], dims[k])
        assert sd[f"target.independent.1.network.{2 * k}.bias"].shape == (dims[k + 1],)
    assert len(sd) == 2 * 2 * 2 * (len(dims) - 1)
    # the reference's RNG order at the true shapes: a second FCNetwork-style draw from the same seed gives the same tensors
    torch.manual_seed(5)
    want, _ = M.init_flat_params([D] * P, layers, [A] * P, True, None)
    assert torch.equal(live_blocks(sd, "critic", P), want)
    start = live_blocks(sd, "critic", P) + 0.01  # biases off zero so every tensor takes part
    net.load_state_dict({k: v + 0.01 for k, v in sd.items()})
    ref = dp.Learner(start, D, hid, A, lr=1e-3, gamma=0.99, grad_clip=1.0, double_q=True, target_update_interval_or_tau=2, mode=mode)
    ref.target = start.clone()
    for i in range(4):
        b = dp.synthetic_batch(P, T, B, D, A, seed=30 + i)
        got = net.update(h.Batch(b["obss"], b["actions"], b["rewards"], b["dones"], b["filled"], None))
        exp = ref.update(b)
        assert abs(got["loss"] - exp["loss"]) <= 3e-5 * abs(exp["loss"]), (i, got["loss"], exp["loss"])
    sd = net.state_dict()
    np.testing.assert_allclose(live_blocks(sd, "critic", P).numpy(), ref.flat().detach().numpy(), rtol=0, atol=5e-6)
    np.testing.assert_allclose(live_blocks(sd, "target", P).numpy(), ref.target.numpy(), rtol=0, atol=5e-6)
    # the padding never moves: the number of non-zero parameters is at most the live count
    live = dp.nparams(D, hid, A)
    assert int((net.params != 0).sum(dim=1).max()) <= live and int((net.target_params != 0).sum(dim=1).max()) <= live
    # act(): greedy actions of the live sub-network
    obs = [np.random.default_rng(p).integers(-1, 8, D).astype(np.float32) for p in range(P)]
    acts, _ = net.act(obs, net.init_hiddens(1), 0.0)
    q = [dp.mlp(ref.flat().detach()[p], torch.tensor(obs[p]), D, hid, A) for p in range(P)]
    for p in range(P):
        top = torch.sort(q[p]).values
        if top[-1] - top[-2] > 1e-4:
            assert acts[p] == int(q[p].argmax())


@pytest.mark.parametrize("layers", [[200, 96], [256, 256], [64, 64, 64]])
def test_idqn_standardise_returns_on_the_gemm_path_matches_the_port(layers):
    """standardise_returns with a layer list that has no fused kernel (csrc/wide.hip: marlhip_wide_dqn_loss_grad_std) against the port at
    the true widths: losses, parameters and the per-agent running (mean, var, count) of three updates"""
    from codebase_amd import hip as h
    from codebase_amd.dqn import model as M

    P, D, A, T, B = 2, 15, 6, 25, 37
    obs_space, act_space = spaces(P, D, A)
    hyper = dict(optimizer="Adam", lr=1e-3, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=True,
                 target_update_interval_or_tau=2)
    torch.manual_seed(6)
    net = M.QNetwork(obs_space, act_space, hyper, layers, False, False, True, DEV)
    assert net.spec.wide
    sd = net.state_dict()
    net.load_state_dict({k: v + 0.01 for k, v in sd.items()})
    start = live_blocks(net.state_dict(), "critic", P)
    ref = dp.Learner(start, D, tuple(layers), A, lr=1e-3, gamma=0.99, grad_clip=1.0, double_q=True, target_update_interval_or_tau=2,
                     standardise_returns=True)
    ref.target = live_blocks(net.state_dict(), "target", P).clone()
    for i in range(3):
        b = dp.synthetic_batch(P, T, B, D, A, seed=40 + i)
        got = net.update(h.Batch(b["obss"], b["actions"], b["rewards"], b["dones"], b["filled"], None))
        exp = ref.update(b)
        assert abs(got["loss"] - exp["loss"]) <= 5e-5 * abs(exp["loss"]), (i, got["loss"], exp["loss"])
        st = net.ret_ms
        np.testing.assert_allclose(st.mean.cpu().numpy(), ref.ret_ms.mean.numpy(), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(st.var.cpu().numpy(), ref.ret_ms.var.numpy(), rtol=5e-5)
        assert abs(st.count - ref.ret_ms.count) < 1e-6
    np.testing.assert_allclose(live_blocks(net.state_dict(), "critic", P).numpy(), ref.flat().detach().numpy(), rtol=0, atol=1e-5)


def test_layer_lists_the_kernels_do_not_cover_raise():
    from codebase_amd.dqn.model import QNetwork
    obs_space, act_space = spaces(2, 15, 6)
    hyper = dict(optimizer="Adam", lr=3e-4)
    for layers in ([], [64] * 17, [2048, 2048], [0, 64]):
        with pytest.raises(NotImplementedError):
            QNetwork(obs_space, act_space, hyper, layers, False, False, True, DEV)
    for layers in ([32, 48], [64] * 6, [64], [192, 192]):  # recurrent: equal sizes (RNNNetwork asserts it), one to four stacked GRU layers, h <= 128
        with pytest.raises(NotImplementedError):
            QNetwork(obs_space, act_space, hyper, layers, False, True, True, DEV)


@pytest.mark.parametrize("layers,centralised,P", [([32, 48], False, 2), ([100, 20], False, 3), ([64, 64], True, 4), ([48, 48], True, 3),
                                                  ([256, 256], False, 2), ([160, 200], True, 3), ([96], False, 2), ([64, 48, 32], True, 2)])  # the GEMM path
def test_actor_critic_with_other_widths_matches_the_port_at_the_true_widths(layers, centralised, P):
    """A2CNetwork with layers the kernels are not compiled for; [64, 64] centralised critics for 3 / 4 agents run padded to 128"""
    from codebase_amd.ac.model import A2CNetwork
    from tests.test_gpu_ac_update import dev_ac_batch

    D, A, T, N = {2: 15, 3: 18, 4: 21}[P], 6, 10, 19
    obs_space, act_space = spaces(P, D, A)
    cfg = dict(optimizer="Adam", lr=1e-3, gamma=0.97, grad_clip=0.5, n_steps=5, entropy_coef=0.01, value_loss_coef=0.5,
               standardise_returns=False, target_update_interval_or_tau=2)
    net_cfg = dict(layers=layers, parameter_sharing=False, use_orthogonal_init=True, use_rnn=False)
    torch.manual_seed(9)
    net = A2CNetwork(obs_space, act_space, cfg, net_cfg, dict(net_cfg, centralised=centralised), DEV)
    hid = tuple(layers)
    cin = P * D if centralised else D
    sd = net.state_dict()
    last = 2 * len(hid)
    assert sd["actor.independent.0.network.0.weight"].shape == (hid[0], D) and sd["critic.independent.0.network.0.weight"].shape == (hid[0], cin)
    assert sd[f"critic.independent.1.network.{last}.weight"].shape == (1, hid[-1]) and sd[f"actor.independent.0.network.{last}.bias"].shape == (A,)
    net.load_state_dict({k: v + 0.01 for k, v in sd.items()})
    sd = net.state_dict()
    actor, critic, target = (live_blocks(sd, k, P) for k in ("actor", "critic", "target_critic"))
    ref = ap.Learner(actor, critic, D, hid, A, lr=1e-3, gamma=0.97, n_steps=5, entropy_coef=0.01, value_loss_coef=0.5, grad_clip=0.5,
                     target_update_interval_or_tau=2)
    ref.target = target.clone()
    for i in range(3):
        b = ap.synthetic_batch(P, T, N, D, A, seed=60 + i)
        got = net.update(dev_ac_batch(b), i)
        exp = ref.update(b, i)
        for k in ("loss", "actor_loss", "value_loss", "entropy"):
            assert abs(got[k] - float(exp[k])) <= 5e-5 * abs(float(exp[k])) + 5e-6, (i, k, got[k], float(exp[k]))
    sd = net.state_dict()
    np.testing.assert_allclose(live_blocks(sd, "actor", P).numpy(), ref.actor().detach().numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(live_blocks(sd, "critic", P).numpy(), ref.critic().detach().numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(live_blocks(sd, "target_critic", P).numpy(), ref.target.numpy(), rtol=0, atol=1e-5)
    assert int((net.actor_params != 0).sum(dim=1).max()) <= dp.nparams(D, hid, A)
    assert int((net.critic_params != 0).sum(dim=1).max()) <= dp.nparams(cin, hid, 1)


def test_actor_and_critic_may_differ_in_width():
    from codebase_amd.ac.model import PPONetwork
    obs_space, act_space = spaces(2, 15, 6)
    cfg = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=0.5, n_steps=5, entropy_coef=0.001, value_loss_coef=0.5,
               standardise_returns=False, target_update_interval_or_tau=0.01, num_epochs=4, ppo_clip=0.2)
    a = dict(layers=[64, 32], parameter_sharing=False, use_orthogonal_init=True, use_rnn=False)
    c = dict(layers=[128, 96], parameter_sharing=False, use_orthogonal_init=True, use_rnn=False, centralised=False)
    net = PPONetwork(obs_space, act_space, cfg, a, c, DEV)
    sd = net.state_dict()
    assert sd["actor.independent.0.network.2.weight"].shape == (32, 64) and sd["critic.independent.0.network.2.weight"].shape == (96, 128)
    assert net.spec.hidden == 128
    obs = [torch.rand(5, 15) for _ in range(2)]
    v, _ = net.get_value(obs, None)
    ref = torch.cat([dp.mlp(live_blocks(sd, "critic", 2)[p], obs[p], 15, (128, 96), 1) for p in range(2)], dim=-1)
    np.testing.assert_allclose(v.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


def test_wide_layers_run_end_to_end(tmp_path, monkeypatch):
    """layers [256, 256] through the reference-shaped drivers: modular collection (GEMM forward -> Philox action choice -> env step ->
    replay add), the GEMM learner, evaluation, checkpoints in the reference's shapes"""
    from codebase_amd import run
    NAME = "lbforaging:Foraging-8x8-2p-3f-v3"
    for algo, extra in (("idqn", ["algorithm.model.layers=[256,256]", "algorithm.batch_size=64"]),
                        ("vdn", ["algorithm.model.layers=[192,256]", "algorithm.batch_size=64"]),
                        ("qmix", ["algorithm.model.layers=[64,64,64]", "algorithm.batch_size=64"]),
                        ("ia2c", ["algorithm.model.actor.layers=[256,256]", "algorithm.model.critic.layers=[256,256]"])):
        monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / algo))
        df = run.main([f"+algorithm={algo}", f"env.name={NAME}", "env.time_limit=25", "env.parallel_envs=256", "seed=1",
                       "algorithm.total_steps=60000", "algorithm.eval_interval=20000"] + extra)
        assert df.shape[0] >= 2 and np.isfinite(df["loss"].dropna()).all() and np.isfinite(df["mean_episode_returns"]).all()
