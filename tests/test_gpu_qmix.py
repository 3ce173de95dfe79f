"""GPU parity of the QMIX learner step (marlhip_qmix_loss_grad*, csrc/qmix.h) through the C-ABI:
against the reference's own QMixNetwork (tests/golden/learner_qmix_*.npz) and against the oracle port
(oracle/qmix_port.py) on other shapes.  fp32; tolerance as north_star states for the loss (1e-5 relative),
gradients to 1e-4 relative of their largest entry (different f32 summation order than torch's GEMMs)."""
import numpy as np
import pytest
import torch

from oracle import dqn_port as dp
from oracle import qmix_port as qp
from codebase_amd.hip import Batch
from tests.test_gpu_parity import DEV, dev_batch, golden_batch, hip, load

pytestmark = pytest.mark.gpu


def assert_grad_close(got, ref, rel=1e-4):
    np.testing.assert_allclose(got, ref, rtol=rel, atol=rel * max(1e-3, float(np.abs(ref).max())))


def make_updater(h, g, H=64, **kw):
    P, D, A = int(g["P"]), int(g["D"]), int(g["A"])
    spec = h.NetSpec(P, D, H, A)
    t = lambda k: torch.tensor(g[k], device=DEV)  # noqa: E731
    return h.QmixUpdater(spec, t("params0"), t("target0"), t("mixer0"), t("tmixer0"), lr=3e-4, gamma=0.99, grad_clip=1.0,
                         double_q=True, **kw)


@pytest.mark.parametrize("name", ["learner_qmix_H64.npz", "learner_qmix_p4_H64.npz"])
def test_qmix_loss_and_grads_match_reference_golden(name):
    h = hip()
    g = load(name)
    P, D, T, B = int(g["P"]), int(g["D"]), int(g["T"]), int(g["B"])
    up = make_updater(h, g)
    b0 = golden_batch(g, 0)
    loss, grad = up.loss_grad(dev_batch(h, b0))
    assert abs(loss.cpu().numpy()[0] - g["loss0"]) <= 1e-5 * abs(g["loss0"])
    assert loss.cpu().numpy()[1] == b0["filled"].sum().item()
    assert_grad_close(grad.cpu().numpy(), g["grad0"])
    assert_grad_close(up.mixer_grad.cpu().numpy(), g["mgrad0"])
    l1, g1, m1 = loss.clone(), grad.clone(), up.mixer_grad.clone()
    # bitwise the same when the episodes are gathered in-kernel from a replay holding them
    rb = h.DeviceReplay(B, P, D, T)
    rb.obs.copy_(b0["obss"].permute(2, 0, 1, 3))
    rb.act.copy_(b0["actions"].permute(2, 0, 1).to(torch.uint8))
    rb.rew.copy_(b0["rewards"].permute(2, 0, 1))
    rb.done.copy_(b0["dones"].t().to(torch.uint8))
    rb.filled.copy_(b0["filled"].t().to(torch.uint8))
    up.mixer_grad.zero_()
    l2, g2 = up.loss_grad_replay(rb, B, idx=torch.arange(B, dtype=torch.int32, device=DEV))
    assert torch.equal(l1, l2) and torch.equal(g1, g2) and torch.equal(m1, up.mixer_grad)


def test_qmix_update_sequence_matches_reference_golden():
    """3 x QMixNetwork.update: clip over the critic only, one Adam step count for critic and mixer, hard copy of
    target and target mixer at update 2 (dqn/model.py:165-185, 429-443)."""
    h = hip()
    g = load("learner_qmix_H64.npz")
    up = make_updater(h, g)
    last = 0
    for i in range(3):
        loss, _ = up.loss_grad(dev_batch(h, golden_batch(g, i)))
        hard = (i + 1 - last) >= 2
        up.apply(hard_update=hard)
        if hard:
            last = i + 1
        assert abs(loss.cpu().numpy()[0] - g["losses"][i]) <= 2e-5 * abs(g["losses"][i])
        np.testing.assert_allclose(up.params.cpu().numpy(), g[f"params{i + 1}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(up.target.cpu().numpy(), g[f"target{i + 1}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(up.mixer.cpu().numpy(), g[f"mixer{i + 1}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(up.target_mixer.cpu().numpy(), g[f"tmixer{i + 1}"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(up.mixer_exp_avg.cpu().numpy(), g["mixer_exp_avg"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(up.mixer_exp_avg_sq.cpu().numpy(), g["mixer_exp_avg_sq"], rtol=2e-4, atol=1e-9)


@pytest.mark.parametrize("P,T,B,D,H", [(2, 25, 33, 15, 64), (3, 6, 16, 18, 64), (8, 25, 40, 39, 64), (4, 25, 50, 27, 128),
                                       (2, 3, 1, 12, 64), (3, 25, 130, 24, 128), (8, 5, 700, 39, 128),
                                       # more 64-row groups than resident workgroups (256): every workgroup of the fused mixer kernels takes
                                       # several steps - in-kernel weight gradients in two rounds (2 agents) and four (3 agents), the K-chunk
                                       # prefetch carried from one step to the next (4 agents, warehouse rows)
                                       (2, 25, 1400, 15, 64), (3, 25, 700, 18, 64), (4, 25, 701, 27, 64), (2, 30, 600, 71, 64)])
def test_qmix_other_shapes_vs_oracle_port(P, T, B, D, H):
    """every compiled (agents, obs) pair, ragged batch sizes around the 16/32/128-row tiles, both agent-network paths"""
    h = hip()
    A = 5 if D == 71 else 6
    spec = h.NetSpec(P, D, H, A)
    params = dp.init_params(P, D, H, A, seed=1) + 0.05
    target = dp.init_params(P, D, H, A, seed=3)
    mixer = qp.mixer_init(P, P * D, seed=11)
    tmixer = qp.mixer_init(P, P * D, seed=12)
    batch = dp.synthetic_batch(P, T, B, D, A, seed=5)
    batch["rewards"][1:] = batch["rewards"][0]
    batch["obss"] = batch["obss"] * 0.25
    pr, mr = params.clone().requires_grad_(True), mixer.clone().requires_grad_(True)
    ref = qp.compute_loss(pr, target, mr, tmixer, batch, 0.99, True, D, H, A)
    ref.backward()
    up = h.QmixUpdater(spec, params.to(DEV), target.to(DEV), mixer.to(DEV), tmixer.to(DEV))
    loss, grad = up.loss_grad(dev_batch(h, batch))
    assert abs(loss.cpu().numpy()[0] - ref.item()) <= 3e-5 * abs(ref.item())
    assert_grad_close(grad.cpu().numpy(), pr.grad.numpy(), 3e-4)
    assert_grad_close(up.mixer_grad.cpu().numpy(), mr.grad.numpy(), 3e-4)


def test_qmix_narrow_mixer_matches_reference_golden():
    """mixing = {embed_dim 24, hypernet_layers 2, hypernet_embed 16} (any widths are reference branches: QMixer.__init__, dqn/model.py:283-300)
    through QMixNetwork: the narrow mixer sits zero-padded inside the 64 / 32 kernels' block (exact, tests/test_mixer_padding_cpu.py);
    loss, critic and mixer gradients, then 3 x update() with a hard target copy at update 2 against the reference's own QMixNetwork"""
    from codebase_amd.dqn.model import QMixNetwork
    from codebase_amd.spaces import Box, Discrete, Tuple

    g = load("learner_qmix_e24_h16_H64.npz")
    P, D, A, E, HE = int(g["P"]), int(g["D"]), int(g["A"]), int(g["E"]), int(g["HE"])
    assert (E, HE) == (24, 16)
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False, target_update_interval_or_tau=2)
    net = QMixNetwork(Tuple([Box(-1, 8, (D,)) for _ in range(P)]), Tuple([Discrete(A) for _ in range(P)]), hyper, [64, 64], False, False, True,
                      dict(embed_dim=E, hypernet_layers=2, hypernet_embed=HE), "cuda")
    sd = net.state_dict()
    assert tuple(sd["mixer.hyper_w_1.2.weight"].shape) == (P * E, HE) and tuple(sd["mixer.V.2.weight"].shape) == (1, E)
    net.params.copy_(torch.tensor(g["params0"]))
    net.target_params.copy_(torch.tensor(g["target0"]))
    # the reference's flat blocks -> the live views of the padded blocks
    for block, flat in ((net.mixer_params, g["mixer0"]), (net.target_mixer_params, g["tmixer0"])):
        o = 0
        for view, shape in net._mixer_views(block, "mixer").values():
            n = int(np.prod(shape))
            view.copy_(torch.tensor(flat[o:o + n]).reshape(view.shape))
            o += n
        assert o == flat.size
    h = hip()
    up = net.updater
    loss, grad = up.loss_grad(dev_batch(h, golden_batch(g, 0)))
    assert abs(loss.cpu().numpy()[0] - g["loss0"]) <= 1e-5 * abs(g["loss0"])
    assert_grad_close(grad.cpu().numpy(), g["grad0"])
    mg = net.mixer_flat(up.mixer_grad).cpu().numpy()
    assert_grad_close(mg, g["mgrad0"])
    live = torch.zeros_like(up.mixer_grad)
    for view, _ in net._mixer_views(live, "mixer").values():
        view.fill_(1.0)
    assert float(up.mixer_grad[live == 0].abs().max()) == 0.0   # nothing flows into the padding
    for i in range(3):
        b = golden_batch(g, i)
        lo = net.update(Batch(b["obss"].to(DEV), b["actions"].to(DEV), b["rewards"].to(DEV), b["dones"].to(DEV), b["filled"].to(DEV), None))["loss"]
        assert abs(lo - g["losses"][i]) <= 2e-5 * abs(g["losses"][i])
        np.testing.assert_allclose(net.params.cpu().numpy(), g[f"params{i + 1}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(net.mixer_flat().cpu().numpy(), g[f"mixer{i + 1}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(net.mixer_flat(net.target_mixer_params).cpu().numpy(), g[f"tmixer{i + 1}"], rtol=0, atol=3e-6)
    assert float(net.mixer_params[live == 0].abs().max()) == 0.0       # the padding is still exactly zero after Adam


def test_qmix_library_loop_equals_the_host_loop_bit_for_bit(monkeypatch):
    """marlhip_qmix_update_n (the trainer's default since round 4) issues exactly the launches of the per-update host loop
    (QMixNetwork.update_async x U): two trainers on the same seeds, one forced onto the host loop, must end with the same bytes -
    critic, target, mixer, target mixer, Adam moments, counters - across a hard target copy inside a round"""
    from codebase_amd.dqn import train as T
    from codebase_amd.dqn.model import QMixNetwork
    from codebase_amd.utils.envs import _space_pair

    h = hip()
    out = []
    for host_loop in (False, True):
        monkeypatch.setattr(T, "_NO_FUSED_LOOP", host_loop)
        cfg = h.env_config("lbforaging:Foraging-8x8-2p-3f-v3", 256, 25, seed=9, cooperative=True)
        torch.manual_seed(4)
        obs_space, act_space = _space_pair(cfg)
        hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False, target_update_interval_or_tau=4)
        m = QMixNetwork(obs_space, act_space, hyper, [64, 64], False, False, True, dict(embed_dim=64, hypernet_layers=2, hypernet_embed=32), "cuda")
        tr = T.VectorisedIDQN(cfg, m, 512, 25, 96, 3, seed=2)
        for _ in range(3):
            tr.round(0.4)
        torch.cuda.synchronize()
        assert (tr._fused is not None) == (not host_loop)
        out.append((m.updates, m.last_target_update, m.updater.step, [t.cpu().clone() for t in (m.params, m.target_params, m.mixer_params, m.target_mixer_params,
                                                                                               m.updater.exp_avg_sq, m.updater.mixer_exp_avg, tr.last_loss)]))
    (ua, la, sa, ta), (ub, lb, sb, tb) = out
    assert (ua, la, sa) == (ub, lb, sb) == (9, 8, 9)
    for x, y in zip(ta, tb):
        assert torch.equal(x, y)


def test_qmix_rejects_what_the_reference_rejects():
    """QMixer.__init__ (dqn/model.py:298-301) raises for hypernet_layers outside {1, 2}; a mixer block of the wrong size is refused before any launch"""
    h = hip()
    from codebase_amd._lib import MarlHipError

    spec = h.NetSpec(2, 15, 64, 6)
    z = torch.zeros(10, device=DEV)
    with pytest.raises(MarlHipError):
        h.QmixUpdater(spec, dp.init_params(2, 15, 64, 6).to(DEV), dp.init_params(2, 15, 64, 6).to(DEV), z, z,
                      mixing=dict(embed_dim=32, hypernet_layers=3, hypernet_embed=64))
    with pytest.raises(ValueError):
        h.QmixUpdater(spec, dp.init_params(2, 15, 64, 6).to(DEV), dp.init_params(2, 15, 64, 6).to(DEV), z, z,
                      mixing=dict(embed_dim=32, hypernet_layers=1, hypernet_embed=64))


# ---- round 5: the QMixer configurations outside the fused kernels, on the generic mixer stage (csrc/qmix_gen.hip) ---------------------------
@pytest.mark.parametrize("name", ["learner_qmix_L1_H64.npz", "learner_qmix_L1_e40_p3_H64.npz", "learner_qmix_e96_h48_H64.npz"])
def test_generic_mixer_matches_reference_goldens(name):
    """hypernet_layers = 1 (one Linear per hypernet, dqn/model.py:283-285) and mixers wider than 64 / 32 through QMixNetwork: state_dict keys
    and shapes of the reference's QMixer, loss, critic and mixer gradients, then 3 x update() with a hard copy at update 2 - all against
    the reference's own QMixNetwork"""
    from codebase_amd.dqn.model import QMixNetwork
    from codebase_amd.spaces import Box, Discrete, Tuple

    g = load(name)
    P, D, A, E, HE, L = (int(g[k]) for k in ("P", "D", "A", "E", "HE", "L"))
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False, target_update_interval_or_tau=2)
    net = QMixNetwork(Tuple([Box(-1, 8, (D,)) for _ in range(P)]), Tuple([Discrete(A) for _ in range(P)]), hyper, [64, 64], False, False, True,
                      dict(embed_dim=E, hypernet_layers=L, hypernet_embed=HE), "cuda")
    sd = net.state_dict()
    assert [k for k in sd if k.startswith("mixer.")] == ["mixer." + k for k in g["mixer_keys"]]
    assert net.mixer_params.numel() == g["mixer0"].size == qp.mixer_nparams(P, P * D, E, HE, L)
    net.params.copy_(torch.tensor(g["params0"]))
    net.target_params.copy_(torch.tensor(g["target0"]))
    net.mixer_params.copy_(torch.tensor(g["mixer0"]))
    net.target_mixer_params.copy_(torch.tensor(g["tmixer0"]))
    h = hip()
    up = net.updater
    loss, grad = up.loss_grad(dev_batch(h, golden_batch(g, 0)))
    assert abs(loss.cpu().numpy()[0] - g["loss0"]) <= 1e-5 * abs(g["loss0"])
    assert_grad_close(grad.cpu().numpy(), g["grad0"])
    assert_grad_close(up.mixer_grad.cpu().numpy(), g["mgrad0"])
    for i in range(3):
        b = golden_batch(g, i)
        lo = net.update(Batch(b["obss"].to(DEV), b["actions"].to(DEV), b["rewards"].to(DEV), b["dones"].to(DEV), b["filled"].to(DEV), None))["loss"]
        assert abs(lo - g["losses"][i]) <= 2e-5 * abs(g["losses"][i])
        np.testing.assert_allclose(net.params.cpu().numpy(), g[f"params{i + 1}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(net.mixer_params.cpu().numpy(), g[f"mixer{i + 1}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(net.target_mixer_params.cpu().numpy(), g[f"tmixer{i + 1}"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(up.mixer_exp_avg.cpu().numpy(), g["mixer_exp_avg"], rtol=1e-4, atol=1e-6)


def test_qmix_around_agent_networks_wider_than_the_fused_kernels_matches_reference_golden():
    """QMixNetwork takes any `layers` (dqn/model.py:334-372): layers = [136, 136] puts the agent networks on the GEMM path
    (hip.WideQmixUpdater -> marlhip_wide_qmix_loss_grad) around qmix.yaml's mixer.  Against the reference's own QMixNetwork: state_dict
    shapes, loss, critic and mixer gradients, 2 x update() with a hard copy of target and target mixer at update 2."""
    from codebase_amd import hip as hh
    from codebase_amd.dqn.model import QMixNetwork
    from codebase_amd.spaces import Box, Discrete, Tuple

    g = load("learner_qmix_layers136.npz")
    P, D, A, E, HE, L = (int(g[k]) for k in ("P", "D", "A", "E", "HE", "L"))
    layers = [int(x) for x in g["layers"]]
    assert layers == [136, 136]
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False, target_update_interval_or_tau=2)
    net = QMixNetwork(Tuple([Box(-1, 8, (D,)) for _ in range(P)]), Tuple([Discrete(A) for _ in range(P)]), hyper, layers, False, False, True,
                      dict(embed_dim=E, hypernet_layers=L, hypernet_embed=HE), "cuda")
    assert type(net.updater) is hh.WideQmixUpdater and net.spec.wide
    sd = net.state_dict()
    assert tuple(sd["critic.independent.0.network.0.weight"].shape) == (136, D) and tuple(sd["critic.independent.1.network.2.weight"].shape) == (136, 136)
    # the blocks live zero-padded at the GEMM path's width (136 -> 144: dqn/model.py compiled_width); the golden's are the reference's own
    from codebase_amd.dqn.model import block_views, pad_blocks

    Hk = net.spec.hidden
    assert Hk == 144 and net.params.shape[1] > g["params0"].shape[1]

    def live(block):  # [P][padded] -> [P][n(136, 136)] in parameters() order
        return np.stack([torch.cat([v.reshape(-1) for _, v in block_views(block[i].cpu(), D, layers, A, Hk)]).numpy() for i in range(P)])

    net.params.copy_(pad_blocks(torch.tensor(g["params0"]), D, layers, A, Hk))
    net.target_params.copy_(pad_blocks(torch.tensor(g["target0"]), D, layers, A, Hk))
    net.mixer_params.copy_(torch.tensor(g["mixer0"]))
    net.target_mixer_params.copy_(torch.tensor(g["tmixer0"]))
    h = hip()
    up = net.updater
    loss, grad = up.loss_grad(dev_batch(h, golden_batch(g, 0)))
    assert abs(loss.cpu().numpy()[0] - g["loss0"]) <= 1e-5 * abs(g["loss0"])
    assert_grad_close(live(grad), g["grad0"])
    assert float(grad.abs().sum()) == pytest.approx(float(np.abs(live(grad)).sum()), rel=1e-6)  # nothing lands in the padding
    assert_grad_close(up.mixer_grad.cpu().numpy(), g["mgrad0"])
    for i in range(2):
        b = golden_batch(g, i)
        lo = net.update(Batch(b["obss"].to(DEV), b["actions"].to(DEV), b["rewards"].to(DEV), b["dones"].to(DEV), b["filled"].to(DEV), None))["loss"]
        assert abs(lo - g["losses"][i]) <= 2e-5 * abs(g["losses"][i])
        np.testing.assert_allclose(live(net.params), g[f"params{i + 1}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(live(net.target_params), g[f"target{i + 1}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(net.mixer_params.cpu().numpy(), g[f"mixer{i + 1}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(net.target_mixer_params.cpu().numpy(), g[f"tmixer{i + 1}"], rtol=0, atol=3e-6)


@pytest.mark.parametrize("P,T,B,D,H,E,HE,L,kind", [(8, 6, 70, 39, 128, 64, 32, 1, "fused"), (4, 25, 33, 27, 64, 128, 64, 2, "fused"), (2, 9, 130, 15, 64, 200, 5, 1, "fused"),
                                                   (5, 7, 21, 11, 64, 64, 32, 2, "wide"),   # an (agents, obs) pair with no compiled mixer: the generic stage at qmix.yaml's widths
                                                   (3, 8, 19, 18, 96, 48, 32, 1, "wide")])
def test_generic_mixer_other_shapes_vs_oracle_port(P, T, B, D, H, E, HE, L, kind):
    """8 agents, embeddings wider than a wave, ragged rows; around the fused agent kernels (also gathered in-kernel from the replay) and
    around agent networks on the GEMM path"""
    h = hip()
    A = 6
    hid = H if kind == "fused" else (H, H)
    spec = h.NetSpec(P, D, H, A) if kind == "fused" else h.NetSpec(P, D, H, A, wide=True, n_hidden=2)
    params = dp.init_params(P, D, hid, A, seed=1) + 0.05
    target = dp.init_params(P, D, hid, A, seed=3)
    mixer, tmixer = qp.mixer_init(P, P * D, E, HE, seed=11, L=L), qp.mixer_init(P, P * D, E, HE, seed=12, L=L)
    batch = dp.synthetic_batch(P, T, B, D, A, seed=5)
    batch["rewards"][1:] = batch["rewards"][0]
    batch["obss"] = batch["obss"] * 0.25
    pr, mr = params.clone().requires_grad_(True), mixer.clone().requires_grad_(True)
    ref = qp.compute_loss(pr, target, mr, tmixer, batch, 0.99, True, D, hid, A, E, HE, L)
    ref.backward()
    cls = h.QmixUpdater if kind == "fused" else h.WideQmixUpdater
    up = cls(spec, params.to(DEV), target.to(DEV), mixer.to(DEV), tmixer.to(DEV), mixing=dict(embed_dim=E, hypernet_layers=L, hypernet_embed=HE))
    loss, grad = up.loss_grad(dev_batch(h, batch))
    assert abs(loss.cpu().numpy()[0] - ref.item()) <= 3e-5 * abs(ref.item())
    assert_grad_close(grad.cpu().numpy(), pr.grad.numpy(), 3e-4)
    assert_grad_close(up.mixer_grad.cpu().numpy(), mr.grad.numpy(), 3e-4)
    if kind == "fused":  # the same numbers with the rows gathered from a replay holding the episodes in another order
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))
        rb = h.DeviceReplay(B, P, D, T)
        inv = torch.argsort(perm)
        rb.obs.copy_(batch["obss"].permute(2, 0, 1, 3)[inv])
        rb.act.copy_(batch["actions"].permute(2, 0, 1).to(torch.uint8)[inv])
        rb.rew.copy_(batch["rewards"].permute(2, 0, 1)[inv])
        rb.done.copy_(batch["dones"].t().to(torch.uint8)[inv])
        rb.filled.copy_(batch["filled"].t().to(torch.uint8)[inv])
        l1, g1, m1 = loss.clone(), grad.clone(), up.mixer_grad.clone()
        up.mixer_grad.zero_()
        l2, g2 = up.loss_grad_replay(rb, B, idx=perm.to(torch.int32).to(DEV))
        assert torch.equal(l1, l2) and torch.equal(g1, g2) and torch.equal(m1, up.mixer_grad)


def test_generic_mixer_reproduces_the_compiled_configurations_goldens():
    """MARLHIP_QMIX_GENERIC=1 sends mixing = {64, 2, 32} on the compiled shapes through the generic stage too: the reference's goldens and the
    port comparisons of this file must hold for it unchanged (a fresh interpreter: the switch is read once per process)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_qmix.py"), "-q", "-x", "-m", "gpu", "-k",
                          "loss_and_grads_match_reference_golden or update_sequence_matches_reference_golden or narrow_mixer or (other_shapes_vs_oracle_port and not generic)"],
                         cwd=root, env=dict(os.environ, MARLHIP_QMIX_GENERIC="1"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("P,T,B,D,H,L", [(2, 25, 33, 15, 256, 2), (4, 9, 50, 27, 160, 2), (2, 12, 40, 15, 64, 3), (3, 7, 21, 18, 96, 1)])
def test_qmix_with_agent_networks_on_the_gemm_path_vs_oracle_port(P, T, B, D, H, L):
    """layers wider than 128 or not two deep: marlhip_wide_qmix_loss_grad (GEMM agent networks around the same mixer stage)"""
    h = hip()
    A = 6
    hid = (H,) * L
    spec = h.NetSpec(P, D, H, A, wide=True, n_hidden=L)
    params = dp.init_params(P, D, hid, A, seed=1) + 0.05
    target = dp.init_params(P, D, hid, A, seed=3)
    mixer = qp.mixer_init(P, P * D, seed=11)
    tmixer = qp.mixer_init(P, P * D, seed=12)
    batch = dp.synthetic_batch(P, T, B, D, A, seed=5)
    batch["rewards"][1:] = batch["rewards"][0]
    batch["obss"] = batch["obss"] * 0.25
    pr, mr = params.clone().requires_grad_(True), mixer.clone().requires_grad_(True)
    ref = qp.compute_loss(pr, target, mr, tmixer, batch, 0.99, True, D, hid, A)
    ref.backward()
    up = h.WideQmixUpdater(spec, params.to(DEV), target.to(DEV), mixer.to(DEV), tmixer.to(DEV))
    loss, grad = up.loss_grad(dev_batch(h, batch))
    assert abs(loss.cpu().numpy()[0] - ref.item()) <= 3e-5 * abs(ref.item())
    assert_grad_close(grad.cpu().numpy(), pr.grad.numpy(), 3e-4)
    assert_grad_close(up.mixer_grad.cpu().numpy(), mr.grad.numpy(), 3e-4)


@pytest.mark.parametrize("P,T,B,D,H", [(2, 25, 33, 15, 64), (8, 9, 40, 39, 128), (4, 25, 50, 27, 64)])
def test_qmix_opt_in_fp16_first_mixer_layers_stay_close_to_fp32(P, T, B, D, H):
    """marlhip_qmix_mixer.l1_fp16 (BASELINE config 5's "fp16 mixer on MFMA", a deviation from the fp32 reference): weights of the mixers'
    state-fed first layers rounded to fp16, states exact, fp32 accumulation - loss and gradients move by fp16 rounding, no more"""
    h = hip()
    A = 6
    spec = h.NetSpec(P, D, H, A)
    params = dp.init_params(P, D, H, A, seed=1) + 0.05
    target = dp.init_params(P, D, H, A, seed=3)
    mixer = qp.mixer_init(P, P * D, seed=11)
    tmixer = qp.mixer_init(P, P * D, seed=12)
    batch = dp.synthetic_batch(P, T, B, D, A, seed=5)
    batch["rewards"][1:] = batch["rewards"][0]
    out = []
    for fp16 in (False, True):
        up = h.QmixUpdater(spec, params.clone().to(DEV), target.clone().to(DEV), mixer.clone().to(DEV), tmixer.clone().to(DEV),
                           mixing=dict(embed_dim=64, hypernet_layers=2, hypernet_embed=32, fp16=fp16))
        loss, grad = up.loss_grad(dev_batch(h, batch))
        out.append((float(loss[0]), grad.cpu().numpy().copy(), up.mixer_grad.cpu().numpy().copy()))
    (l32, g32, m32), (l16, g16, m16) = out
    assert l16 != l32 and abs(l16 - l32) <= 5e-3 * abs(l32), (l32, l16)
    assert np.abs(g16 - g32).max() <= 2e-2 * np.abs(g32).max()
    assert np.abs(m16 - m32).max() <= 2e-2 * np.abs(m32).max()
