"""The OPT-IN split-fp16 learner (csrc/dqn_update_h16.h; marlhip_dqn_loss_grad_split16 / marlhip_idqn_update_n_split16) has to clear
the SAME gate as the exact-f32 default: the reference's own goldens (tests/golden/learner_H64.npz: QNetwork._compute_loss, its
gradient, three QNetwork.update steps; marlbase/dqn/model.py:118-196) and the torch-CPU port at the bench size, at the default
tolerances - loss 1e-5 relative (2e-5 for the update sequence, as there), gradient rtol 1e-4, parameters after three Adam steps
3e-6 absolute.  It is never selected by default; this file is what allows bench.py to report it as a second row."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dqn_port as dp
from tests.test_gpu_bench_path_vs_oracle import _perturbed, run_case

G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def dev_batch(h, b):
    return h.Batch(b["obss"].to(DEV).contiguous(), b["actions"].to(DEV).contiguous(), b["rewards"].to(DEV).contiguous(),
                   b["dones"].to(DEV).contiguous(), b["filled"].to(DEV).contiguous(), None)


def golden_batch(g, i):
    return {k: torch.tensor(g[f"batch{i}_{k}"]) for k in ("obss", "actions", "rewards", "dones", "filled")}


def test_loss_and_gradient_match_the_reference_golden():
    from codebase_amd import hip as h

    g = np.load(os.path.join(G, "learner_H64.npz"))
    P, D, H, A = int(g["P"]), int(g["D"]), 64, int(g["A"])
    up = h.DqnUpdater(h.NetSpec(P, D, H, A), torch.tensor(g["params0"], device=DEV), torch.tensor(g["target0"], device=DEV), split16=True)
    loss, grad = up.loss_grad(dev_batch(h, golden_batch(g, 0)))
    loss = loss.cpu().numpy()
    assert abs(loss[0] - g["loss0"]) <= 1e-5 * abs(g["loss0"]), (loss, g["loss0"])
    assert loss[1] == g["batch0_filled"].sum()
    np.testing.assert_allclose(grad.cpu().numpy(), g["grad0"], rtol=1e-4, atol=2e-5)
    g1 = grad.clone()
    _, g2 = up.loss_grad(dev_batch(h, golden_batch(g, 0)))
    assert torch.equal(g1, g2)  # fixed summation order: bitwise reproducible like the default


def test_update_sequence_matches_the_reference_golden():
    """3 x QNetwork.update (clip 1.0, Adam 3e-4, hard target update at update 2) - the tolerances of test_gpu_parity's f32 test"""
    from codebase_amd import hip as h

    g = np.load(os.path.join(G, "learner_H64.npz"))
    P, D, H, A = int(g["P"]), int(g["D"]), 64, int(g["A"])
    params, target = torch.tensor(g["params0"], device=DEV), torch.tensor(g["target0"], device=DEV)
    up = h.DqnUpdater(h.NetSpec(P, D, H, A), params, target, lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, split16=True)
    last = 0
    for i in range(3):
        loss, _ = up.loss_grad(dev_batch(h, golden_batch(g, i)))
        hard = (i + 1 - last) >= 2
        up.apply(hard_update=hard)
        if hard:
            last = i + 1
        assert abs(loss.cpu().numpy()[0] - g["losses"][i]) <= 2e-5 * abs(g["losses"][i])
        if i == 0:
            assert abs(up.gnorm.item() - g["gnorm0"]) <= 1e-4 * g["gnorm0"]
        np.testing.assert_allclose(params.cpu().numpy(), g[f"params{i + 1}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(target.cpu().numpy(), g[f"target{i + 1}"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(up.exp_avg.cpu().numpy(), g["exp_avg3"], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(up.exp_avg_sq.cpu().numpy(), g["exp_avg_sq3"], rtol=1e-3, atol=1e-10)


@pytest.mark.parametrize("P,T,B,D,double_q", [(2, 7, 20, 15, True), (2, 25, 32, 15, False), (3, 5, 16, 18, True), (4, 25, 100, 27, True),
                                               (2, 25, 4096, 15, True), (2, 9, 33, 31, True)])
def test_loss_and_gradient_vs_the_port(P, T, B, D, double_q):
    """ragged batches, other agent counts / observation widths (incl. two dW1 column tiles), max-Q targets, the bench batch size"""
    from codebase_amd import hip as h

    H, A = 64, 6
    params = dp.init_params(P, D, H, A, seed=1) + 0.05 * torch.randn(P, dp.nparams(D, H, A), generator=torch.Generator().manual_seed(2))
    target = dp.init_params(P, D, H, A, seed=3) + 0.05 * torch.randn(P, dp.nparams(D, H, A), generator=torch.Generator().manual_seed(4))
    batch = dp.synthetic_batch(P, T, B, D, A, seed=5)
    pr = params.clone().requires_grad_(True)
    ref = dp.compute_loss(pr, target, batch, 0.99, double_q, D, H, A)
    ref.backward()
    up = h.DqnUpdater(h.NetSpec(P, D, H, A), params.to(DEV), target.to(DEV), double_q=double_q, split16=True)
    loss, grad = up.loss_grad(dev_batch(h, batch))
    assert abs(loss.cpu().numpy()[0] - ref.item()) <= 2e-5 * abs(ref.item())
    gref = pr.grad.numpy()
    np.testing.assert_allclose(grad.cpu().numpy(), gref, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(gref).max()))
    # and how far it is from the exact-f32 kernel on the same inputs (reported, loosely bounded: both sit within roundoff of the port)
    up32 = h.DqnUpdater(h.NetSpec(P, D, H, A), params.to(DEV), target.to(DEV), double_q=double_q)
    l32, g32 = up32.loss_grad(dev_batch(h, batch))
    assert abs(float(l32[0]) - float(loss[0])) <= 2e-5 * abs(ref.item())
    assert float((g32 - grad).abs().max()) <= 4e-5 * max(1.0, np.abs(gref).max())


def test_bench_path_n_updates_from_the_replay_vs_the_port():
    """marlhip_idqn_update_n_split16 (what `bench.py --split16` times): Philox index draw -> in-kernel gather -> split-fp16 loss / gradient
    -> reduce -> clip + Adam + Polyak target, at the bench line's B = 4096 / lr 3e-3 / tau 0.1 and at the golden's B = 32 with hard copies"""
    P, D, H, A, T = 2, 15, 64, 6, 25
    run_case(0, P, D, H, A, T, B=4096, cap=8192, lr=3e-3, tui=0.1, n_calls=2, per_call=(1, 3), params0=_perturbed(P, D, H, A, 1),
             target0=_perturbed(P, D, H, A, 3), atol=3e-5, split16=True)
    g = np.load(os.path.join(G, "learner_H64.npz"))
    run_case(0, P, D, H, A, T, B=32, cap=96, lr=3e-4, tui=2, n_calls=2, per_call=(3, 2), params0=torch.tensor(g["params0"]),
             target0=torch.tensor(g["target0"]), atol=3e-6, split16=True)


def test_shapes_outside_the_experiment_raise():
    from codebase_amd import hip as h

    for spec, kw in ((h.NetSpec(2, 15, 128, 6), {}), (h.NetSpec(8, 39, 64, 6), {}), (h.NetSpec(2, 15, 64, 6), dict(standardise_returns=True))):
        n = spec.nparams()
        with pytest.raises(NotImplementedError):
            h.DqnUpdater(spec, torch.zeros(spec.n_blocks, n, device=DEV), torch.zeros(spec.n_blocks, n, device=DEV), split16=True, **kw)
