"""marlhip_idqn_update_n's 3-launch update (loss/grad with packs kept current by the previous Adam launch, reduce + clip-norm
partials, clip + Adam + target + next packs) against the generic 4-launch loop (pack, loss/grad, reduce, clip + Adam): the two
differ only in the summation order of the clip norm, so parameters, Adam moments and targets agree to a few ulp after a run that
crosses a hard target update; `adam_pack_kernel`'s scatter into the MFMA packs is exact or the second update already diverges."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _fill_replay(h, rb, P, D, T, A, n, seed):
    g = torch.Generator().manual_seed(seed)
    rb.obs.copy_(torch.randint(-1, 8, rb.obs.shape, generator=g).float())
    rb.act.copy_(torch.randint(0, A, rb.act.shape, generator=g).to(torch.uint8))
    rb.rew.copy_(torch.rand(rb.rew.shape, generator=g))
    ln = torch.randint(3, T + 1, (rb.done.shape[0],), generator=g)
    t = torch.arange(T + 1)[None, :]
    rb.done.copy_((t == ln[:, None]).to(torch.uint8))
    rb.filled.copy_((t[:, :T] < ln[:, None]).to(torch.uint8))


def _run(h, spec, mode, fused, tui, n_updates, B=96, cap=300, T=25, seed=11):
    from oracle import dqn_port as dp

    P, D, H, A = spec.n_agents, spec.obs_dim, spec.hidden, spec.n_actions
    nb = spec.n_blocks
    p0 = dp.init_params(nb, D, H, A, seed=5)
    params, target = p0.to(DEV), dp.init_params(nb, D, H, A, seed=6).to(DEV)
    rb = h.DeviceReplay(cap, P, D, T)
    _fill_replay(h, rb, P, D, T, A, cap, seed)
    up = h.DqnUpdater(spec, params, target, grad_clip=0.5)  # small max_norm: the clip coefficient matters in every update
    fl = h.FusedLearner(up, rb, B, tui, mode=mode)
    if fused:
        os.environ.pop("MARLHIP_NO_FUSED_EPILOGUE", None)
    else:
        os.environ["MARLHIP_NO_FUSED_EPILOGUE"] = "1"
    try:
        upd, last = 0, 0
        for call in range(2):  # two calls: the packs are rebuilt at the start of each
            upd, last = fl.run(n_updates, cap, seed, 100 * call, upd, last)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("MARLHIP_NO_FUSED_EPILOGUE", None)
    return dict(params=params.cpu(), target=target.cpu(), m=up.exp_avg.cpu(), v=up.exp_avg_sq.cpu(), loss=up.loss.cpu(),
                gnorm=up.gnorm.cpu(), upd=upd, last=last, step=up.step)


@pytest.mark.parametrize("mode,D,sharing,tui", [(0, 15, None, 3), (1, 15, None, 0.25), (0, 27, None, 200), (0, 17, [0, 0], 3)])
def test_fused_epilogue_matches_generic_loop(mode, D, sharing, tui):
    from codebase_amd import hip as h

    P = 4 if D == 27 else 2
    spec = h.NetSpec(P, D, 64, 6, None if sharing is None else tuple(sharing))
    a = _run(h, spec, mode, True, tui, 4)
    b = _run(h, spec, mode, False, tui, 4)
    assert (a["upd"], a["last"], a["step"]) == (b["upd"], b["last"], b["step"]) and a["upd"] == 8
    np.testing.assert_allclose(a["gnorm"].numpy(), b["gnorm"].numpy(), rtol=2e-6)
    np.testing.assert_allclose(a["loss"].numpy(), b["loss"].numpy(), rtol=1e-5)
    for k in ("params", "target", "m", "v"):
        np.testing.assert_allclose(a[k].numpy(), b[k].numpy(), rtol=2e-5, atol=2e-7, err_msg=k)
    if tui == 3:
        assert a["last"] == 6  # hard copies after updates 3 and 6: the target packs were rewritten by adam_pack_kernel
        assert not torch.equal(a["target"], a["params"])


def test_replay_gather_above_2gb_takes_the_64bit_path_and_matches_the_buffer_path():
    """The learner gathers its rows through buffer descriptors (32-bit offsets) while every replay array is below 2 GB and through
    64-bit global loads above: the same 512 episodes placed in a 2.2 GB replay (704,512 episodes x 3,120 B of observations) and in
    a small one must give bitwise the same loss and gradient."""
    from codebase_amd import hip as h
    from oracle import dqn_port as dp

    P, D, H, A, T, B = 2, 15, 64, 6, 25, 512
    big_cap, small_cap = 704512, 1024
    assert big_cap * P * (T + 1) * D * 4 >= 2**31
    g = torch.Generator().manual_seed(5)
    ep_obs = torch.randint(-1, 8, (B, P, T + 1, D), generator=g).float()
    ep_act = torch.randint(0, A, (B, P, T), generator=g).to(torch.uint8)
    ep_rew = torch.rand((B, P, T), generator=g)
    ln = torch.randint(3, T + 1, (B,), generator=g)
    t = torch.arange(T + 1)[None, :]
    ep_done = (t == ln[:, None]).to(torch.uint8)
    ep_fill = (t[:, :T] < ln[:, None]).to(torch.uint8)
    spec = h.NetSpec(P, D, H, A)
    params, target = dp.init_params(P, D, H, A, seed=1).to(DEV), dp.init_params(P, D, H, A, seed=2).to(DEV)
    out = []
    for cap in (small_cap, big_cap):
        rb = h.DeviceReplay(cap, P, D, T)
        slots = (torch.arange(B) * (cap // B) + 3) % cap  # spread over the whole buffer (the last ones sit above the 2 GB mark)
        sl = slots.to(DEV)
        rb.obs[sl] = ep_obs.to(DEV)
        rb.act[sl] = ep_act.to(DEV)
        rb.rew[sl] = ep_rew.to(DEV)
        rb.done[sl] = ep_done.to(DEV)
        rb.filled[sl] = ep_fill.to(DEV)
        up = h.DqnUpdater(spec, params, target)
        loss, grad = up.loss_grad_replay(rb, B, idx=slots.to(torch.int32).to(DEV))
        out.append((loss.clone().cpu(), grad.clone().cpu()))
        del rb
        torch.cuda.empty_cache()
    assert torch.isfinite(out[0][0]).all() and out[0][0][1] == float(ep_fill.sum())
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
