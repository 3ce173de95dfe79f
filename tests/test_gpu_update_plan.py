"""Filled-aware learner updates (csrc/update_plan.h).  The reference's loss is a filled-weighted sum
(marlbase/dqn/model.py:160-163): rows behind an episode's last transition contribute exactly zero.  marlhip_idqn_update_n therefore
orders an update's sampled episodes by stored length and lets a 16-episode tile walk only the steps its longest episode has.

Checked here: the plan itself (marlhip_update_plan: a stable sort of the library's own Philox draws, every tile's [0, L) covered exactly
once, nothing beyond it, balanced over the waves), the planned update against the unplanned one (MARLHIP_NO_PLAN=1: same loss / gradient
up to summation order; the SAME BITS when every episode runs to the time limit), and the planned path's throughput gain on a ragged
replay.  Against the float64 port the planned path is what tests/test_gpu_bench_path_vs_oracle.py runs at B = 4096 (its replay is ragged)."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_gpu_bench_path_vs_oracle import _perturbed, lbf_like_replay, philox_indices, to_device

DEV = "cuda"


def _plans(h, rb, P, B, length, seed, counter0, n):
    from codebase_amd._lib import check, lib

    dims = (ctypes.c_int32 * 8)()
    check(lib.marlhip_update_plan(ctypes.byref(rb.shape), ctypes.byref(rb.bufs), P, B, length, seed, counter0, 0, None, 0, None, dims, None), "update_plan")
    d = dict(zip(("planned", "stride", "hdr", "waves", "slots", "nc", "tiles", "T"), list(dims)))
    out = torch.zeros(n * d["stride"], dtype=torch.int32, device=DEV)
    idx = torch.zeros(B, dtype=torch.int32, device=DEV)
    check(lib.marlhip_update_plan(ctypes.byref(rb.shape), ctypes.byref(rb.bufs), P, B, length, seed, counter0, n, out.data_ptr(), out.numel(), idx.data_ptr(),
                                  dims, torch.cuda.current_stream().cuda_stream), "update_plan")
    torch.cuda.synchronize()
    return d, out.cpu().numpy().reshape(n, d["stride"]), idx.cpu().numpy()


@pytest.mark.parametrize("P,B,cap,full", [(2, 4096, 8192, False), (2, 4096, 8192, True), (4, 8192, 9000, False), (2, 256, 300, False), (3, 1000, 1500, False)])
def test_plan_is_a_stable_length_sort_of_the_draws_and_covers_every_filled_step_once(P, B, cap, full):
    from codebase_amd import hip as h

    D, T, A, seed, n = 15, 25, 6, 99, 3
    host = lbf_like_replay(cap, P, D, T, A, seed=5)
    if full:
        host["filled"][:] = 1
        host["done"][:] = 0
    rb = h.DeviceReplay(cap, P, D, T)
    to_device(h, rb, host)
    d, plans, idx_last = _plans(h, rb, P, B, cap, seed, 7, n)
    assert d["planned"] == 1 and d["T"] == T and d["tiles"] == (B + 15) // 16
    lens = host["filled"].sum(1).numpy().astype(np.int64)
    W = d["waves"]
    for u in range(n):
        draws = philox_indices(seed, 7 + u, B, cap)
        if u == n - 1:
            np.testing.assert_array_equal(idx_last, draws)  # marlhip_idqn_learner.idx: the last update's draws, in draw order
        pl = plans[u]
        slots, c, lmax, total = (int(x) for x in pl[:4])
        srt = pl[4:4 + B].astype(np.int64)
        order = np.argsort(-lens[draws], kind="stable")  # longest first, equal lengths in draw order
        np.testing.assert_array_equal(srt, draws[order])
        assert lmax == int(lens[draws].max()) and total == int(lens[draws].sum())
        table = pl[4 + B:4 + B + d["slots"] * W].reshape(d["slots"], W)
        assert 1 <= slots <= d["slots"] and not table[slots:].any()
        covered = np.zeros((d["tiles"], T), np.int32)
        steps = np.zeros(W, np.int64)  # per wave: transitions + one bootstrap step per task
        for s in range(slots):
            for w in range(W):
                e = int(table[s, w])
                grp, t0, t1 = e >> 16, (e >> 8) & 255, e & 255
                if t1 <= t0:
                    continue
                covered[grp, t0:t1] += 1
                steps[w] += t1 - t0 + 1
        tile_len = lens[srt[::16]]
        if full:  # the static plan, entry for entry: tile-major tasks, chunk boundaries (k T) / nc
            assert c == T and slots == -(-d["tiles"] * d["nc"] // W)
            flat = table.reshape(-1)
            for task in range(d["tiles"] * d["nc"]):
                grp, k = divmod(task, d["nc"])
                assert int(flat[task]) == (grp << 16) | ((k * T // d["nc"]) << 8) | ((k + 1) * T // d["nc"])
        for k in range(d["tiles"]):
            L = T if full else int(tile_len[k])
            assert (covered[k, :L] == 1).all() and not covered[k, L:].any(), (u, k, L, covered[k])
        if not full:
            # balanced: no wave walks more than `slots` chunks of c (+ their bootstrap steps), and the plan beats the static walk
            assert steps.max() <= slots * (c + 1)
            static_steps = -(-d["tiles"] * d["nc"] // W) * (-(-T // d["nc"]) + 1)
            assert steps.max() <= static_steps and (B < 2048 or steps.max() < static_steps), (steps.max(), static_steps)


def _run(h, host, P, D, H, A, T, B, cap, n, planned, lr=3e-4, tui=200):
    if planned:
        os.environ.pop("MARLHIP_NO_PLAN", None)
    else:
        os.environ["MARLHIP_NO_PLAN"] = "1"
    try:
        spec = h.NetSpec(P, D, H, A)
        rb = h.DeviceReplay(cap, P, D, T)
        to_device(h, rb, host)
        params, target = _perturbed(P, D, H, A, 3).to(DEV), _perturbed(P, D, H, A, 4).to(DEV)
        up = h.DqnUpdater(spec, params, target, lr=lr, gamma=0.99, grad_clip=1.0, double_q=True)
        fl = h.FusedLearner(up, rb, B, tui, mode=0)
        fl.run(n, cap, 4321, 11, 0, 0)
        torch.cuda.synchronize()
        return (params.cpu(), target.cpu(), up.grad.cpu().clone(), up.loss.cpu().clone(), up.gnorm.cpu().clone(), rb._outputs(B)[5].cpu().clone(),
                up.exp_avg.cpu().clone())
    finally:
        os.environ.pop("MARLHIP_NO_PLAN", None)


@pytest.mark.parametrize("P,D,B", [(2, 15, 4096), (3, 18, 2048), (2, 15, 512)])
def test_planned_update_equals_the_unplanned_one_up_to_summation_order(P, D, B):
    """one update on a ragged replay (lengths uniform in 1..T): the gradient entry by entry within 2e-6 of its largest (float sums of the
    same non-zero terms in another order), the loss within 1e-6, sum(filled) and the recorded draws identical"""
    from codebase_amd import hip as h

    H, A, T, cap = 64, 6, 25, 6000
    host = lbf_like_replay(cap, P, D, T, A, seed=17)
    a = _run(h, host, P, D, H, A, T, B, cap, 1, planned=True)
    b = _run(h, host, P, D, H, A, T, B, cap, 1, planned=False)
    assert torch.equal(a[5], b[5]) and float(a[3][1]) == float(b[3][1]) > 0
    assert abs(float(a[3][0]) - float(b[3][0])) <= 1e-6 * abs(float(b[3][0]))
    ga, gb = a[2].double(), b[2].double()
    assert float((ga - gb).abs().max()) <= 2e-6 * float(gb.abs().max()), float((ga - gb).abs().max() / gb.abs().max())
    assert float((a[4] - b[4]).abs()) <= 1e-6 * float(b[4])
    assert not torch.equal(a[2], b[2])  # (the plan did change the walk: other chunk boundaries, another row order)


def test_planned_update_has_the_bits_of_the_unplanned_one_when_every_episode_is_full_length():
    """three updates incl. a hard target copy on a replay whose episodes all run to the time limit (a fresh run's regime, the headline's):
    the stable sort is the identity, the table is the static plan - parameters, target, Adam moments, loss, clip norm: the same bits"""
    from codebase_amd import hip as h

    P, D, H, A, T, B, cap = 2, 15, 64, 6, 25, 4096, 8192
    host = lbf_like_replay(cap, P, D, T, A, seed=23)
    host["filled"][:] = 1
    host["done"][:] = 0
    host["done"][:, T] = 1
    a = _run(h, host, P, D, H, A, T, B, cap, 3, planned=True, tui=2)
    b = _run(h, host, P, D, H, A, T, B, cap, 3, planned=False, tui=2)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_planned_updates_are_faster_on_a_ragged_replay():
    """lengths uniform in 1..25 (mean 13): the planned learner launch group takes well under the static walk's time (library timers)"""
    from codebase_amd import hip as h
    from codebase_amd._lib import lib

    P, D, H, A, T, B, cap = 2, 15, 64, 6, 25, 4096, 8192
    host = lbf_like_replay(cap, P, D, T, A, seed=29)
    us = {}
    for planned in (True, False):
        _run(h, host, P, D, H, A, T, B, cap, 4, planned=planned)  # warm-up (attributes, allocator)
        lib.marlhip_timing_enable(1)
        _run(h, host, P, D, H, A, T, B, cap, 16, planned=planned)
        n, ms = ctypes.c_int64(0), ctypes.c_double(0.0)
        lib.marlhip_timing_read(0, ctypes.byref(n), ctypes.byref(ms))
        lib.marlhip_timing_enable(0)
        assert n.value == 16
        us[planned] = 1e3 * ms.value / n.value
    print(f"[plan] loss/grad launch, B = 4096, mean episode length 13: planned {us[True]:.1f} us, static {us[False]:.1f} us")
    assert us[True] < 0.8 * us[False], us


def test_more_updates_than_one_planning_launch_holds():
    """B = 256 leaves room for ~100 plans in the workspace's idle mixer planes: a call of 260 updates plans in three launches (the counter of
    each chunk continues the stream, the last chunk records the last update's draws) - same parameters as the unplanned call up to summation
    order, the recorded draws those of update 259"""
    from codebase_amd import hip as h

    P, D, H, A, T, B, cap, n = 2, 15, 64, 6, 25, 256, 2000, 260
    host = lbf_like_replay(cap, P, D, T, A, seed=31)
    a = _run(h, host, P, D, H, A, T, B, cap, n, planned=True, lr=1e-4)
    b = _run(h, host, P, D, H, A, T, B, cap, n, planned=False, lr=1e-4)
    np.testing.assert_array_equal(a[5].numpy(), philox_indices(4321, 11 + n - 1, B, cap))
    assert torch.equal(a[5], b[5]) and float(a[3][1]) == float(b[3][1])
    # 260 Adam steps amplify last-bit differences of the gradients (lr g / (sqrt(v) + eps)); the trajectories stay together to a few lr
    assert float((a[0] - b[0]).abs().max()) <= 20 * 1e-4 and float((a[0] - b[0]).abs().mean()) <= 1e-5
    assert abs(float(a[3][0]) - float(b[3][0])) <= 1e-3 * abs(float(b[3][0]))
