"""The actors' forward pass kept by the rollout (marlhip_ac_collect_keep / marlhip_rware_ac_collect_keep, marlhip_ac_config.actor_forward_kept;
csrc/mlp_keep.h, AcKeep in csrc/common.h).  A2C updates once per rollout on the parameters the rollout was sampled with
(marlbase/ac/train.py:203-212 -> ac/model.py:189-246), so the logits and hidden layers the step recomputes for every batch row are the
values the collector held when it sampled that row's action.  The claim tested here is the strong one: the step on the kept pass and the
step that runs the pass itself give the SAME BITS - metrics, actor and critic gradients - on every wave organisation of the collector
(one wave per env block / per agent / two per agent; packs in LDS / in global memory), both record layouts (hidden 64: h1 | h2 per slot;
hidden 128: h2 record | h1 record with the odd-block pad), rollouts longer than the episodes (rows the collector never computes) and
centralised critics.  Against the float64 port the kept path runs in tests/test_gpu_at_size_vs_oracle.py (configs 4 and MAA2C)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_gpu_at_size_vs_oracle import _collect_ac


def _step(model, b, kept):
    from codebase_amd.ac.train import Batch

    batch = Batch(b["obss"], b["actions"], b["rewards"], b["dones"].float(), b["filled"], None)
    up = model.updater
    m = up.a2c_loss_grad(batch, kept=kept).clone()
    torch.cuda.synchronize()
    return m, up.actor_grad.clone(), up.critic_grad.clone(), up.last_step_used_kept_forward


CASES = [
    # name, envs, time limit, rollout rows, hidden, centralised critic
    ("lbforaging:Foraging-8x8-2p-3f-v3", 4096, 25, 25, 64, False),   # two waves per agent (mlp_forward_h2_keep), h1 | h2 slots
    ("lbforaging:Foraging-8x8-2p-3f-v3", 4096, 25, 25, 128, False),  # two waves per agent with the output operands in registers, h2 | h1 records
    ("lbforaging:Foraging-8x8-2p-3f-v3", 80, 25, 25, 128, False),    # 5 env blocks per step: the pair slot tp_bwd walks past is zero-filled
    ("lbforaging:Foraging-8x8-2p-3f-v3", 16384, 25, 25, 64, False),  # one wave per env block (mlp_forward_p)
    ("lbforaging:Foraging-8x8-2p-3f-v3", 512, 6, 11, 64, False),     # every episode is over at t = 6: rows 6..10 are never computed
    ("lbforaging:Foraging-8x8-2p-3f-v3", 512, 6, 11, 128, False),
    ("lbforaging:Foraging-10x10-3p-3f-v3", 1024, 25, 25, 64, False),  # odd agent count: one wave per env block
    ("lbforaging:Foraging-15x15-4p-5f-v3", 2048, 25, 25, 128, False),  # agent per wave, packs from global memory (mlp_forward_g_keep)
    ("lbforaging:Foraging-8x8-2p-3f-v3", 1024, 25, 25, 64, True),     # centralised critics (fused 2-agent shape) next to kept actors
    ("lbforaging:Foraging-15x15-4p-5f-v3", 2048, 25, 25, 128, True),  # fused 4-agent centralised critics
    ("lbforaging:Foraging-15x15-8p-5f-v3", 1024, 25, 25, 128, True),  # MAA2C: 8 actors, wide critics
    ("rware:rware-tiny-2ag-v2", 512, 40, 40, 64, False),
    ("rware:rware-tiny-4ag-v2", 256, 60, 60, 128, False),             # BASELINE config 4's kernels
]


@pytest.mark.parametrize("name,N,T,L,H,central", CASES)
def test_a2c_step_on_the_kept_forward_pass_has_the_bits_of_the_recomputed_one(name, N, T, L, H, central):
    from codebase_amd import hip as h

    seed, rnd = 5, 2
    _, model, b, fin_len, _ = _collect_ac(h, name, N, T, H, seed, rnd, central=central, scale=2.0, keep=True, max_len=L)
    assert model.updater._kept is not None
    m1, ga1, gc1, used1 = _step(model, b, kept=True)
    assert used1
    # the same rollout again without the record (same round index: same streams), into fresh tensors
    _, model2, b2, fin_len2, _ = _collect_ac(h, name, N, T, H, seed, rnd, central=central, scale=2.0, keep=False, max_len=L)
    for k in b:
        assert torch.equal(b[k], b2[k]), k
    assert torch.equal(fin_len, fin_len2)
    m2, ga2, gc2, used2 = _step(model2, b2, kept=True)  # nothing was kept: the step runs the pass itself
    assert not used2
    assert torch.equal(m1, m2), (m1.tolist(), m2.tolist())
    assert torch.equal(ga1, ga2) and torch.equal(gc1, gc2)
    assert float(ga1.abs().max()) > 0 and float(m1[4]) == float(b["filled"].sum())
    if L > T:
        assert float(b["filled"][T:].sum()) == 0.0
    # forced recomputation on the model that holds a record: the same bits again
    m3, ga3, gc3, used3 = _step(model, b, kept=False)
    assert not used3 and torch.equal(m1, m3) and torch.equal(ga1, ga3) and torch.equal(gc1, gc3)


@pytest.mark.parametrize("name,N,T,H,central", [("lbforaging:Foraging-8x8-2p-3f-v3", 4096, 25, 64, False), ("lbforaging:Foraging-8x8-2p-3f-v3", 1024, 25, 128, True),
                                                 ("rware:rware-tiny-4ag-v2", 256, 60, 128, True), ("rware:rware-tiny-2ag-v2", 512, 40, 64, False)])
def test_ppo_update_reads_the_kept_pass_for_old_log_probs_and_the_first_epoch(name, N, T, H, central):
    """PPONetwork.update (marlbase/ac/model.py:264-352): the old log-probs and the first of the num_epochs steps run on the parameters the
    rollout was sampled with - both take the collector's forward pass; the first apply() voids it.  Whole update, 4 epochs with clip and
    Polyak target: the joint parameter block, the target critics and the mean metrics have the bits of the update that recomputes."""
    from codebase_amd import hip as h
    from codebase_amd.ac.train import Batch

    outs = []
    for keep in (True, False):
        _, model, b, _, _ = _collect_ac(h, name, N, T, H, 11, 1, central=central, scale=2.0, keep=keep, ppo=True)
        up = model.updater
        used = []
        orig = up.ppo_loss_grad
        up.ppo_loss_grad = lambda batch, kept=True: (orig(batch, kept), used.append(up.last_step_used_kept_forward))[0]
        batch = Batch(b["obss"], b["actions"], b["rewards"], b["dones"].float(), b["filled"], None)
        m = model.update_async(batch, step=N * T).clone()
        torch.cuda.synchronize()
        assert used == ([True, False, False, False] if keep else [False] * 4)
        assert up._kept is None or not keep
        outs.append((m, model.block.clone(), model.target_critic_params.clone(), {k: v.clone() for k, v in b.items()}))
    (m1, p1, t1, b1), (m2, p2, t2, b2) = outs
    for k in b1:
        assert torch.equal(b1[k], b2[k]), k
    assert torch.equal(m1, m2) and torch.equal(p1, p2) and torch.equal(t1, t2)


def test_kept_pass_is_void_after_the_parameters_moved_and_for_other_batches():
    from codebase_amd import hip as h

    name, N, T, H = "lbforaging:Foraging-8x8-2p-3f-v3", 256, 25, 64
    _, model, b, _, _ = _collect_ac(h, name, N, T, H, 9, 0, scale=2.0, keep=True)
    other = {k: v.clone() for k, v in b.items()}
    assert not _step(model, other, kept=True)[3]  # same contents, another observation tensor: not this rollout's batch
    assert _step(model, b, kept=True)[3]
    model.updater.apply()
    assert model.updater._kept is None and not _step(model, b, kept=True)[3]


def test_keep_refusals():
    from codebase_amd import hip as h
    _, model, b, _, _ = _collect_ac(h, "lbforaging:Foraging-8x8-2p-3f-v3", 40, 25, 64, 3, 0, keep=True)  # 40 envs: not whole blocks of 16
    assert model.updater._kept is None
    # the library itself refuses a record for a rollout it was not laid out for
    import ctypes

    from codebase_amd._lib import lib
    up = model.updater
    cfg = h.env_config("lbforaging:Foraging-8x8-2p-3f-v3", 40, 25, seed=3)
    s = model.spec.c()
    ws = up._workspace(25, 40)
    P, D = 2, 15
    dev = "cuda"
    bufs = (torch.empty(26, 40, P * D, device=dev), torch.empty(25, 40, P, dtype=torch.int64, device=dev), torch.empty(25, 40, P, device=dev),
            torch.empty(26, 40, dtype=torch.uint8, device=dev), torch.empty(25, 40, device=dev), torch.zeros(P, 40, device=dev),
            torch.zeros(40, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
    fws = h._fwd_ws(model.spec, model.actor_params.device)
    rc = lib.marlhip_ac_collect_keep(ctypes.byref(cfg), ctypes.byref(s), model.actor_params.data_ptr(), 0, 25, 0, *(t.data_ptr() for t in bufs), *fws, 0,
                                     ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b"16" in lib.marlhip_last_error()


# ---- the critics' half of an A2C update next to the following rollout (marlhip_ac_config.defer_critic_backward) --------------------------
def _rounds(h, name, N, T, H, central, overlap, tui, rounds=4):
    """rollout -> A2CNetwork.update_async x rounds, a fresh batch per rollout as ac/train.py allocates them"""
    from codebase_amd.ac.train import Batch

    cfg, model, b, _, (P, D, A) = _collect_ac(h, name, N, T, H, 13, 0, central=central, scale=1.5, keep=True)
    model.target_update_interval_or_tau = tui
    dev, ms, deferred = model.device, [], []
    torch.cuda.synchronize()
    with torch.cuda.stream(torch.cuda.Stream(device=dev)):  # (the critics only leave a caller that is not on the default stream)
        out = _rounds_body(h, cfg, model, b, N, T, P, D, rounds, overlap, ms, deferred)
    torch.cuda.synchronize()
    return out, deferred, model


def _rounds_body(h, cfg, model, b, N, T, P, D, rounds, overlap, ms, deferred):
    from codebase_amd.ac.train import Batch

    dev = model.device
    for r in range(rounds):
        if r > 0:
            b = dict(obss=torch.empty(T + 1, N, P * D, device=dev), actions=torch.empty(T, N, P, dtype=torch.int64, device=dev),
                     rewards=torch.empty(T, N, P, device=dev), dones=torch.empty(T + 1, N, dtype=torch.uint8, device=dev),
                     filled=torch.empty(T, N, device=dev))
            fr, fl, tm = torch.zeros(P, N, device=dev), torch.zeros(N, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
            assert h.ac_collect(cfg, model.spec, model.actor_params, r, T, False, b["obss"], b["actions"], b["rewards"], b["dones"], b["filled"], fr, fl, tm,
                                keep_for=model.updater)
        batch = Batch(b["obss"], b["actions"], b["rewards"], b["dones"].float(), b["filled"], None)
        ms.append(model.update_async(batch, step=200 * r, overlap=overlap).clone())
        deferred.append(model.updater._critic_event is not None)
    return (torch.stack(ms), model.block.clone(), model.target_critic_params.clone(), model.updater.exp_avg.clone(), model.updater.exp_avg_sq.clone())


@pytest.mark.parametrize("name,N,T,H,central,tui,overlap,expect", [
    ("rware:rware-tiny-4ag-v2", 256, 60, 128, False, 200, True, True),             # BASELINE config 4's kernels: the rollout leaves most of the chip to the critics
    ("rware:rware-tiny-2ag-v2", 512, 40, 64, False, 0.01, True, True),             # Polyak target on the critics' stream
    ("lbforaging:Foraging-8x8-2p-3f-v3", 2048, 25, 64, False, 400, True, True),    # two waves per agent: 128 workgroups of 4 waves
    ("lbforaging:Foraging-8x8-2p-3f-v3", 4096, 25, 64, False, 400, True, False),   # a rollout that fills the chip: nothing is deferred
    ("lbforaging:Foraging-8x8-2p-3f-v3", 1024, 25, 128, True, 200, True, True),    # fused centralised critics, 30 inputs: about an actor row's cost
    ("lbforaging:Foraging-15x15-8p-5f-v3", 512, 25, 128, True, 0.05, True, False),  # 312-input critics: their half-speed backward pass would outlast the rollout
    ("lbforaging:Foraging-15x15-8p-5f-v3", 512, 25, 128, True, 0.05, "force", True),  # ... the same on request (wc_* kernels on the critics' stream)
    ("lbforaging:Foraging-8x8-2p-3f-v3", 4096, 25, 128, False, 200, "force", True),
])
def test_update_with_the_critics_half_overlapping_the_next_rollout_has_the_bits_of_the_sequential_one(name, N, T, H, central, tui, overlap, expect):
    """A2CNetwork.update_async(overlap=True): actors' backward + step on the caller's stream, critics' backward + step + target update on a
    stream of their own while the next rollout runs; four rollout -> update rounds end with the same joint parameter block, target critics,
    Adam moments and per-update metrics as the one-stream update (the launches and their inputs are the same; only their placement moved)"""
    from codebase_amd import hip as h

    (m1, p1, t1, a1, v1), d1, _ = _rounds(h, name, N, T, H, central, overlap, tui)
    (m2, p2, t2, a2, v2), d2, _ = _rounds(h, name, N, T, H, central, False, tui)
    # the critics leave the caller's stream where AcUpdater.can_defer says it pays (or on request)
    assert d1 == [expect] * len(d1) and not any(d2)
    assert torch.equal(m1, m2) and torch.equal(p1, p2) and torch.equal(t1, t2) and torch.equal(a1, a2) and torch.equal(v1, v2)
    assert float(v1[-64:].abs().max()) > 0  # (the critics' slice did step)


def test_a_joint_gradient_clip_keeps_the_update_on_one_stream():
    from codebase_amd import hip as h

    name, N, T, H = "lbforaging:Foraging-8x8-2p-3f-v3", 256, 25, 64
    _, model, b, _, _ = _collect_ac(h, name, N, T, H, 4, 0, keep=True)
    from codebase_amd.ac.train import Batch
    model.updater.grad_clip = 0.5  # clip_grad_norm_ over actor AND critic (ac/model.py:227-229): the actors' step needs the critics' gradient
    model.update_async(Batch(b["obss"], b["actions"], b["rewards"], b["dones"].float(), b["filled"], None), step=0, overlap=True)
    assert model.updater._critic_event is None and model.updater._critic_pending is None
