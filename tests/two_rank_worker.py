"""Worker of tests/test_gpu_two_ranks.py: run under torch.distributed.run with 2 processes sharing cuda:0 (gloo).
Each rank owns its env / replay shard (different Philox keys), gradients are all-reduced every update; after a few
rounds every rank must hold bit-identical parameters, optimiser moments and targets - for IDQN, QMIX and IA2C."""
import os
import time
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def same_on_all_ranks(t, what):
    ref = t.detach().cpu().clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(ref, t.detach().cpu()), f"rank {dist.get_rank()}: {what} diverged"


DIGEST = __import__("hashlib").sha256()


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from codebase_amd import hip as h
    from codebase_amd.ac.model import A2CNetwork
    from codebase_amd.ac.train import Batch
    from codebase_amd.dqn.model import QMixNetwork, QNetwork
    from codebase_amd.dqn.train import VectorisedIDQN
    from codebase_amd.parallel import GradSync, rank_env_seed
    from codebase_amd.utils.envs import _space_pair

    N, T = 256, 25
    cfg = h.lbf_config("lbforaging:Foraging-8x8-2p-3f-v3", N, T, seed=rank_env_seed(7, rank), cooperative=True)
    obs_space, act_space = _space_pair(cfg)
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False,
                 target_update_interval_or_tau=3)
    # (class, extra constructor arguments, hidden width, standardise_returns): hidden-64 IDQN takes the library's n-updates call in its
    # data-parallel form (marlhip_idqn_update_n_dist: reduce -> exchange -> clip + Adam + packs), hidden 128 the same call's generic
    # loop, QMIX its own library loop (marlhip_qmix_update_n, the joint [critic | mixer] gradient as one message), standardise_returns the
    # per-update host loop
    cases = ((QNetwork, (), 64, False), (QNetwork, (), 128, False), (QNetwork, (), 64, True),
             (QMixNetwork, (dict(embed_dim=64, hypernet_layers=2, hypernet_embed=32),), 64, False))
    for cls, extra, hidden, std in cases:
        torch.manual_seed(1)
        model = cls(obs_space, act_space, dict(hyper, standardise_returns=std), [hidden, hidden], False, False, True, *extra, "cuda")
        tr = VectorisedIDQN(cfg, model, 2 * N, T, 64, 4, seed=3, dist=dist)
        for r in range(3):
            tr.round(0.3)
        torch.cuda.synchronize()
        assert (tr._fused is not None) == (not std), "which cases take an n-updates library call changed"  # IDQN: marlhip_idqn_update_n_dist; QMIX: marlhip_qmix_update_n
        if os.environ.get("MARLHIP_P2P", "1") != "0":  # the gradient went through marlhip_p2p_allreduce (a C pointer inside the library loop, a ctypes call in the host loops)
            assert tr._sync is not None and tr._sync.p2p is not None and tr._sync.p2p.status() == 0, "p2p exchange missing or timed out"
        assert model.updates == 12 and model.updater.step == 12
        for name in ("params", "target_params"):
            same_on_all_ranks(getattr(model, name), f"{cls.__name__}.{name}")
        same_on_all_ranks(model.updater.exp_avg_sq, f"{cls.__name__} Adam moments")
        DIGEST.update(model.params.cpu().numpy().tobytes())  # compared between the exchanges by tests/test_gpu_two_ranks.py
        same_on_all_ranks(model.updater.gnorm, f"{cls.__name__} clip norm (taken from the REDUCED gradient)")
        if std:  # RunningMeanStd moved by the GLOBAL batch moments: the same (mean, var, count) on every rank
            st = model.ret_ms
            same_on_all_ranks(st.mean, "return statistics mean")
            same_on_all_ranks(st.var, "return statistics var")
            assert abs(st.count - (1e-4 + 12 * world * T * 64)) < 1e-6, st.count  # every update adds world * T * B entries per agent
            assert float(st.var.min()) > 0 and not torch.equal(st.mean, torch.zeros_like(st.mean))
        if cls is QMixNetwork:
            same_on_all_ranks(model.mixer_params, "mixer")
            same_on_all_ranks(model.target_mixer_params, "target mixer")
        # the shards really differ: the replay contents are rank-specific
        mine = tr.replay.obs[:8].cpu().clone()
        other = mine.clone()
        dist.broadcast(other, src=0)
        assert rank == 0 or not torch.equal(mine, other), "ranks collected identical episodes"
    # actor-critic
    torch.manual_seed(1)
    net = dict(layers=[64, 64], parameter_sharing=False, use_orthogonal_init=True, use_rnn=False)
    ac_hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=0.5, n_steps=5, entropy_coef=0.01, value_loss_coef=0.5,
                    standardise_returns=True, target_update_interval_or_tau=200)
    cfg2 = h.lbf_config("lbforaging:Foraging-8x8-2p-3f-v3", N, T, seed=rank_env_seed(7, rank))
    ac = A2CNetwork(obs_space, act_space, ac_hyper, net, dict(net, centralised=False), "cuda")
    P, D = 2, 15
    bufs = dict(o=torch.empty(T + 1, N, P * D, device="cuda"), a=torch.empty(T, N, P, dtype=torch.int64, device="cuda"),
                r=torch.empty(T, N, P, device="cuda"), d=torch.empty(T + 1, N, dtype=torch.uint8, device="cuda"),
                f=torch.empty(T, N, device="cuda"))
    fr, fl, tm = torch.zeros(P, N, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    sync = GradSync(dist, max_floats=ac.updater.grad.numel())  # what ac.train.main builds: the in-library exchange when it can be set up
    assert os.environ.get("MARLHIP_P2P", "1") == "0" or sync.p2p is not None, "the in-library p2p exchange did not come up on two ranks of one GPU"
    ac.updater.attach_exchange(lambda t: dist.all_reduce(t))  # what ac.train.main does under torchrun
    for r in range(3):
        h.ac_collect(cfg2, ac.spec, ac.actor_params, r, T, False, bufs["o"], bufs["a"], bufs["r"], bufs["d"], bufs["f"], fr, fl, tm)
        ac.update_async(Batch(bufs["o"], bufs["a"], bufs["r"], bufs["d"].float(), bufs["f"], None), r * 200, grad_sync=sync, world=world)
    torch.cuda.synchronize()
    sync.check()
    same_on_all_ranks(ac.block, "A2C actor|critic block")
    same_on_all_ranks(ac.target_critic_params, "A2C target critic")
    same_on_all_ranks(ac.updater.ret_stats.mean, "A2C return statistics mean")
    same_on_all_ranks(ac.updater.ret_stats.var, "A2C return statistics var")
    assert abs(ac.updater.ret_stats.count - (1e-4 + 3 * world * T * N)) < 1e-6
    dist.barrier()
    DIGEST.update(ac.block.cpu().numpy().tobytes())
    sync.close()
    # ia2c.yaml's own setting (no joint clip): the critics' half of every update on their own stream NEXT TO a gradient exchange - the
    # actors' slice reduced and stepped on the caller's stream, the critics' slice through the exchange's second lane on theirs
    # (A2CNetwork.update_async(overlap=True) after attach_grad_sync) - against the same three rounds with everything on one stream and
    # ONE exchange of the joint block: the same bits.  BASELINE config 4's kernels (rware-tiny-4ag, 128-128).
    Nw, Tw = 256, 60
    cfg3 = h.env_config("rware:rware-tiny-4ag-v2", Nw, Tw, seed=rank_env_seed(11, rank))
    o3, a3 = _space_pair(cfg3)
    Pw, (Dw, Aw) = cfg3.n_agents, h.env_dims(cfg3)
    net3 = dict(layers=[128, 128], parameter_sharing=False, use_orthogonal_init=True, use_rnn=False)
    hyp3 = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=False, n_steps=5, entropy_coef=0.001, value_loss_coef=0.5,
                standardise_returns=False, target_update_interval_or_tau=200)
    ends = {}
    for overlap in (True, False):
        torch.manual_seed(2)
        m3 = A2CNetwork(o3, a3, hyp3, net3, dict(net3, centralised=False), "cuda")
        s3 = GradSync(dist, max_floats=m3.updater.grad.numel(), side_floats=m3.updater.critic_grad.numel() if overlap else 0)
        sets = [dict(o=torch.empty(Tw + 1, Nw, Pw * Dw, device="cuda"), a=torch.empty(Tw, Nw, Pw, dtype=torch.int64, device="cuda"),
                     r=torch.empty(Tw, Nw, Pw, device="cuda"), d=torch.empty(Tw + 1, Nw, dtype=torch.uint8, device="cuda"),
                     df=torch.empty(Tw + 1, Nw, device="cuda"), f=torch.empty(Tw, Nw, device="cuda")) for _ in range(2)]
        fr3, fl3 = torch.zeros(Pw, Nw, device="cuda"), torch.zeros(Nw, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        deferred, stamps, t_start = [], [], time.time()
        if os.environ.get("MARLHIP_TWO_RANK_DIAG"):  # where is this rank's host when a lane's wait runs into its bound (scripts/gpu_runs/r6W.sh)
            import faulthandler

            faulthandler.dump_traceback_later(8, exit=False)
        with torch.cuda.stream(torch.cuda.Stream(device="cuda")):  # (nothing leaves a caller that is on the default stream)
            assert m3.attach_grad_sync(s3) == overlap
            for r in range(4):  # two alternating batch sets, as bench.py keeps them: round r + 2 rewrites the set round r's critics read
                b = sets[r & 1]
                stamps.append(round(time.time() - t_start, 2))
                h.ac_collect(cfg3, m3.spec, m3.actor_params, r, Tw, False, b["o"], b["a"], b["r"], b["d"], b["f"], fr3, fl3, tm, keep_for=m3.updater)
                b["df"].copy_(b["d"])
                m3.update_async(Batch(b["o"], b["a"], b["r"], b["df"], b["f"], None), r * Tw * Nw, grad_sync=s3, world=world, overlap=overlap)
                deferred.append(m3.updater._critic_event is not None)
        stamps.append(("enqueued", round(time.time() - t_start, 2)))
        torch.cuda.synchronize()
        stamps.append(("drained", round(time.time() - t_start, 2)))
        if os.environ.get("MARLHIP_TWO_RANK_DIAG"):
            faulthandler.cancel_dump_traceback_later()
            print(f"[diag] rank {rank} overlap={overlap} stamps {stamps}", file=sys.stderr, flush=True)
        assert deferred == [overlap] * 4, deferred
        if overlap and os.environ.get("MARLHIP_P2P", "1") != "0":
            lanes = dict(main=None if s3.p2p is None else s3.p2p.status(), side=None if s3.side is None or s3.side.p2p is None else s3.side.p2p.status())
            assert lanes == dict(main=0, side=0), f"rank {rank}: a p2p lane is missing (None) or ran into its peer timeout (1): {lanes}; round starts {stamps}"
        s3.check()
        for what, t in (("block", m3.block), ("target critic", m3.target_critic_params), ("exp_avg", m3.updater.exp_avg), ("exp_avg_sq", m3.updater.exp_avg_sq)):
            same_on_all_ranks(t, f"A2C (overlap={overlap}) {what}")
        ends[overlap] = [t.clone() for t in (m3.block, m3.target_critic_params, m3.updater.exp_avg, m3.updater.exp_avg_sq)]
        s3.close()
    for x, y in zip(ends[True], ends[False]):
        assert torch.equal(x, y), "the split exchange beside the deferred critics left other bits than the one-stream update"
    assert float(ends[True][3][-64:].abs().max()) > 0
    DIGEST.update(ends[True][0].cpu().numpy().tobytes())
    dist.barrier()
    if rank == 0:
        print("TWO_RANK_OK", DIGEST.hexdigest())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
