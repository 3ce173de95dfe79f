"""GPU: the reference-shaped python surface (make_env, QNetwork, ReplayBuffer, dqn.train.main, run)
on top of the HIP library."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.lbf import MarlbaseEnv
from oracle.philox import DrawStream

G = os.path.join(os.path.dirname(__file__), "golden")
NAME = "lbforaging:Foraging-8x8-2p-3f-v3"


def test_single_env_api_matches_oracle_episode_by_episode():
    from codebase_amd.utils.envs import make_env

    env = make_env(seed=5, name=NAME, time_limit=25, clear_info=False, observe_id=False, standardise_rewards=False,
                   wrappers=None)
    assert env.unwrapped.n_agents == 2 and len(env.observation_space) == 2 and env.observation_space[0].shape == (15,)
    rng = np.random.default_rng(0)
    for episode in range(3):
        obs, info = env.reset()
        ref = MarlbaseEnv(NAME, 25)
        robs, _ = ref.reset(DrawStream(5, 0, episode))
        assert isinstance(obs, tuple) and all(o.dtype == np.float32 for o in obs)
        for p in range(2):
            np.testing.assert_array_equal(obs[p], robs[p])
        done = False
        while not done:
            a = [int(x) for x in rng.integers(0, 6, 2)]
            obs, rew, term, trunc, info = env.step(a)
            robs, rrew, rterm, rtrunc, rinfo = ref.step(a)
            assert isinstance(rew, list) and len(rew) == 2 and isinstance(term, bool) and isinstance(trunc, bool)
            for p in range(2):
                np.testing.assert_array_equal(obs[p], robs[p])
            assert [np.float32(r) for r in rrew] == [np.float32(r) for r in rew]
            assert (term, trunc) == (rterm, rtrunc)
            done = term or trunc
        np.testing.assert_array_equal(info["episode_returns"], rinfo["episode_returns"].astype(np.float32))
        assert info["episode_length"] == rinfo["episode_length"] and "agent1/episode_returns" in info and "episode_time" in info


def test_cooperative_wrapper_and_unsupported_options():
    from codebase_amd.utils.envs import make_env

    env = make_env(seed=1, name=NAME, time_limit=25, wrappers=["CooperativeReward"])
    assert env.cfg.cooperative == 1
    with pytest.raises(NotImplementedError):
        make_env(seed=1, name=NAME, time_limit=25, wrappers=["FlattenObservation"])
    with pytest.raises(NotImplementedError):
        make_env(seed=1, name="smaclite/2s3z-v0", time_limit=150)
    rw = make_env(seed=1, name="rware:rware-tiny-4ag-v2", time_limit=500)  # the warehouse has a HIP env
    assert rw.n_agents == 4 and rw.observation_space[0].shape == (71,) and rw.action_space[0].n == 5


def reference_style_net(sd, i, D, H, A):
    net = torch.nn.Sequential(torch.nn.Linear(D, H), torch.nn.ReLU(), torch.nn.Linear(H, H), torch.nn.ReLU(), torch.nn.Linear(H, A))
    net.load_state_dict({k.split(f"independent.{i}.network.")[1]: v.cpu() for k, v in sd.items() if k.startswith(f"critic.independent.{i}.")})
    return net


def test_qnetwork_interface_state_dict_and_act():
    from codebase_amd.dqn.model import QNetwork
    from codebase_amd.utils.envs import _space_pair
    from codebase_amd import hip as h

    g = np.load(os.path.join(G, "init.npz"))
    cfg = h.lbf_config(NAME, 1, 25)
    obs_space, act_space = _space_pair(cfg)
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False,
                 target_update_interval_or_tau=200)
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    torch.manual_seed(123)
    m = QNetwork(obs_space, act_space, hyper, [64, 64], False, False, True, "cuda")
    torch.set_num_threads(nt)
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["keys_H64"])  # the reference's checkpoint key names, same order
    # LAPACK's QR rounding differs between host CPUs: bit-exact on the golden's machine (tests/test_host_cpu.py)
    np.testing.assert_allclose(m.params.cpu().numpy(), g["critic_H64_orth1"], rtol=0, atol=2e-5)
    # act(): greedy branch == torch argmax of a reference-shaped net loaded from the state_dict
    nets = [reference_style_net(sd, i, 15, 64, 6) for i in range(2)]
    rng = np.random.default_rng(0)
    for _ in range(20):
        obs = tuple(rng.integers(-1, 8, 15).astype(np.float32) for _ in range(2))
        acts, hid = m.act(obs, m.init_hiddens(1), 0.0)
        q = [nets[i](torch.tensor(obs[i])) for i in range(2)]
        top = [torch.sort(x).values for x in q]
        for i in range(2):
            if top[i][-1] - top[i][-2] > 1e-4:
                assert acts[i] == int(q[i].argmax())
        assert all(isinstance(a, int) for a in acts)
    acts, _ = m.act(obs, None, 1.0)  # epsilon 1: action_space.sample()
    assert len(acts) == 2 and all(0 <= a < 6 for a in acts)
    # load_state_dict round trip + save/load through torch.save
    sd2 = {k: v + 1.0 for k, v in sd.items()}
    m.load_state_dict(sd2)
    assert torch.equal(m.state_dict()["target.independent.1.network.4.bias"], sd2["target.independent.1.network.4.bias"])
    with pytest.raises(NotImplementedError):  # recurrent networks: one to four stacked GRU layers of width <= 128 (tests/test_gru.py, test_gru_stacked.py)
        QNetwork(obs_space, act_space, hyper, [32] * 6, False, True, True, "cuda")
    shared = QNetwork(obs_space, act_space, hyper, [64, 64], True, False, True, "cuda")  # parameter_sharing=True
    assert shared.params.shape[0] == 1 and "critic.networks.0.network.0.weight" in shared.state_dict()


def test_replay_adapter_follows_reference_trace():
    from codebase_amd.dqn.train import ReplayBuffer
    from codebase_amd import spaces

    g = dict(np.load(os.path.join(G, "replay.npz")))
    P, D, T, CAP = int(g["P"]), int(g["D"]), int(g["T"]), int(g["CAP"])
    obs_space = spaces.Tuple([spaces.Box(-1.0, 8.0, shape=(D,)) for _ in range(P)])
    rb = ReplayBuffer(CAP, P, obs_space, spaces.Tuple([spaces.Discrete(6)] * P), T, "cuda")
    for kind, o, a, r, d in zip(g["kind"], g["obs"], g["acts"], g["rews"], g["done"]):
        if kind == 0:
            rb.init_episode(list(o))
        else:
            rb.add(list(o), a, r, bool(d))
    assert rb.pos == int(g["pos"]) and len(rb) == int(g["length"]) and rb.can_sample(4) and not rb.can_sample(7)
    orig = np.random.randint
    np.random.randint = lambda lo, hi, size: g["idx"][:size]
    try:
        b = rb.sample(len(g["idx"]))
    finally:
        np.random.randint = orig
    for k in ("obss", "actions", "rewards", "dones", "filled"):
        np.testing.assert_array_equal(getattr(b, k).cpu().numpy(), g[k])
    assert b.action_mask is None


def test_run_entry_point_scalar_and_vectorised(tmp_path, monkeypatch):
    from codebase_amd import run

    monkeypatch.chdir(tmp_path)
    os.makedirs("scalar")
    monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / "scalar"))
    df = run.main(["+algorithm=idqn", f"env.name={NAME}", "env.time_limit=25", "algorithm.model.layers=[64,64]", "seed=1",
                   "algorithm.total_steps=400", "algorithm.training_start=100", "algorithm.batch_size=4",
                   "algorithm.eval_interval=200", "algorithm.eval_episodes=3", "algorithm.save_interval=300"])
    assert df.shape[0] >= 1 and {"updates", "mean_episode_returns", "loss"} <= set(df.columns)
    assert np.isfinite(df["loss"]).all() and any(f.startswith("model_s") for f in os.listdir(tmp_path / "scalar" / "checkpoints"))
    monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / "vec"))
    # 128 envs, 128 updates of 128 episodes per round (1 update per 25 env-steps, as the reference)
    df = run.main(["+algorithm=idqn", f"env.name={NAME}", "env.time_limit=25", "env.parallel_envs=128",
                   "algorithm.model.layers=[64,64]", "seed=1", "algorithm.total_steps=1500000", "algorithm.eval_interval=150000",
                   "algorithm.eval_episodes=512", "algorithm.updates_per_round=128"])
    assert df.shape[0] >= 8 and np.isfinite(df["loss"]).all()
    r = df["mean_episode_returns"].to_numpy()
    print("vectorised IDQN mean eval returns:", r, "updates:", df["updates"].to_numpy())
    assert r[-3:].mean() > r[:2].mean() + 0.02  # it learns: evaluation returns go up


def test_vdn_algorithm_end_to_end(tmp_path, monkeypatch):
    """+algorithm=vdn: CooperativeReward env + VDNetwork through the vectorised driver"""
    from codebase_amd import run

    monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / "vdn"))
    df = run.main(["+algorithm=vdn", f"env.name={NAME}", "env.time_limit=25", "env.parallel_envs=128",
                   "algorithm.model.layers=[64,64]", "seed=2", "algorithm.total_steps=300000", "algorithm.eval_interval=100000",
                   "algorithm.eval_episodes=256", "algorithm.updates_per_round=64"])
    assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all() and np.isfinite(df["mean_episode_returns"]).all()
    # cooperative reward: both agents log the same per-episode team return split (raw env rewards stay per agent)
    assert (df["updates"].to_numpy() > 0).all()


def test_qmix_network_interface_and_algorithm_end_to_end(tmp_path, monkeypatch):
    """QMixNetwork: the reference's constructor signature, its state_dict key set and order (golden from the reference),
    checkpoint round trip, then +algorithm=qmix through the vectorised driver"""
    from codebase_amd import run
    from codebase_amd.dqn.model import QMixNetwork
    from codebase_amd.spaces import Box, Discrete, Tuple

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "init_qmix.npz"))
    obs_space = Tuple([Box(-1, 8, (15,)) for _ in range(2)])
    act_space = Tuple([Discrete(6) for _ in range(2)])
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False,
                 target_update_interval_or_tau=200)
    torch.manual_seed(5)
    net = QMixNetwork(obs_space, act_space, hyper, [64, 64], False, False, True,
                      dict(embed_dim=64, hypernet_layers=2, hypernet_embed=32), "cuda")
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    np.testing.assert_array_equal(net.mixer_params.cpu().numpy(), g["mixer"])
    assert len(net.parameters()) == 2 * 6 + 14
    net2 = QMixNetwork(obs_space, act_space, hyper, [64, 64], False, False, True, net.mixing, "cuda")
    net2.load_state_dict(sd)
    assert torch.equal(net2.mixer_params, net.mixer_params) and torch.equal(net2.params, net.params)
    net.mixer_params.add_(1.0)
    net.hard_update()
    assert torch.equal(net.target_mixer_params, net.mixer_params)
    monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / "qmix"))
    df = run.main(["+algorithm=qmix", f"env.name={NAME}", "env.time_limit=25", "env.parallel_envs=128",
                   "algorithm.model.layers=[64,64]", "seed=2", "algorithm.total_steps=300000", "algorithm.eval_interval=100000",
                   "algorithm.eval_episodes=256", "algorithm.updates_per_round=32"])
    assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all() and np.isfinite(df["mean_episode_returns"]).all()
    assert (df["updates"].to_numpy() > 0).all()


def test_standardise_rewards_option(tmp_path, monkeypatch):
    """env.standardise_rewards=True (StandardiseReward, utils/wrappers.py:111-142): the scalar env API reproduces the
    oracle's wrapper bit for bit across episodes; the fused collector stores exactly the rewards the step-by-step
    path produces; the drivers accept the option."""
    from codebase_amd import hip as h
    from codebase_amd import run
    from codebase_amd.utils.envs import make_env

    env = make_env(seed=5, name=NAME, time_limit=25, clear_info=False, observe_id=False, standardise_rewards=True, wrappers=None)
    ref = MarlbaseEnv(NAME, 25, standardise_rewards=True)
    rng = np.random.default_rng(0)
    for ep in range(3):
        env.reset()
        ref.reset(DrawStream(env.cfg.seed, 0, ep))
        done = False
        while not done:
            acts = [int(a) for a in rng.choice(6, size=2, p=[0.05, 0.15, 0.15, 0.15, 0.15, 0.35])]
            o, r, d, tr, info = env.step(acts)
            o2, r2, d2, tr2, info2 = ref.step(acts)
            assert [np.float32(x) for x in r2] == [np.float32(x) for x in r] and (d, tr) == (d2, tr2)
            done = d or tr
        np.testing.assert_allclose(info["episode_returns"], info2["episode_returns"], rtol=1e-6)  # RAW returns
    # fused collector == modular path (same Philox streams), rewards included
    N, T = 64, 25
    spec = h.NetSpec(2, 15, 64, 6)
    from oracle import dqn_port as dp
    params = dp.init_params(2, 15, 64, 6, seed=2).cuda()
    outs = []
    for fused in (True, False):
        cfg = h.lbf_config(NAME, N, T, seed=9)
        stats = h.attach_reward_stats(cfg)
        rb = h.DeviceReplay(N, 2, 15, T)
        fr, fl = torch.zeros(2, N, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda")
        if fused:
            for rnd in range(2):
                h.idqn_collect(cfg, spec, params, 0.3, rnd, rb, 0, fr, fl)
        else:
            for rnd in range(2):
                envb = h.BatchedForaging(cfg)
                envb.episode.fill_(rnd)
                obs = envb.reset()
                slot = torch.arange(N, dtype=torch.int32, device="cuda")
                rb.init_episode(slot, obs)
                alive = torch.ones(N, dtype=torch.bool, device="cuda")
                for t in range(T):
                    acts = h.dqn_act(spec, params, obs, 0.3, seed=cfg.seed, episode=torch.full((N,), rnd, dtype=torch.int32, device="cuda"),
                                     ep_length=torch.full((N,), t, dtype=torch.int32, device="cuda"))
                    obs, rew, dn, tr = envb.step(acts, active=alive.to(torch.uint8), auto_reset=False)  # finished envs are not stepped
                    rb.add(slot, torch.full((N,), t, dtype=torch.int32, device="cuda"), obs, acts, rew, (dn | tr).to(torch.uint8),
                           active=alive.to(torch.uint8))
                    alive &= ~(dn | tr).bool()
        outs.append((rb.rew.clone(), rb.act.clone(), stats.clone()))
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2])
    assert float(outs[0][0].abs().sum()) > 0
    monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / "sr"))
    df = run.main(["+algorithm=idqn", f"env.name={NAME}", "env.time_limit=25", "env.parallel_envs=128", "env.standardise_rewards=True",
                   "algorithm.model.layers=[64,64]", "seed=2", "algorithm.total_steps=300000", "algorithm.eval_interval=100000",
                   "algorithm.eval_episodes=128", "algorithm.updates_per_round=16"])
    assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all()


def test_observe_id_option(tmp_path, monkeypatch):
    """env.observe_id=True (ObserveID, utils/wrappers.py:73-103): one-hot agent index in front of every observation - scalar
    env API vs the oracle's wrapper, fused collector == step-by-step path on the widened observations, learner parity on the
    widened shape, and the option through the driver together with parameter sharing (its usual companion)."""
    from codebase_amd import hip as h
    from codebase_amd import run
    from codebase_amd.utils.envs import make_env
    from oracle import dqn_port as dp

    env = make_env(seed=5, name=NAME, time_limit=25, clear_info=False, observe_id=True, standardise_rewards=False, wrappers=None)
    assert env.observation_space[0].shape == (17,)
    ref = MarlbaseEnv(NAME, 25, observe_id=True)
    rng = np.random.default_rng(0)
    for ep in range(2):
        o, _ = env.reset()
        o2, _ = ref.reset(DrawStream(env.cfg.seed, 0, ep))
        done = False
        while not done:
            for a, b in zip(o, o2):
                np.testing.assert_array_equal(a, b)
            acts = [int(a) for a in rng.integers(0, 6, 2)]
            o, r, d, tr, info = env.step(acts)
            o2, r2, d2, tr2, info2 = ref.step(acts)
            done = d or tr
    # fused collector == modular path on D + P = 17 inputs
    N, T, P, D = 64, 25, 2, 17
    spec = h.NetSpec(P, D, 64, 6)
    params = dp.init_params(P, D, 64, 6, seed=2).cuda()
    outs = []
    for fused in (True, False):
        cfg = h.lbf_config(NAME, N, T, seed=9, observe_id=1)
        rb = h.DeviceReplay(N, P, D, T)
        fr, fl = torch.zeros(P, N, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda")
        if fused:
            h.idqn_collect(cfg, spec, params, 0.3, 0, rb, 0, fr, fl)
        else:
            envb = h.BatchedForaging(cfg)
            obs = envb.reset()
            slot = torch.arange(N, dtype=torch.int32, device="cuda")
            rb.init_episode(slot, obs)
            alive = torch.ones(N, dtype=torch.bool, device="cuda")
            zero = torch.zeros(N, dtype=torch.int32, device="cuda")
            for t in range(T):
                acts = h.dqn_act(spec, params, obs, 0.3, seed=cfg.seed, episode=zero, ep_length=torch.full((N,), t, dtype=torch.int32, device="cuda"))
                obs, rew, dn, tr = envb.step(acts, active=alive.to(torch.uint8), auto_reset=False)
                rb.add(slot, torch.full((N,), t, dtype=torch.int32, device="cuda"), obs, acts, rew, (dn | tr).to(torch.uint8),
                       active=alive.to(torch.uint8))
                alive &= ~(dn | tr).bool()
        outs.append((rb.obs.clone(), rb.act.clone(), rb.rew.clone()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert torch.equal(outs[0][0][:, 0, 0, :2], torch.tensor([1.0, 0.0], device="cuda").expand(N, 2))  # agent 0's id prefix
    assert torch.equal(outs[0][0][:, 1, 0, :2], torch.tensor([0.0, 1.0], device="cuda").expand(N, 2))
    # learner on the widened shape, both widths
    for H in (64, 128):
        sp = h.NetSpec(P, D, H, 6)
        pr0 = dp.init_params(P, D, H, 6, seed=1) + 0.05
        tg0 = dp.init_params(P, D, H, 6, seed=3)
        batch = dp.synthetic_batch(P, T, 40, D, 6, seed=5)
        pr = pr0.clone().requires_grad_(True)
        want = dp.compute_loss(pr, tg0, batch, 0.99, True, D, H, 6)
        want.backward()
        up = h.DqnUpdater(sp, pr0.cuda(), tg0.cuda())
        loss, grad = up.loss_grad(h.Batch(*(batch[k].cuda() for k in ("obss", "actions", "rewards", "dones", "filled")), None))
        assert abs(loss.cpu().numpy()[0] - want.item()) <= 3e-5 * abs(want.item())
        np.testing.assert_allclose(grad.cpu().numpy(), pr.grad.numpy(), rtol=3e-4, atol=3e-5 * max(1.0, float(pr.grad.abs().max())))
    monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / "oid"))
    df = run.main(["+algorithm=idqn", f"env.name={NAME}", "env.time_limit=25", "env.parallel_envs=128", "env.observe_id=True",
                   "algorithm.model.parameter_sharing=True", "algorithm.model.layers=[64,64]", "seed=2", "algorithm.total_steps=300000",
                   "algorithm.eval_interval=100000", "algorithm.eval_episodes=128", "algorithm.updates_per_round=16"])
    assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all()
    monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / "oid_ac"))
    df = run.main(["+algorithm=ia2c", f"env.name={NAME}", "env.time_limit=25", "env.parallel_envs=128", "env.observe_id=True",
                   "algorithm.model.actor.layers=[64,64]", "algorithm.model.critic.layers=[64,64]", "algorithm.model.actor.parameter_sharing=True",
                   "algorithm.model.critic.parameter_sharing=True", "seed=2", "algorithm.total_steps=60000", "algorithm.eval_interval=20000"])
    assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all()
