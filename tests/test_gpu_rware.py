"""GPU parity of the multi-robot warehouse path (config 4) through the C-ABI: batched env kernels, both fused collectors
and the 71-wide / 5-action learner shapes, against oracle/rware.py (parity unpinned: the rware package is absent - see
that file's header) and the oracle ports of the learners.  Bit-exact: state records, observations, rewards, done flags,
stored batches; fp32 learners to the bars of test_gpu_parity.py / test_gpu_ac_update.py."""
import numpy as np
import pytest
import torch

from oracle import ac_update_port as ap
from oracle import dqn_port as dp
from oracle import rware as rw
from oracle.ac_port import OracleVecEnv, sample_inverse_cdf, step_uniforms
from oracle.lbf import MarlbaseEnv
from oracle.philox import DrawStream
from tests.test_gpu_ac_update import assert_grad_close, dev_ac_batch
from tests.test_gpu_parity import DEV, dev_batch, hip
from tests.test_rware_oracle import pack_state

pytestmark = pytest.mark.gpu
TINY4 = "rware:rware-tiny-4ag-v2"


@pytest.mark.parametrize("name,over,coop", [(TINY4, {}, False), ("rware:rware-tiny-2ag-easy-v2", {}, True),
                                            ("rware:rware-small-4ag-hard-v2", {"reward_type": rw.REWARD_TWO_STAGE}, False),
                                            ("rware:rware-tiny-8ag-v2", {"reward_type": rw.REWARD_GLOBAL, "max_inactivity_steps": 60}, False)])
def test_env_reset_step_bit_exact_vs_oracle(name, over, coop):
    h = hip()
    N, T, seed = 70, 90, 99
    cfg = h.rware_config(name, N, T, seed=seed, cooperative=coop, **over)
    env = h.BatchedForaging(cfg)
    P = env.P
    orc = [MarlbaseEnv(name, T, cooperative=coop, **over) for _ in range(N)]
    R, C = orc[0].env.grid_size
    assert (env.rows, env.cols) == (R, C) and env.D == 71 and env.A == 5
    rng = np.random.default_rng(3)
    delivered = 0
    for episode in range(2):
        obs = env.reset().cpu().numpy()
        state = env.state.cpu().numpy()
        for n, e in enumerate(orc):
            o, _ = e.reset(DrawStream(seed, n, episode))
            assert env.n_shelves == len(e.env.shelfs)
            np.testing.assert_array_equal(pack_state(e.env, P), state[n])
            for p in range(P):
                np.testing.assert_array_equal(o[p], obs[p, n])
        if episode == 1:  # seed deliveries: a requested shelf on agent 0, one step above a goal, facing it
            for n, e in enumerate(orc):
                w = e.env
                if any((b.x, b.y) == (C // 2 - 1, R - 2) for b in w.agents[1:]):
                    continue
                a, sh = w.agents[0], w.request_queue[n % len(w.request_queue)]
                a.x, a.y, a.dir, a.carrying_shelf = C // 2 - 1, R - 2, rw.DOWN, sh
                sh.x, sh.y = a.x, a.y
                w._recalc_grid()
                state[n] = pack_state(w, P)
            env.state.copy_(torch.tensor(state))
        alive = np.ones(N, bool)
        for t in range(T):
            acts = rng.choice(5, size=(P, N), p=[0.1, 0.5, 0.1, 0.1, 0.2]).astype(np.int32)
            if episode == 1 and t == 0:
                acts[0, :] = 1
            active = torch.tensor(alive.astype(np.uint8), device=DEV)
            o_d, r_d, d_d, tr_d = (x.cpu().numpy() for x in env.step(torch.tensor(acts, device=DEV), active=active))
            state = env.state.cpu().numpy()
            finr, finl = env.fin_return.cpu().numpy(), env.fin_length.cpu().numpy()
            for n in range(N):
                if not alive[n]:
                    continue
                o, r, d, tr, info = orc[n].step([int(a) for a in acts[:, n]])
                np.testing.assert_array_equal(pack_state(orc[n].env, P), state[n], err_msg=f"env {n} step {t}")
                for p in range(P):
                    np.testing.assert_array_equal(o[p], o_d[p, n])
                np.testing.assert_array_equal(np.array(r, dtype=np.float32), r_d[:, n])
                assert bool(d_d[n]) == d and bool(tr_d[n]) == tr
                delivered += sum(r) > 0
                if d or tr:
                    alive[n] = False
                    np.testing.assert_array_equal(info["episode_returns"].astype(np.float32), finr[:, n])
                    assert info["episode_length"] == finl[n]
            if not alive.any():
                break
        assert not alive.any()
    assert delivered > 0


@pytest.mark.parametrize("name,H", [(TINY4, 64), (TINY4, 128), ("rware:rware-tiny-2ag-v2", 64), ("rware:rware-small-4ag-v2", 64),
                                    ("rware:rware-tiny-8ag-hard-v2", 64), ("rware:rware-large-2ag-easy-v2", 128)])
def test_fused_ac_collector_matches_oracle(name, H):
    """the fused rollout collector on the warehouse env == the oracle vector env driven by the kernel's own logits
    through the restated inverse-CDF sampler (same check as tests/test_ac_collector.py does for Level-Based Foraging)"""
    from codebase_amd.ac.train import ActorNetworks, _collect_trajectories
    from codebase_amd.utils.envs import make_env

    N, T, seed, rnd = 40, 30, 77, 2
    torch.manual_seed(5)
    over = {"max_steps": 24}  # some envs end on the env's own limit before the TimeLimit
    envs = make_env(seed=seed, name=name, time_limit=T, parallel_envs=N, **over)
    model = ActorNetworks(envs.single_observation_space, envs.single_action_space, [H, H])
    assert model.spec.obs_dim == 71 and model.spec.n_actions == 5
    model.actor_params.mul_(3.0)
    P = envs.n_agents
    t, batch, infos = _collect_trajectories(envs, model, T, N, P, "cuda", False, round_idx=rnd)
    vec = OracleVecEnv(name, N, T, seed, **over)
    vec.set_episode(2 * rnd)
    b_act = batch.actions.cpu().numpy()
    mism = [0, 0]

    def act_fn(obss, step):
        logits = model.logits(torch.tensor(np.stack(obss), device="cuda")).cpu().numpy()
        acts = np.zeros((N, P), np.int64)
        for n in range(N):
            us = step_uniforms(seed, n, 2 * rnd, step, P)
            for p in range(P):
                acts[n, p] = sample_inverse_cdf(logits[p, n], us[p])
                mism[1] += 1
                if step < b_act.shape[0] and acts[n, p] != b_act[step, n, p] and batch.filled[step, n] > 0:
                    mism[0] += 1  # libm vs device expf, an ulp at a CDF boundary: follow the kernel
                    acts[n, p] = b_act[step, n, p]
        return acts

    from oracle.ac_port import collect_trajectories

    t_o, ob, _ = collect_trajectories(vec, act_fn, T)
    assert t == t_o and mism[0] <= 0.002 * mism[1]
    np.testing.assert_array_equal(batch.obss.cpu().numpy(), ob["obss"])
    np.testing.assert_array_equal(b_act, ob["actions"])
    np.testing.assert_array_equal(batch.rewards.cpu().numpy(), ob["rewards"])
    np.testing.assert_array_equal(batch.dones.cpu().numpy().astype(bool), ob["dones"])
    np.testing.assert_array_equal(batch.filled.cpu().numpy(), ob["filled"])
    assert int(ob["filled"].sum(0).min()) == 24


@pytest.mark.parametrize("H", [64, 128])
def test_fused_dqn_collector_replays_through_oracle(H):
    h = hip()
    N, T, seed, rnd, eps = 48, 40, 4242, 3, 0.4
    cfg = h.rware_config(TINY4, N, T, seed=seed)
    P, D, A = 4, 71, 5
    spec = h.NetSpec(P, D, H, A)
    params = (dp.init_params(P, D, H, A, seed=1) * 3.0).to(DEV)
    CAP = 64
    rb = h.DeviceReplay(CAP, P, D, T)
    finr = torch.zeros(P, N, device=DEV)
    finl = torch.zeros(N, dtype=torch.int32, device=DEV)
    h.idqn_collect(cfg, spec, params, eps, rnd, rb, 40, finr, finl)
    ro, ra, rr, rd, rf = (x.cpu().numpy() for x in (rb.obs, rb.act, rb.rew, rb.done, rb.filled))
    fl, fr = finl.cpu().numpy(), finr.cpu().numpy()
    # greedy steps agree with the modular act kernel on the stored observations
    q = torch.empty(P, N, A, device=DEV)
    for n in range(0, N, 2):
        s = (40 + n) % CAP
        e = MarlbaseEnv(TINY4, T)
        o, _ = e.reset(DrawStream(seed, n, rnd))
        for p in range(P):
            np.testing.assert_array_equal(o[p], ro[s, p, 0])
        assert fl[n] == T and rf[s, :T].all()
        for t in range(T):
            o, r, d, tr, info = e.step([int(a) for a in ra[s, :, t]])
            for p in range(P):
                np.testing.assert_array_equal(o[p], ro[s, p, t + 1])
            np.testing.assert_array_equal(np.array(r, dtype=np.float32), rr[s, :, t])
            assert rd[s, t + 1] == int(d or tr)
        assert tr and not d
        np.testing.assert_array_equal(info["episode_returns"].astype(np.float32), fr[:, n])
    obs0 = rb.obs[(40 + torch.arange(N)) % CAP][:, :, 0].permute(1, 0, 2).contiguous()  # [P][N][D]
    greedy = h.dqn_act(spec, params, obs0, 0.0, u=torch.ones(N, device=DEV), rand_actions=torch.zeros(P, N, dtype=torch.int32, device=DEV), q_out=q)
    from oracle.philox import act_noise

    for n in range(N):
        u, rnd_a = act_noise(seed, n, rnd, 0, P, A)
        s = (40 + n) % CAP
        expect = rnd_a if eps > u else greedy[:, n].cpu().tolist()
        assert ra[s, :, 0].tolist() == expect


@pytest.mark.parametrize("H,mode", [(64, "idqn"), (128, "idqn"), (64, "vdn"), (128, "vdn")])
def test_dqn_learner_on_warehouse_shapes_vs_torch_port(H, mode):
    h = hip()
    P, T, B, D, A = 4, 9, 37, 71, 5
    spec = h.NetSpec(P, D, H, A)
    params = dp.init_params(P, D, H, A, seed=1) + 0.03
    target = dp.init_params(P, D, H, A, seed=3)
    batch = dp.synthetic_batch(P, T, B, D, A, seed=5)
    pr = params.clone().requires_grad_(True)
    ref = dp.compute_loss(pr, target, batch, 0.99, True, D, H, A, mode=mode)
    ref.backward()
    up = h.DqnUpdater(spec, params.to(DEV), target.to(DEV))
    loss, grad = up.loss_grad(dev_batch(h, batch), mode=1 if mode == "vdn" else 0)
    assert abs(loss.cpu().numpy()[0] - ref.item()) <= 3e-5 * max(abs(ref.item()), 1e-3)
    gref = pr.grad.numpy()
    np.testing.assert_allclose(grad.cpu().numpy(), gref, rtol=3e-4, atol=3e-5 * max(1.0, np.abs(gref).max()))


@pytest.mark.parametrize("P,H", [(4, 64), (2, 128)])
def test_qmix_on_warehouse_shapes_vs_oracle_port(P, H):
    from oracle import qmix_port as qp
    from tests.test_gpu_qmix import assert_grad_close as qclose

    h = hip()
    T, B, D, A = 7, 35, 71, 5
    spec = h.NetSpec(P, D, H, A)
    params = dp.init_params(P, D, H, A, seed=1) + 0.05
    target = dp.init_params(P, D, H, A, seed=3)
    mixer, tmixer = qp.mixer_init(P, P * D, seed=11), qp.mixer_init(P, P * D, seed=12)
    batch = dp.synthetic_batch(P, T, B, D, A, seed=5)
    batch["rewards"][1:] = batch["rewards"][0]
    batch["obss"] = batch["obss"] * 0.25
    pr, mr = params.clone().requires_grad_(True), mixer.clone().requires_grad_(True)
    ref = qp.compute_loss(pr, target, mr, tmixer, batch, 0.99, True, D, H, A)
    ref.backward()
    up = h.QmixUpdater(spec, params.to(DEV), target.to(DEV), mixer.to(DEV), tmixer.to(DEV))
    loss, grad = up.loss_grad(dev_batch(h, batch))
    assert abs(loss.cpu().numpy()[0] - ref.item()) <= 3e-5 * abs(ref.item())
    qclose(grad.cpu().numpy(), pr.grad.numpy(), 3e-4)
    qclose(up.mixer_grad.cpu().numpy(), mr.grad.numpy(), 3e-4)


@pytest.mark.parametrize("H", [64, 128])
def test_a2c_and_ppo_on_warehouse_shapes_vs_oracle_port(H):
    h = hip()
    P, T, N, D, A, n = 4, 30, 21, 71, 5, 5
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
    # PPO: 2 epochs against the port's learner
    lr = ap.Learner(actor, critic, D, H, A, gamma=0.97, n_steps=n, entropy_coef=0.01, value_loss_coef=0.5, num_epochs=2, ppo_clip=0.2)
    mref = lr.update(batch, 1)
    up = h.AcUpdater(spec, torch.cat([actor.reshape(-1), critic.reshape(-1)]).to(DEV), critic.to(DEV).contiguous(), gamma=0.97, n_steps=n,
                     entropy_coef=0.01, value_loss_coef=0.5, ppo_clip=0.2)
    bt = dev_ac_batch(batch)
    up.ppo_prepare(bt)
    acc = np.zeros(4)
    for _ in range(2):
        acc += up.ppo_loss_grad(bt).cpu().numpy()[:4]
        up.apply()
    np.testing.assert_allclose(acc / 2, [mref["loss"], mref["actor_loss"], mref["value_loss"], mref["entropy"]], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(up.block[:actor.numel()].cpu().numpy(), lr.actor().detach().reshape(-1).numpy(), rtol=0, atol=5e-6)


def test_ia2c_idqn_and_qmix_on_the_warehouse_end_to_end(tmp_path, monkeypatch):
    """config 4's algorithm / env pair through the drop-in surface: run.py +algorithm=ia2c env.name=rware:... (and IDQN)"""
    from codebase_amd import run

    for algo, extra in (("ia2c", []), ("idqn", ["algorithm.model.layers=[64,64]"]), ("qmix", ["algorithm.model.layers=[64,64]"])):
        monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / algo))
        df = run.main([f"+algorithm={algo}", f"env.name={TINY4}", "env.time_limit=50", "env.parallel_envs=128", "seed=1",
                       "algorithm.total_steps=60000", "algorithm.eval_interval=20000"] + extra)
        assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all() and np.isfinite(df["mean_episode_returns"]).all()


def test_warehouse_with_sharing_and_standardisation_end_to_end(tmp_path, monkeypatch):
    """the wrappers and learner options that sit around the env are env-agnostic: parameter sharing, standardise_returns and
    env.standardise_rewards on the warehouse (scalar env API: the wrapper's rewards against the oracle's, bit for bit)"""
    from codebase_amd import run
    from codebase_amd.utils.envs import make_env

    monkeypatch.setenv("MARLHIP_RUN_DIR", str(tmp_path / "ia2c_shared"))
    df = run.main(["+algorithm=ia2c", f"env.name={TINY4}", "env.time_limit=50", "env.parallel_envs=128", "seed=1", "env.standardise_rewards=True",
                   "algorithm.total_steps=60000", "algorithm.eval_interval=20000", "algorithm.standardise_returns=True",
                   "algorithm.model.actor.parameter_sharing=True", "algorithm.model.critic.parameter_sharing=True"])
    assert df.shape[0] >= 2 and np.isfinite(df["loss"]).all()
    env = make_env(seed=5, name="rware:rware-tiny-2ag-v2", time_limit=40, standardise_rewards=True, wrappers=["CooperativeReward"])
    orc = MarlbaseEnv("rware:rware-tiny-2ag-v2", 40, cooperative=True, standardise_rewards=True)
    for episode in range(2):
        o_h, _ = env.reset()
        o_o, _ = orc.reset(DrawStream(5, 0, episode))
        if episode == 1:  # one delivery so that the streaming statistics see a non-zero reward
            w = orc.env
            a, sh = w.agents[0], w.request_queue[0]
            a.x, a.y, a.dir, a.carrying_shelf = 4, 9, rw.DOWN, sh
            if (w.agents[1].x, w.agents[1].y) == (4, 9):
                w.agents[1].x = 0
            sh.x, sh.y = 4, 9
            w._recalc_grid()
            env.batched.state.copy_(torch.tensor(pack_state(w, 2))[None])
        rng = np.random.default_rng(episode)
        for t in range(40):
            acts = [1, int(rng.integers(5))] if (episode == 1 and t == 0) else [int(a) for a in rng.integers(0, 5, 2)]
            oh, rh, dh, th, _ = env.step(acts)
            oo, ro, do, to, _ = orc.step(acts)
            for p in range(2):
                np.testing.assert_array_equal(oh[p], oo[p])
            np.testing.assert_array_equal(np.array(rh, np.float32), np.array(ro, np.float32))
            assert (dh, th) == (do, to)
