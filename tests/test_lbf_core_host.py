"""CPU: the env core the HIP kernels inline (csrc/lbf_core.h, built with g++) against the
oracle restatement of lbforaging (oracle/lbf.py) - reset (same Philox draw stream) and step
(same state, same joint action) must agree bit for bit: state records, observations,
fp32 rewards, done / truncated."""
import numpy as np
import pytest

from oracle.philox import DrawStream, philox4x32_10
from tests.helpers import host_cfg, host_shim, lbf_cfg, oracle_env, pack_state, ptr, stride

CASES = [
    ("lbforaging:Foraging-8x8-2p-3f-v3", False),
    ("lbforaging:Foraging-8x8-2p-2f-coop-v3", False),
    ("lbforaging:Foraging-10x10-3p-3f-v3", True),
    ("lbforaging:Foraging-15x15-4p-5f-v3", True),
    ("lbforaging:Foraging-15x15-8p-5f-v3", True),
    ("lbforaging:Foraging-2s-8x8-2p-3f-v3".replace("-2s-8x8", "-8x8") + "", False),
    ("lbforaging:Foraging-8x8-2p-3f-2s-v3", False),
    ("lbforaging:Foraging-15x15-4p-3f-2s-pen-v3", False),
    ("lbforaging:Foraging-5x5-3p-3f-v2", False),
]


def test_philox_known_answers():
    # Random123 kat_vectors (philox4x32-10)
    assert philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert philox4x32_10((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert philox4x32_10((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == (
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)
    lib = host_shim()
    rng = np.random.default_rng(0)
    for _ in range(50):
        ctr = rng.integers(0, 2**32, 4, dtype=np.uint32)
        key = rng.integers(0, 2**32, 2, dtype=np.uint32)
        out = np.zeros(4, np.uint32)
        lib.host_philox(ptr(ctr), ptr(key), ptr(out))
        assert tuple(int(x) for x in out) == philox4x32_10(ctr, key)


@pytest.mark.parametrize("name,coop", CASES)
def test_reset_and_step_match_oracle(name, coop):
    lib = host_shim()
    N = 24
    cfg = lbf_cfg(name, N, time_limit=25, seed=1234, cooperative=coop)
    P, F = cfg["n_agents"], cfg["n_food"]
    D = 3 * (P + F)
    S = stride(P, F)
    assert lib.host_lbf_stride(P, F) == S
    hc = host_cfg(cfg)
    rng = np.random.default_rng(7)
    for episode in range(3):
        state = np.zeros((N, S), np.uint8)
        obs = np.zeros((P, N, D), np.float32)
        epi = np.full(N, episode, np.uint32)
        assert lib.host_lbf_reset(ctypes_byref(hc), ptr(state), ptr(epi), ptr(obs)) == 0
        envs = []
        for n in range(N):
            e = oracle_env(name, cfg)
            o, _ = e.reset(DrawStream(cfg["seed"], n, episode))
            envs.append(e)
            np.testing.assert_array_equal(pack_state(e.env), state[n])
            for p in range(P):
                np.testing.assert_array_equal(o[p], obs[p, n])
        alive = np.ones(N, bool)
        for t in range(25):
            # bias towards LOAD so joint loads / failed loads are exercised
            acts = rng.choice(6, size=(P, N), p=[0.1, 0.15, 0.15, 0.15, 0.15, 0.3]).astype(np.int32)
            rew = np.zeros((P, N), np.float32)
            raw = np.zeros((P, N), np.float64)
            done = np.zeros(N, np.uint8)
            trunc = np.zeros(N, np.uint8)
            prev = state.copy()
            assert lib.host_lbf_step(ctypes_byref(hc), ptr(state), ptr(acts), ptr(obs), ptr(rew), ptr(raw), ptr(done), ptr(trunc)) == 0
            for n in range(N):
                if not alive[n]:
                    continue
                o, r, d, tr, info = envs[n].step([int(a) for a in acts[:, n]])
                np.testing.assert_array_equal(pack_state(envs[n].env), state[n], err_msg=f"state n={n} t={t} prev={prev[n]} a={acts[:, n]}")
                for p in range(P):
                    np.testing.assert_array_equal(o[p], obs[p, n])
                np.testing.assert_array_equal(np.array(r, dtype=np.float32), rew[:, n])
                assert bool(done[n]) == d and bool(trunc[n]) == tr
                if d or tr:
                    alive[n] = False


def ctypes_byref(x):
    import ctypes
    return ctypes.byref(x)


def test_standardise_reward_wrapper_matches_oracle():
    """env.standardise_rewards: the per-env streaming record of csrc/lbf_core.h against the oracle's restatement of
    StandardiseReward (utils/wrappers.py:111-142), alone and under CooperativeReward, across episode boundaries"""
    lib = host_shim()
    for name, coop in (("lbforaging:Foraging-8x8-2p-3f-v3", False), ("lbforaging:Foraging-10x10-3p-3f-v3", True)):
        N = 6
        cfg = lbf_cfg(name, N, time_limit=25, seed=77, cooperative=coop)
        P, F = cfg["n_agents"], cfg["n_food"]
        D, S = 3 * (P + F), stride(P, F)
        stats = np.zeros((N, 3 * P + 1), np.float32)
        cfg["reward_stats"] = stats.ctypes.data
        hc = host_cfg(cfg)
        envs = [oracle_env(name, cfg) for _ in range(N)]
        for e in envs:
            e.standardise_rewards = True
        rng = np.random.default_rng(3)
        nonzero = 0
        for episode in range(3):
            state = np.zeros((N, S), np.uint8)
            obs = np.zeros((P, N, D), np.float32)
            epi = np.full(N, episode, np.uint32)
            assert lib.host_lbf_reset(ctypes_byref(hc), ptr(state), ptr(epi), ptr(obs)) == 0
            for n, e in enumerate(envs):
                e.reset(DrawStream(cfg["seed"], n, episode))
            alive = np.ones(N, bool)
            for t in range(25):
                acts = rng.choice(6, size=(P, N), p=[0.05, 0.15, 0.15, 0.15, 0.15, 0.35]).astype(np.int32)
                rew, raw = np.zeros((P, N), np.float32), np.zeros((P, N), np.float64)
                done, trunc = np.zeros(N, np.uint8), np.zeros(N, np.uint8)
                before = stats.copy()
                assert lib.host_lbf_step(ctypes_byref(hc), ptr(state), ptr(acts), ptr(obs), ptr(rew), ptr(raw), ptr(done), ptr(trunc)) == 0
                for n in range(N):
                    if not alive[n]:
                        stats[n] = before[n]  # the shim steps finished envs too; the reference env is not stepped after its episode
                        continue
                    o, r, d, tr, info = envs[n].step([int(a) for a in acts[:, n]])
                    np.testing.assert_array_equal(np.array(r, dtype=np.float32), rew[:, n])
                    np.testing.assert_array_equal(stats[n, :P], envs[n].sr_sumw)
                    np.testing.assert_array_equal(stats[n, P:2 * P], envs[n].sr_wmean)
                    np.testing.assert_array_equal(stats[n, 2 * P:3 * P], envs[n].sr_t)
                    assert stats[n, 3 * P:].view(np.int32)[0] == envs[n].sr_n
                    nonzero += int(np.any(np.abs(rew[:, n]) > 0))
                    if d or tr:
                        alive[n] = False
        assert nonzero > 20  # standardised rewards are non-zero on most steps once food has been eaten
