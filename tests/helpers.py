"""Shared test plumbing: env configs, packed-state bridge between oracle.lbf and the
byte records the HIP env keeps in HBM, and the g++ host shim of csrc/lbf_core.h."""
import ctypes
import os
import subprocess

import numpy as np

from oracle.lbf import MarlbaseEnv, parse_env_name
from oracle.philox import DrawStream

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CFG_FIELDS = [
    "n_envs", "n_agents", "n_food", "rows", "cols", "sight", "max_episode_steps", "time_limit",
    "force_coop", "min_player_level", "max_player_level", "min_food_level", "max_food_level",
    "normalize_reward", "cooperative",
]


def lbf_cfg(name, n_envs, time_limit=25, seed=0, cooperative=False, **over):
    kw = parse_env_name(name)
    kw.update(over)
    return dict(
        n_envs=n_envs, n_agents=kw["players"], n_food=kw["max_num_food"], rows=kw["field_size"][0],
        cols=kw["field_size"][1], sight=kw["sight"], max_episode_steps=kw["max_episode_steps"],
        time_limit=time_limit, force_coop=int(kw["force_coop"]), min_player_level=kw["min_player_level"],
        max_player_level=kw["max_player_level"], min_food_level=kw.get("min_food_level", 1),
        max_food_level=kw.get("max_food_level") or 0, normalize_reward=int(kw.get("normalize_reward", True)),
        cooperative=int(cooperative), penalty=float(kw["penalty"]), seed=int(seed),
    )


def oracle_env(name, cfg, rng=None):
    kw = {}
    return MarlbaseEnv(name, cfg["time_limit"], cooperative=bool(cfg["cooperative"]), rng=rng, **kw)


def stride(P, F):
    return (3 * F + 3 * P + 4 + 3) & ~3


def pack_state(env):
    """oracle ForagingEnv -> packed record bytes"""
    foods, players, step, spawned = env.get_state()
    P, F = len(players), len(foods)
    rec = np.zeros(stride(P, F), np.uint8)
    rec[: 3 * F] = foods.reshape(-1)
    rec[3 * F : 3 * F + 3 * P] = players.reshape(-1)
    rec[3 * F + 3 * P] = step & 0xFF
    rec[3 * F + 3 * P + 1] = step >> 8
    rec[3 * F + 3 * P + 2] = spawned & 0xFF
    rec[3 * F + 3 * P + 3] = spawned >> 8
    return rec


def unpack_state(rec, P, F):
    foods = rec[: 3 * F].reshape(F, 3).astype(np.int32)
    players = rec[3 * F : 3 * F + 3 * P].reshape(P, 3).astype(np.int32)
    t = rec[3 * F + 3 * P :]
    return foods, players, int(t[0]) | (int(t[1]) << 8), int(t[2]) | (int(t[3]) << 8)


class HostCfg(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in CFG_FIELDS] + [("penalty", ctypes.c_double), ("seed", ctypes.c_uint64),
                                                               ("reward_stats", ctypes.c_void_p)]


def host_cfg(cfg):
    return HostCfg(**{k: cfg[k] for k in CFG_FIELDS}, penalty=cfg["penalty"], seed=cfg["seed"], reward_stats=cfg.get("reward_stats"))


_shim = None


def host_shim():
    """g++ build of tests/host_shim/lbf_host.cpp (the env core the kernels inline)."""
    global _shim
    if _shim is None:
        out = os.path.join(ROOT, "tests", "host_shim", "_lbf_host.so")
        src = os.path.join(ROOT, "tests", "host_shim", "lbf_host.cpp")
        deps = [src] + [os.path.join(ROOT, "codebase_amd", "csrc", f) for f in ("lbf_core.h", "philox.h", "rware_core.h")]
        if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-o", out])
        _shim = ctypes.CDLL(out)
    return _shim


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)
