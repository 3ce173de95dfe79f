"""The fused collectors come in three wave organisations (csrc/collect_kernels.h, csrc/ac_collect_kernels.h): one wave per block of 16
envs running every agent's network in turn; NW waves per block, each with its own copy of the env state and a share of the
agents (chosen up to 8192 envs); and, for 2 agents with LDS-resident packs on launches that would still leave SIMDs idle, TWO waves per
agent (mlp_forward_h2: hidden tiles split, layer 3 one chain handed from wave to wave).  All must write the same bytes.  The choice is
read once per process (MARLHIP_COL_NW / MARLHIP_ACOL_NW, MARLHIP_COL_HS / MARLHIP_ACOL_HS), so each variant runs in its own interpreter
and reports digests of everything it wrote."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import hashlib, json, sys
import torch
sys.path.insert(0, %(root)r)
from codebase_amd import hip as h
from codebase_amd.ac.train import ActorNetworks, _collect_trajectories
from codebase_amd.dqn.model import QNetwork
from codebase_amd.utils.envs import make_env, _space_pair

def dig(*ts):
    m = hashlib.sha256()
    for t in ts:
        m.update(t.detach().cpu().contiguous().numpy().tobytes())
    return m.hexdigest()

out = {}
for name, T, N, H in (("lbforaging:Foraging-8x8-2p-3f-v3", 25, 200, 64), ("lbforaging:Foraging-8x8-2p-3f-v3", 25, 136, 128),
                      ("lbforaging:Foraging-15x15-4p-5f-v3", 25, 72, 128),
                      ("rware:rware-tiny-4ag-v2", 40, 100, 128), ("rware:rware-tiny-2ag-v2", 40, 100, 64)):
    torch.manual_seed(3)
    # IDQN collector: replay contents + episode statistics
    cfg = h.env_config(name, N, T, seed=11)
    obs_space, act_space = _space_pair(cfg)
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False, target_update_interval_or_tau=200)
    q = QNetwork(obs_space, act_space, hyper, [H, H], False, False, True, "cuda")
    rep = h.DeviceReplay(2 * N, q.n_agents, q.spec.obs_dim, T, device="cuda")
    fr = torch.zeros(q.n_agents, N, device="cuda"); fl = torch.zeros(N, dtype=torch.int32, device="cuda")
    for rnd in range(2):
        h.idqn_collect(cfg, q.spec, q.params, 0.3, rnd, rep, rnd * N, fr, fl, write_replay=True, clear_stale=True)
    out["idqn:%%d:" %% H + name] = dig(rep.obs, rep.act, rep.rew, rep.done, rep.filled, fr, fl)
    # actor-critic collector: the rollout batch + statistics
    envs = make_env(seed=5, name=name, time_limit=T, parallel_envs=N)
    actors = ActorNetworks(envs.single_observation_space, envs.single_action_space, [H, H])
    t, batch, infos = _collect_trajectories(envs, actors, T, N, q.n_agents, "cuda", False, round_idx=1)
    out["ac:%%d:" %% H + name] = dig(batch.obss, batch.actions, batch.rewards, batch.dones.to(torch.uint8), batch.filled) + ":%%d" %% t
print("DIGESTS " + json.dumps(out))
"""


def run_variant(nw, hs=None):
    env = dict(os.environ)
    for k in ("MARLHIP_COL_HS", "MARLHIP_ACOL_HS"):
        env.pop(k, None)
        if hs:
            env[k] = str(hs)
    if nw:
        env["MARLHIP_COL_NW"] = env["MARLHIP_ACOL_NW"] = str(nw)
    else:
        env.pop("MARLHIP_COL_NW", None)
        env.pop("MARLHIP_ACOL_NW", None)
    r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGESTS ")][-1]
    return json.loads(line[len("DIGESTS "):])


def test_one_wave_and_agent_per_wave_collectors_write_the_same_bytes():
    # 0: the default choice (agent-per-wave at these env counts; two waves per agent for the 2-agent LBF cases, hidden 64 and 128); hs=1: never two per agent
    one, split, split1 = run_variant(1), run_variant(0), run_variant(0, hs=1)
    assert one.keys() == split.keys() == split1.keys() and len(one) == 10
    for k in one:
        assert one[k] == split[k] == split1[k], k
