"""In-kernel region counters of ac_collect_kernel on the warehouse (MARL_ACOL_PROF build, scripts/build_variants.py
acolprof:-DMARL_ACOL_PROF=1:rware_collect.hip; run with MARLHIP_LIB=codebase_amd/csrc/variants/libmarlhip_acolprof.so):
shader cycles per wave and per step in each region of a rollout step.

    python scripts/prof_ac_collect.py [hidden=64] [envs=2048] [env name]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codebase_amd import hip as h
from codebase_amd._lib import lib
from codebase_amd.ac.model import A2CNetwork
from codebase_amd.utils.envs import _space_pair

H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
NAME = sys.argv[3] if len(sys.argv) > 3 else "rware:rware-tiny-4ag-v2"
T = 500 if NAME.startswith("rware") else 25
REGIONS = ["actor forward", "Philox + sample", "action swap (barrier)", "env step", "rewards / bookkeeping / auto-reset", "observation",
           "batch stores (+ idle rows)", "[kernel total]"]
cfg = h.env_config(NAME, N, T, seed=1)
P, (D, A) = cfg.n_agents, h.env_dims(cfg)
torch.manual_seed(0)
obs_space, act_space = _space_pair(cfg)
hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=False, n_steps=5, entropy_coef=0.001, value_loss_coef=0.5,
             standardise_returns=False, target_update_interval_or_tau=200)
net = dict(layers=[H, H], parameter_sharing=False, use_orthogonal_init=True, use_rnn=False)
model = A2CNetwork(obs_space, act_space, hyper, net, dict(net, centralised=False), "cuda")
dev = model.device
b_obs = torch.empty(T + 1, N, P * D, device=dev)
b_act = torch.empty(T, N, P, dtype=torch.int64, device=dev)
b_rew = torch.empty(T, N, P, device=dev)
b_done = torch.empty(T + 1, N, dtype=torch.uint8, device=dev)
b_fill = torch.empty(T, N, device=dev)
fin_ret = torch.zeros(P, N, device=dev)
fin_len = torch.zeros(N, dtype=torch.int32, device=dev)
t_max = torch.zeros(1, dtype=torch.int32, device=dev)
fn = getattr(lib, "marlhip_debug_acol_prof" if NAME.startswith("rware") else "marlhip_debug_acol_prof_lbf", None)
out = (ctypes.c_ulonglong * 16)()
for r in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    h.ac_collect(cfg, model.spec, model.actor_params, r, T, False, b_obs, b_act, b_rew, b_done, b_fill, fin_ret, fin_len, t_max)
    b.record()
    torch.cuda.synchronize()
    print(f"round {r}: {a.elapsed_time(b) * 1e3:.1f} us, t_max {int(t_max.item())}, stored steps {int(b_fill.sum().item())}")
    if fn is not None and fn(out) == 0 and out[8]:
        waves, steps = out[8], int(t_max.item())
        for k, name in list(enumerate(REGIONS)) + [(9, "[pack staging]"), (10, "[first reset + observation + row 0]")]:
            print(f"   {name:38s} {out[k] / waves:12.0f} cycles / wave   {out[k] / waves / steps:9.1f} / step   {100.0 * out[k] / out[7]:5.1f} %")
