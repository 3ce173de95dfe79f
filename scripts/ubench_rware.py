"""Batched warehouse env step (rw_step_kernel) on its own: time per launch and achieved GB/s at the config-4 share
(2048 envs), a full GPU (16384) and a cache-busting size (2^20).  Algorithmic bytes per env-step: agent / queue /
counter bytes of the record read and written (2 * (7P + 4)), the grid bytes touched (<= 9 window cells per agent read,
<= 2 per moving carrier written: counted as 9P + 2P), actions, observations, rewards, flags."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codebase_amd import hip as h

NAME = sys.argv[1] if len(sys.argv) > 1 else "rware:rware-tiny-4ag-v2"
PEAK = 8000.0


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters


out = {"env": NAME}
for N in (2048, 16384, 1 << 20):
    cfg = h.rware_config(NAME, N, 0, seed=1, max_steps=0)  # no limits: envs keep stepping
    env = h.BatchedForaging(cfg)
    env.reset()
    P, D = env.P, env.D
    acts = torch.randint(0, 5, (P, N), dtype=torch.int32, device="cuda")
    dt = timed(lambda: env.step(acts), 50 if N < (1 << 20) else 10)
    per = 2 * (7 * P + 4) + 11 * P + 4 * P + 4 * P * D + 4 * P + 2
    out[f"rw_step_kernel_{N}"] = dict(n_envs=N, bytes_per_env_step=per, state_stride=env.stride, us=dt * 1e6, env_steps_per_s=N / dt,
                                      achieved_GBs=per * N / dt / 1e9, frac_of_8TBs=per * N / dt / 1e9 / PEAK)
print(json.dumps(out))
