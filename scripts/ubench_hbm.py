"""HBM-bound kernels at sizes that leave the caches (SURVEY.md 8d "scaled run"): batched env step at
2^22 envs (750 MB of state + outputs per step: past the 256 MB Infinity Cache, which 2^20 envs = 187 MB were not - argv[1] overrides),
the 8-player step kernel at 2^19 envs (720 MB), replay sample-gather of 65,536 episodes from a 1.25 x 2^20-episode (4.5 GB) replay, replay add.
Prints one JSON object: achieved GB/s = algorithmic bytes / measured time, against 8 TB/s."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codebase_amd import hip as h

NAME = "lbforaging:Foraging-8x8-2p-3f-v3"
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


out = {}


def step_row(name, n):
    cfg = h.lbf_config(name, n, 0, seed=1)  # no time limit: envs keep stepping
    env = h.BatchedForaging(cfg)
    env.reset()
    acts = torch.randint(0, 6, (env.P, n), dtype=torch.int32, device="cuda")
    dt = timed(lambda: env.step(acts), 20)
    per = 2 * env.stride + 4 * env.P + 4 * env.P * env.D + 4 * env.P + 2
    row = dict(env=name.split(":")[-1], n_envs=n, bytes_per_env_step=per, working_set_MB=per * n / 1e6, us=dt * 1e6, env_steps_per_s=n / dt,
               achieved_GBs=per * n / dt / 1e9, frac_of_8TBs=per * n / dt / 1e9 / PEAK)
    return row, env, acts


N_BIG = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
out["lbf_step_kernel"], env, acts = step_row(NAME, N_BIG)
del env, acts
torch.cuda.empty_cache()
out["lbf_step_kernel_8p5f"], env, acts = step_row("lbforaging:Foraging-15x15-8p-5f-v3", 1 << 19)
del env, acts
torch.cuda.empty_cache()
N = 1 << 20
out["lbf_step_kernel_2e20_cache_resident"], env, acts = step_row(NAME, N)  # the round-1/2 figure: 187 MB, inside the Infinity Cache
P, D = env.P, env.D

T, CAP, B = 25, 1310720, 65536  # 1.25 x 2^20 episodes = 4.5 GB of replay (>= 4 GB)
rb = h.DeviceReplay(CAP, P, D, T)
rb.obs.uniform_(-1, 7)
idx = torch.randint(0, CAP, (B,), dtype=torch.int32, device="cuda")
dt = timed(lambda: rb.sample(B, idx=idx), 10)
rd = 4 * P * D * (T + 1) + P * T * 5 + (T + 1) + T
wr = 4 * P * D * (T + 1) + 8 * P * T + 4 * P * T + 4 * (T + 1) + 4 * T
out["replay_sample_kernel"] = dict(batch=B, replay_episodes=CAP, replay_GB=CAP * rd / 1e9, read_B_per_episode=rd,
                                   write_B_per_episode=wr, us=dt * 1e6, episodes_per_s=B / dt,
                                   achieved_GBs=(rd + wr) * B / dt / 1e9, frac_of_8TBs=(rd + wr) * B / dt / 1e9 / PEAK)

slot = torch.randperm(CAP, device="cuda")[:N].to(torch.int32)
tt = torch.randint(0, T, (N,), dtype=torch.int32, device="cuda")
rew = torch.rand(P, N, device="cuda")
done = torch.zeros(N, dtype=torch.uint8, device="cuda")
dt = timed(lambda: rb.add(slot, tt, env.obs, acts, rew, done), 10)
per = 4 * P * D + 4 * P * D + 4 * P + P + 4 * P + 4 * P + 1 + 1 + 1 + 8  # read obs/act/rew/done/slot/t, write rows
out["replay_add_kernel"] = dict(n_envs=N, bytes_per_env_step=per, us=dt * 1e6, achieved_GBs=per * N / dt / 1e9,
                                frac_of_8TBs=per * N / dt / 1e9 / PEAK)
# what the collectors do: N consecutive slots of the ring (rows of consecutive envs a whole episode record apart, in order)
slot_seq = ((torch.arange(N, device="cuda") + 12345) % CAP).to(torch.int32)
tt_same = torch.full((N,), 7, dtype=torch.int32, device="cuda")
dt = timed(lambda: rb.add(slot_seq, tt_same, env.obs, acts, rew, done), 10)
out["replay_add_kernel_consecutive_slots"] = dict(n_envs=N, bytes_per_env_step=per, us=dt * 1e6, achieved_GBs=per * N / dt / 1e9,
                                                  frac_of_8TBs=per * N / dt / 1e9 / PEAK)
print(json.dumps(out))
