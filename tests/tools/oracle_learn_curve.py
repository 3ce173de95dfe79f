"""CPU learning-curve probe of the ORACLE restatement of the reference loop (dqn/train.py:298-327):
python tests/tools/oracle_learn_curve.py TOTAL_STEPS [HIDDEN]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import dqn_port as dp
from oracle.lbf import MarlbaseEnv

total, H = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.set_num_threads(1)
torch.manual_seed(1)
np.random.seed(1)
P, D, A, T = 2, 15, 6, 25
name = "lbforaging:Foraging-8x8-2p-3f-v3"
env = MarlbaseEnv(name, T, rng=np.random.default_rng(0))
eval_env = MarlbaseEnv(name, T, rng=np.random.default_rng(1))
learner = dp.Learner(dp.init_params(P, D, H, A, seed=0), D, H, A)
rb = dp.ReplayBuffer(10000, P, D, T)
rng = np.random.default_rng(2)
eps_sched = dp.epsilon_schedule("linear", 0.5, 1.0, 0.05, 6.5, total)


def episode(e, eps, store):
    obs, _ = e.reset()
    if store:
        rb.init_episode(obs)
    done, t = False, 0
    while not done:
        o = torch.tensor(np.stack(obs)).unsqueeze(1)
        acts, _ = dp.act(learner.flat().detach(), o, eps, torch.tensor([rng.random()], dtype=torch.float32),
                         torch.tensor(rng.integers(0, A, (P, 1))), D, H, A)
        acts = [int(a) for a in acts[:, 0]]
        obs, rew, d, tr, info = e.step(acts)
        done = d or tr
        if store:
            rb.add(obs, acts, rew, done)
        t += 1
    return t, info


step = updates = last_eval = 0
t0 = time.time()
while step < total + 1:
    t, _ = episode(env, eps_sched(step), True)
    step += t
    if step > 2000 and rb.can_sample(32):
        m = learner.update(rb.sample(32))
        updates += 1
    if step - last_eval >= total // 6:
        rets = [episode(eval_env, 0.05, False)[1]["episode_returns"].sum() for _ in range(100)]
        print(f"step {step} updates {updates} eval_return {np.mean(rets):.4f} loss {m['loss']:.5f} eps {eps_sched(step):.3f} wall {time.time() - t0:.0f}s", flush=True)
        last_eval = step
