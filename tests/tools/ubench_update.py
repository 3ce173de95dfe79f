"""Micro-benchmark of the fused loss/grad kernel alone (for rocprofv3 --pmc passes):
python tests/tools/ubench_update.py [B] [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from codebase_amd import hip as h
from oracle import dqn_port as dp

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
P, T, D, H, A = 2, 25, 15, 64, 6
spec = h.NetSpec(P, D, H, A)
params = dp.init_params(P, D, H, A, seed=1).cuda()
target = dp.init_params(P, D, H, A, seed=2).cuda()
b = dp.synthetic_batch(P, T, B, D, A, seed=3)
batch = h.Batch(*(b[k].cuda().contiguous() for k in ("obss", "actions", "rewards", "dones", "filled")), None)
up = h.DqnUpdater(spec, params, target)
for _ in range(3):
    up.loss_grad(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    up.loss_grad(batch)
torch.cuda.synchronize()
print(f"B={B} loss_grad {1e6 * (time.perf_counter() - t0) / iters:.1f} us/call (pack+lossgrad+reduce)")
if os.environ.get("MARLHIP_PROF"):
    ws = up._ws[(T, B)]
    ws[-128:].zero_()
    up.loss_grad(batch)
    torch.cuda.synchronize()
    pc = ws[-128:].view(torch.int64).cpu().numpy()[:12]
    names = {0: "row loads + masks", 1: "critic forward", 2: "target forward + TD + tile writes", 3: "dH2 + dW3", 4: "dH1 + bootstrap",
             5: "dW2 + dW1", 8: "[staging]", 9: "[task loop]", 10: "[fold+write]", 11: "[kernel total]"}  # 0-5: MARL_STEP_PROF builds only
    print("cycles per wave (1024 waves; one lane per wave reports):")
    for k, n in names.items():
        print(f"  {n:34s} {pc[k] / 1024:10.0f}")
