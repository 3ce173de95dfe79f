"""In-kernel cycle counters of dqn_lossgrad_kernel (MARLHIP_PROF=1: the last 128 bytes of the update workspace; sums over lane 0 of every wave)
at B = 32 (the reference's own cadence) and B = 4096 (the bench's ratio cadence): [8] pack staging, [9] task loop, [10] fold + record, [11] whole
kernel.  Measured (MI355X, scripts/gpu_runs/r3V.sh), per wave: B = 32 - staging 4.0 k cycles, ONE time step + its bootstrap forward 24.6 k,
fold 5.7 k, 34.3 k in all = 15.6 us of the 19.4 us launch; B = 4096 - 14.9 k cycles per step over 13.5 steps.  The small update is one
step per wave already: its latency floor, not a scheduling problem."""
import os, sys, torch, numpy as np
os.environ["MARLHIP_PROF"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from codebase_amd import hip as h
def run(B, n):
    P, D, H, A, T = 2, 15, 64, 6, 25
    spec = h.NetSpec(P, D, H, A)
    cap = max(2 * B, 64)
    rb = h.DeviceReplay(cap, P, D, T)
    rb.obs.normal_(); rb.act.random_(0, A); rb.rew.uniform_(); rb.filled.fill_(1); rb.done.zero_()
    n = spec.nparams() if callable(spec.nparams) else spec.nparams
    params = (0.1 * torch.randn(P, n, generator=torch.Generator().manual_seed(1))).cuda(); target = params.clone()
    up = h.DqnUpdater(spec, params, target, lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True)
    fl = h.FusedLearner(up, rb, B, 200, mode=0)
    upd, last = fl.run(4, cap, 1, 0, 0, 0)
    torch.cuda.synchronize()
    ws = fl._keep[1]
    tail = ws.view(torch.uint8)[-128:].view(torch.int64)
    tail.zero_()
    upd, last = fl.run(n, cap, 1, 4, upd, last)
    torch.cuda.synchronize()
    c = tail.cpu().numpy()
    return c
for B in (32, 4096):
    n = 64
    c = run(B, n)
    print("B", B, "per-launch sums over lane0 of every wave:", {k: int(c[k]) // n for k in (8, 9, 10, 11)})
