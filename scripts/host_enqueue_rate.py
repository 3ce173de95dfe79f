"""Host cost of the per-update python loop the N > 1 path uses (loss/grad -> grad sync -> clip+Adam, 4 launches): enqueue time
per update on the host against the device time per update.  Measured on the GPU box: 27 us of host work per 129 us update -
the data-parallel loop is device-bound with room for the RCCL all-reduce enqueue."""
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
from codebase_amd import hip as h
from codebase_amd.dqn.model import QNetwork
from codebase_amd.dqn.train import VectorisedIDQN
from codebase_amd.utils.envs import _space_pair
N, T = 4096, 25
cfg = h.lbf_config("lbforaging:Foraging-8x8-2p-3f-v3", N, T, seed=1)
obs_space, act_space = _space_pair(cfg)
hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False, target_update_interval_or_tau=200)
m = QNetwork(obs_space, act_space, hyper, [64, 64], False, False, True, "cuda")
tr = VectorisedIDQN(cfg, m, 4 * N, T, N, 32, seed=3)
tr.round(0.5); torch.cuda.synchronize()
# python per-update loop, host enqueue rate (no sync inside), with a dummy "grad_sync" doing a tiny device op like an all-reduce enqueue would
def fake_sync(g): g.add_(0.0)
K = 400
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(K):
    m.update_async(N, grad_sync=fake_sync, world=1, replay=tr.replay, length=N, seed=1, counter=i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {1e6*(t1-t0)/K:.1f} us/update; total {1e6*(t2-t0)/K:.1f} us/update (GPU-bound if total > enqueue)")
