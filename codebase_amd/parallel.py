"""Data-parallel plumbing of the vectorised path (one process per GPU, torch.distributed; backend
"nccl" == RCCL over xGMI on ROCm, "gloo" for CPU tests).

The path shards trivially: each rank owns N envs and its own replay shard (independent Philox keys),
weights are replicated.  The ONLY exchange is one all-reduce(SUM) of the flat gradient block per update
(44.6 KB at 2 agents x 64-64); clip + Adam then run on every rank with grad_scale = 1/world on the
identical reduced gradient, so replicas stay bitwise in sync without ever broadcasting weights.
Nothing here exists in the reference (it has no collective)."""
import os

import torch


def init_distributed(backend=None):
    """(dist module or None, rank, world, local_rank) from the torchrun environment."""
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world == 1:
        return None, 0, 1, local_rank
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    return dist, rank, world, local_rank


def rank_env_seed(seed, rank):
    """Philox key of a rank's env shard (distinct layouts / action noise per rank)."""
    return (int(seed) + 1000003 * int(rank)) & (2**64 - 1)


def rank_sample_seed(seed, rank):
    """Philox key of a rank's replay-index draws."""
    return (int(seed) + 7919 * int(rank)) & (2**64 - 1)


def setup_device(local_rank):
    """one process per GPU: bind this process to its device before anything allocates (LOCAL_RANK of torchrun)"""
    if torch.cuda.is_available() and not os.environ.get("MARLHIP_ONE_DEVICE"):  # MARLHIP_ONE_DEVICE: N ranks share device 0 (1-GPU test boxes)
        torch.cuda.set_device(int(local_rank) % max(torch.cuda.device_count(), 1))


def all_sum(dist, t):
    """in-place all-reduce(SUM) of a tensor; identity on one process"""
    if dist is not None:
        dist.all_reduce(t)
    return t


def gather_stack(dist, t):
    """[world, *t.shape]: every rank's `t` (same shape everywhere), on every rank.  Built from all-reduce(SUM) of a zero-padded
    stack, the one collective every backend offers for device tensors (gloo has no device all_gather)."""
    if dist is None:
        return t.unsqueeze(0)
    out = torch.zeros((dist.get_world_size(),) + tuple(t.shape), dtype=t.dtype, device=t.device)
    out[dist.get_rank()] = t
    dist.all_reduce(out)
    return out


class GradSync:
    """all-reduce(SUM) of the flat gradient; `scale` is what clip+Adam must multiply by (1/world)."""

    def __init__(self, dist):
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.scale = 1.0 / self.world

    def __call__(self, grad):
        if self.dist is not None:
            self.dist.all_reduce(grad)
        return grad
