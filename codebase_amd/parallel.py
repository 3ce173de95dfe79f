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


class P2PExchange:
    """The in-library exchange (csrc/p2p.hip, marlhip_p2p_*): every rank publishes its gradient in an IPC-shared buffer of its own and
    sums all ranks' buffers in rank order inside ONE kernel - no collective-library launch, no host hop.  `c_fn` / `c_ctx` are what
    marlhip_idqn_update_n_dist takes as its exchange callback (a C function pointer: the update loop never re-enters Python);
    calling the object all-reduces a tensor from the host side (one ctypes call).  torch.distributed only carries the 64-byte
    handles once, at construction.  Build with `try_create`: every rank runs the same sequence of collectives there whatever
    fails locally, and all ranks end up with the exchange or all without."""

    def __init__(self, lib, state, handles, rank, world, max_floats):
        import ctypes

        self.lib, self.state, self._handles = lib, state, handles
        self.rank, self.world, self.max_floats = rank, world, int(max_floats)
        self.c_fn = ctypes.cast(lib.marlhip_p2p_allreduce, ctypes.c_void_p)
        self.c_ctx = self.state

    def __call__(self, t):
        from ._lib import check

        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() <= self.max_floats
        check(self.lib.marlhip_p2p_allreduce(self.state, t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream), "p2p_allreduce")
        return t

    def status(self):
        """0 = every exchange saw all its peers (synchronises with the device)"""
        return int(self.lib.marlhip_p2p_status(self.state))

    def close(self):
        """frees this rank's buffer and unmaps the peers': call after a job-wide barrier with the device idle (as for a communicator)"""
        if getattr(self, "state", None):
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            self.lib.marlhip_p2p_destroy(self.state)
            self.state = None

    @staticmethod
    def try_create(dist, max_floats):
        """the exchange, checked against torch.distributed's all-reduce on a test vector - or None (with the reason logged) when the
        buffers cannot be shared on this system: the caller then keeps the collective"""
        import ctypes
        import logging

        from ._lib import last_error, lib

        log = logging.getLogger(__name__)
        rank, world = dist.get_rank(), dist.get_world_size()

        def everyone(ok):  # the same answer on every rank
            votes = [None] * world
            dist.all_gather_object(votes, bool(ok))
            return all(votes)

        hb = lib.marlhip_p2p_handle_bytes()
        mine = ctypes.create_string_buffer(hb)
        state = ctypes.c_void_p()
        why = None
        if lib.marlhip_p2p_create(rank, world, int(max_floats), ctypes.byref(state), mine) != 0:
            why, state = last_error(), ctypes.c_void_p()
        handles = [None] * world
        dist.all_gather_object(handles, bytes(mine.raw) if why is None else None)
        ok = why is None and all(h is not None for h in handles)
        packed = None
        if ok:
            packed = ctypes.create_string_buffer(b"".join(handles), hb * world)
            if lib.marlhip_p2p_connect(state, packed) != 0:
                why, ok = last_error(), False
        if not everyone(ok):
            if why:
                log.warning("marlhip p2p exchange not available on rank %d (%s); keeping torch.distributed's all-reduce", rank, why)
            if state:
                lib.marlhip_p2p_destroy(state)
            return None
        ex = P2PExchange(lib, state, packed, rank, world, max_floats)
        # Self-test against torch.distributed's all-reduce, at the REAL gradient size and through BOTH launch geometries: the stand-alone
        # kernel (1024 floats per workgroup) and the one the learner's fused reduce uses (marlhip_p2p_allreduce_wave64: a workgroup and a
        # flag per 64 floats, i.e. max_floats / 64 workgroups each spinning on its peer's same-index workgroup).  Several exchanges in a
        # row, so both slots / flag parities and the "peer still reads my previous slot" window are exercised on the links themselves.
        n = int(max_floats)
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        good = True
        # the self-test waits at most MARLHIP_P2P_SELFTEST_TIMEOUT_MS (default 5 s) for a peer: a link that does not carry the flags must cost
        # the set-up seconds, not the training run's diagnostic bound (minutes) per exchange
        keep = os.environ.get("MARLHIP_P2P_TIMEOUT_MS")
        os.environ["MARLHIP_P2P_TIMEOUT_MS"] = os.environ.get("MARLHIP_P2P_SELFTEST_TIMEOUT_MS", "5000")
        # (the try sits INSIDE the loop body: a rank that fails locally - out of memory at the real gradient size, a device fault - still
        # issues the same six all_reduces as its peers, so the collective sequences match and the vote below is reached - ADVICE r5)
        try:
            for k in range(6):
                ref = None
                try:
                    x = torch.randn(n, generator=g).cuda()
                    ref = x.clone()
                except Exception as e:  # noqa: BLE001
                    why, good = str(e), False
                if ref is None:
                    try:
                        ref = torch.zeros(n, device="cuda")
                    except Exception:  # noqa: BLE001 - still take part in the collective, on the host if the device refuses
                        ref = torch.zeros(n)
                dist.all_reduce(ref)
                if not good:
                    continue
                try:
                    fn = lib.marlhip_p2p_allreduce_wave64 if k & 1 else lib.marlhip_p2p_allreduce
                    rc = fn(ex.state, x.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
                    torch.cuda.synchronize()
                    # (rank-ordered float sums: identical on every rank, and equal to the collective's up to its own summation order)
                    good = good and rc == 0 and ex.status() == 0 and bool(torch.allclose(x, ref, rtol=1e-5, atol=1e-5))
                except Exception as e:  # noqa: BLE001 - "not available", never a broken run
                    why, good = str(e), False
        finally:
            if keep is None:
                os.environ.pop("MARLHIP_P2P_TIMEOUT_MS", None)
            else:
                os.environ["MARLHIP_P2P_TIMEOUT_MS"] = keep
        if everyone(good):
            return ex
        log.warning("marlhip p2p exchange: self-test failed on some rank (%s); keeping torch.distributed's all-reduce", why or "sum mismatch / peer timeout")
        ex.close()
        return None


def _shares_a_device(dist):
    """True when two ranks of the job run on the same physical GPU (uuid / PCI bus id gathered over the ranks).  The in-library exchange
    spin-waits for the peer's same-index workgroup: with two ranks time-sharing one device a spinning launch can keep the peer's off
    the chip until the timeout, so such jobs keep the collective unless MARLHIP_P2P_SHARED_DEVICE=1 (the one-GPU test boxes)."""
    try:
        pr = torch.cuda.get_device_properties(torch.cuda.current_device())
        me = str(getattr(pr, "uuid", None) or getattr(pr, "pci_bus_id", None) or torch.cuda.current_device())
    except Exception:  # noqa: BLE001
        me = str(torch.cuda.current_device())
    ids = [None] * dist.get_world_size()
    dist.all_gather_object(ids, me)
    return len(set(ids)) < len(ids)


class GradSync:
    """all-reduce(SUM) of the flat gradient; `scale` is what clip+Adam must multiply by (1/world).
    max_floats > 0 on GPU ranks: the in-library peer-to-peer exchange (P2PExchange) when it can be set up and passes its self-test
    (MARLHIP_P2P=0 keeps torch.distributed's collective: backend nccl == RCCL, gloo in the CPU tests).
    side_floats > 0: a SECOND, independent exchange lane (`.side`, itself a GradSync) for a slice of the gradient that is reduced on
    another stream while this one is in use - the critics' slice of an actor-critic update whose backward pass runs next to the following
    rollout (A2CNetwork.update_async(overlap=True)).  It owns its own IPC buffers / flags and its own process group (communicator), so
    the two lanes never order against each other; every rank must construct it (a collective set-up like this one's)."""

    def __init__(self, dist, max_floats=0, side_floats=0, _group=None):
        self.dist = dist
        self.group = _group
        self.world = dist.get_world_size() if dist is not None else 1
        self.scale = 1.0 / self.world
        self.p2p = None
        self.fallback_reason = None  # why the collective carries the gradients although the in-library exchange was wanted
        if dist is not None and max_floats > 0 and torch.cuda.is_available():
            if os.environ.get("MARLHIP_P2P", "1") == "0":
                self.fallback_reason = "MARLHIP_P2P=0"
            elif _shares_a_device(dist) and os.environ.get("MARLHIP_P2P_SHARED_DEVICE", "0") != "1":
                import logging

                logging.getLogger(__name__).warning("marlhip p2p exchange: two ranks share a GPU; keeping torch.distributed's all-reduce "
                                                    "(MARLHIP_P2P_SHARED_DEVICE=1 overrides)")
                self.fallback_reason = "two ranks share a GPU"
            else:
                self.p2p = P2PExchange.try_create(dist, max_floats)
                if self.p2p is None:
                    self.fallback_reason = "set-up or self-test failed on some rank (see the log)"
        # what the library's n-updates loop takes instead of a Python callback (hip.FusedLearner.run)
        self.c_fn = self.p2p.c_fn if self.p2p is not None else None
        self.c_ctx = self.p2p.c_ctx if self.p2p is not None else None
        self.side = None
        if dist is not None and side_floats > 0:
            self.side = GradSync(dist, max_floats=side_floats, _group=dist.new_group())

    def describe(self):
        """what carried the gradients, for logs and bench lines"""
        return {"exchange": "p2p (in-library)" if self.p2p is not None else "collective (torch.distributed.all_reduce)",
                "self_test": "passed" if self.p2p is not None else None, "fallback_reason": self.fallback_reason,
                "side_lane": self.side.describe() if self.side is not None else None}

    def __call__(self, grad):
        if self.p2p is not None and grad.is_cuda and grad.dtype == torch.float32 and grad.numel() <= self.p2p.max_floats:
            self.p2p(grad)
        elif self.dist is not None:
            self.dist.all_reduce(grad, group=self.group)
        return grad

    def check(self):
        """raises ON EVERY RANK when an in-library exchange ran into its peer timeout on any of them (one small collective + a 4-byte
        device read: for log / evaluation / save points and the end of a run, not for the update loop)"""
        if self.side is not None:
            self.side.check()
        if self.p2p is None:
            return
        bad = self.p2p.status() != 0
        if self.dist is not None and self.world > 1:
            votes = [None] * self.world
            self.dist.all_gather_object(votes, bool(bad))
            bad = any(votes)
        if bad:
            raise RuntimeError("marlhip p2p exchange: a peer did not publish its gradient in time (MARLHIP_P2P_TIMEOUT_MS); replicas have diverged")

    def close(self):
        """end of a run: verify, then free the exchange behind a job-wide barrier (no peer may still be reading this rank's buffer)"""
        if self.side is not None:
            self.side.close()
            self.side = None
        if self.p2p is None:
            return
        try:
            self.check()
        finally:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            if self.dist is not None and self.world > 1:
                self.dist.barrier()
            self.p2p.close()
            self.p2p = self.c_fn = self.c_ctx = None
