"""Tensor-level Python face of libmarlhip.so: PyTorch-ROCm tensors in, kernels enqueued on the
current torch stream, nothing copied to the host.  torch is plumbing here (device memory,
streams); every computation below happens in the HIP library."""
import ctypes
from collections import namedtuple
from dataclasses import dataclass

import torch

import os

from . import _lib
from ._lib import (AcConfig, BatchStruct, IdqnLearner, QmixLearner, RetStatsStruct, QmixMixer, LbfBuffers, LbfConfig, MarlHipError, RwareConfig, NetShape, ReplayBuffers, ReplayShape, check, lib)

Batch = namedtuple("Batch", ["obss", "actions", "rewards", "dones", "filled", "action_mask"])  # dqn/train.py:14-16


# MARLHIP_AC_NO_KEEP=1: A2C's step always runs the actors' forward pass itself (diagnostics; the collector then keeps nothing)
_NO_KEEP = bool(os.environ.get("MARLHIP_AC_NO_KEEP"))
# MARLHIP_AC_NO_OVERLAP=1: the critics' backward pass, step and target update stay on the caller's stream (diagnostics);
# MARLHIP_AC_FORCE_OVERLAP=1: overlap whatever the rollout's size; MARLHIP_SIDE_PATTERN=0|1: which half of the compute units the critics' stream owns
_NO_OVERLAP = bool(os.environ.get("MARLHIP_AC_NO_OVERLAP"))
_FORCE_OVERLAP = bool(os.environ.get("MARLHIP_AC_FORCE_OVERLAP"))
_SIDE_PATTERN = int(os.environ.get("MARLHIP_SIDE_PATTERN", "0"))
# MARLHIP_SIDE_SHARE=1..100: the percentage of the compute units the critics' stream owns (50; the two-ranks-on-one-device test rigs set 100:
# see tests/test_gpu_two_ranks.py)
_SIDE_SHARE = min(100, max(1, int(os.environ.get("MARLHIP_SIDE_SHARE", "50"))))


# streams that own a share of the compute units (marlhip_stream_create_cu_share): one per device, shared by the updaters of the process and
# destroyed before the interpreter goes down (a stream of this kind that outlives the runtime's own teardown crashes a profiled process
# in __cxa_finalize)
_HALF_CHIP_STREAMS = {}


def _half_chip_stream(device):
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _HALF_CHIP_STREAMS:
        h = ctypes.c_void_p()
        with torch.cuda.device(key):
            check(lib.marlhip_stream_create_cu_share(_SIDE_SHARE, _SIDE_PATTERN, ctypes.byref(h)), "stream_create_cu_share")
        _HALF_CHIP_STREAMS[key] = (torch.cuda.ExternalStream(h.value, device=torch.device("cuda", key)), h)
    return _HALF_CHIP_STREAMS[key][0]


def side_stream_description(device=None):
    """what the critics' stream owns, for bench lines (the mask's meaning is device-specific: include/marlhip.h)"""
    cus = torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device()).multi_processor_count
    return {"api": "hipExtStreamCreateWithCUMask", "percent": _SIDE_SHARE, "pattern": _SIDE_PATTERN, "compute_units": cus, "mask_bits_set": (cus * _SIDE_SHARE + 99) // 100,
            "mask": ("the lowest-numbered mask bits" if _SIDE_PATTERN == 0 else "every other mask bit") + " (MARLHIP_SIDE_PATTERN; bit -> physical unit is the runtime's order)"}


def _destroy_half_chip_streams():
    for key, (stream, h) in list(_HALF_CHIP_STREAMS.items()):
        try:
            with torch.cuda.device(key):
                stream.synchronize()
                lib.marlhip_stream_destroy(h)
        except Exception:  # noqa: BLE001 - interpreter shutdown: nothing left to report to
            pass
    _HALF_CHIP_STREAMS.clear()


import atexit  # noqa: E402

atexit.register(_destroy_half_chip_streams)


def _require_gpu():
    if not torch.cuda.is_available() or not lib.marlhip_device_available():
        raise MarlHipError("marlhip needs an AMD GPU (gfx950): no HIP device is visible and there is no CPU path")


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _mask_ptr(mask, shape):
    """device pointer of an optional f32 action-mask tensor of the given shape (marlhip_batch.action_mask), None when absent"""
    if mask is None:
        return None
    assert mask.dtype == torch.float32 and tuple(mask.shape) == tuple(shape) and mask.is_contiguous() and mask.is_cuda, \
        f"action mask must be a contiguous f32 device tensor of shape {tuple(shape)}"
    return mask.data_ptr()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_FWD_WS = {}


def _fwd_ws(spec, device):
    """(pointer, bytes) of the weight-pack workspace the forward-only entry points take (marlhip_forward_workspace_bytes): one
    buffer per (net shape, device, stream) - a call's packs are read by that call's kernels only, and calls on one stream are ordered"""
    key = (spec.n_agents, spec.obs_dim, spec.hidden, torch.device(device).index, torch.cuda.current_stream().cuda_stream)
    ws = _FWD_WS.get(key)
    if ws is None:
        s = spec.c()
        n = check(lib.marlhip_forward_workspace_bytes(ctypes.byref(s)), "forward_workspace_bytes")
        ws = _FWD_WS[key] = torch.empty(n, dtype=torch.uint8, device=device)
    return ctypes.c_void_p(ws.data_ptr()), ws.numel()


def parse_lbf_name(name):
    """'lbforaging:Foraging-8x8-2p-3f[-coop][-2s][-pen]-v3' -> upstream registration kwargs."""
    base = name.split(":")[-1]
    parts = base.split("-")
    if not parts[0].startswith("Foraging"):
        raise ValueError(f"not a Level-Based Foraging id: {name}")
    size = next(p for p in parts if "x" in p and p.replace("x", "").isdigit())
    s = int(size.split("x")[0])
    p = int(next(q for q in parts if q.endswith("p") and q[:-1].isdigit())[:-1])
    f = int(next(q for q in parts if q.endswith("f") and q[:-1].isdigit())[:-1])
    return dict(n_agents=p, n_food=f, rows=s, cols=s, sight=2 if "2s" in parts else s, max_episode_steps=50,
                force_coop=int("coop" in parts), min_player_level=1, max_player_level=2 if parts[-1] == "v3" else 3,
                min_food_level=1, max_food_level=0, normalize_reward=1, penalty=0.1 if "pen" in parts else 0.0)


def lbf_config(name, n_envs, time_limit, seed=0, cooperative=False, **over):
    kw = parse_lbf_name(name)
    kw.update(over)
    return LbfConfig(n_envs=n_envs, time_limit=int(time_limit or 0), seed=int(seed) & (2**64 - 1),
                     cooperative=int(cooperative), **kw)


RWARE_SIZES = {"tiny": (1, 3), "small": (2, 3), "medium": (2, 5), "large": (3, 5)}  # (shelf_rows, shelf_columns)


def parse_rware_name(name):
    """'rware:rware-tiny-4ag[-easy|-hard]-v2' (also 'rware:tiny-4ag') -> upstream registration kwargs (rware/__init__.py)."""
    parts = [p for p in name.split(":")[-1].split("-") if p != "rware" and not (p[:1] == "v" and p[1:].isdigit())]
    size = next((p for p in parts if p in RWARE_SIZES), None)
    ag = next((p for p in parts if p.endswith("ag") and p[:-2].isdigit()), None)
    if size is None or ag is None:
        raise ValueError(f"not a warehouse id: {name}")
    agents = int(ag[:-2])
    scale = 2.0 if "easy" in parts else (0.5 if "hard" in parts else 1.0)
    return dict(n_agents=agents, shelf_rows=RWARE_SIZES[size][0], shelf_columns=RWARE_SIZES[size][1], column_height=8,
                request_queue_size=int(agents * scale), max_steps=500, max_inactivity_steps=0, reward_type=1)


def rware_config(name, n_envs, time_limit, seed=0, cooperative=False, **over):
    kw = parse_rware_name(name)
    kw.update(over)
    kw["max_steps"] = int(kw["max_steps"] or 0)
    kw["max_inactivity_steps"] = int(kw["max_inactivity_steps"] or 0)
    return RwareConfig(n_envs=n_envs, time_limit=int(time_limit or 0), seed=int(seed) & (2**64 - 1),
                       cooperative=int(cooperative), **kw)


def is_rware(cfg):
    return isinstance(cfg, RwareConfig)


def env_config(name, n_envs, time_limit, seed=0, cooperative=False, **over):
    """LbfConfig or RwareConfig from a gymnasium id"""
    if "rware" in name:
        return rware_config(name, n_envs, time_limit, seed=seed, cooperative=cooperative, **over)
    return lbf_config(name, n_envs, time_limit, seed=seed, cooperative=cooperative, **over)


def env_dims(cfg):
    """(obs_dim, n_actions) of a batched env config"""
    if is_rware(cfg):
        return 71 + (cfg.n_agents if cfg.observe_id else 0), 5
    return 3 * (cfg.n_agents + cfg.n_food) + (cfg.n_agents if cfg.observe_id else 0), 6


def attach_reward_stats(cfg, device="cuda"):
    """env.standardise_rewards: allocate the per-env streaming records [n_envs][3P+1] and point the config at them."""
    _require_gpu()
    t = torch.zeros(cfg.n_envs, 3 * cfg.n_agents + 1, dtype=torch.float32, device=device)
    cfg.reward_stats = t.data_ptr()
    cfg._reward_stats_tensor = t  # ctypes.Structure instances accept attributes: the tensor lives as long as the config
    return t


class BatchedForaging:
    """N Level-Based Foraging (or, with an RwareConfig, warehouse) envs resident in HBM (K1)."""

    def __init__(self, cfg, device="cuda"):
        _require_gpu()
        self.cfg = cfg
        self.device = torch.device(device)
        self.N, self.P = cfg.n_envs, cfg.n_agents
        if is_rware(cfg):  # the warehouse env behind the same buffers and call shapes
            self._fn = (lib.marlhip_rware_reset, lib.marlhip_rware_observe, lib.marlhip_rware_step)
            self.stride = check(lib.marlhip_rware_state_stride(ctypes.byref(cfg)), "rware_state_stride")
            r, c, n = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
            check(lib.marlhip_rware_grid(ctypes.byref(cfg), ctypes.byref(r), ctypes.byref(c), ctypes.byref(n)), "rware_grid")
            self.rows, self.cols, self.n_shelves = r.value, c.value, n.value
        else:
            self._fn = (lib.marlhip_lbf_reset, lib.marlhip_lbf_observe, lib.marlhip_lbf_step)
            self.F = cfg.n_food
            self.stride = check(lib.marlhip_lbf_state_stride(ctypes.byref(cfg)), "lbf_state_stride")
        self.D, self.A = env_dims(cfg)  # ObserveID: one-hot agent index first
        dev = self.device
        self.state = torch.zeros(self.N, self.stride, dtype=torch.uint8, device=dev)
        self.episode = torch.zeros(self.N, dtype=torch.int32, device=dev)  # u32 bit pattern
        self.ep_return = torch.zeros(self.P, self.N, dtype=torch.float32, device=dev)
        self.ep_length = torch.zeros(self.N, dtype=torch.int32, device=dev)
        self.fin_return = torch.zeros(self.P, self.N, dtype=torch.float32, device=dev)
        self.fin_length = torch.zeros(self.N, dtype=torch.int32, device=dev)
        self.obs = torch.zeros(self.P, self.N, self.D, dtype=torch.float32, device=dev)
        self.rewards = torch.zeros(self.P, self.N, dtype=torch.float32, device=dev)
        self.done = torch.zeros(self.N, dtype=torch.uint8, device=dev)
        self.truncated = torch.zeros(self.N, dtype=torch.uint8, device=dev)
        self.final_obs = torch.zeros(self.P, self.N, self.D, dtype=torch.float32, device=dev)
        self._buf = LbfBuffers(self.state.data_ptr(), self.episode.data_ptr(), self.ep_return.data_ptr(),
                               self.ep_length.data_ptr())

    def reset(self, mask=None):
        check(self._fn[0](ctypes.byref(self.cfg), ctypes.byref(self._buf), _ptr(mask), _ptr(self.obs), _stream()), "env reset")
        return self.obs

    def observe(self):
        check(self._fn[1](ctypes.byref(self.cfg), ctypes.byref(self._buf), _ptr(self.obs), _stream()), "env observe")
        return self.obs

    def step(self, actions, active=None, auto_reset=False):
        """actions int32 [P][N] (device).  Returns (obs, rewards, done, truncated) device tensors."""
        assert actions.dtype == torch.int32 and actions.shape == (self.P, self.N) and actions.is_contiguous()
        check(self._fn[2](ctypes.byref(self.cfg), ctypes.byref(self._buf), _ptr(active), _ptr(actions), _ptr(self.obs),
                                   _ptr(self.rewards), _ptr(self.done), _ptr(self.truncated), _ptr(self.fin_return),
                                   _ptr(self.fin_length), int(auto_reset), _ptr(self.final_obs) if auto_reset else None,
                                   _stream()), "env step")
        return self.obs, self.rewards, self.done, self.truncated


@dataclass
class NetSpec:
    n_agents: int
    obs_dim: int
    hidden: int
    n_actions: int
    sharing: tuple = None  # agent -> network index (parameter sharing / SePS); None = independent networks
    wide: bool = False     # no fused kernels for this shape (hidden > 128, ...): the GEMM path (marlhip_wide_*, csrc/wide_mlp.h)
    n_hidden: int = 2      # hidden layers (the fused kernels: 2; 1..16 on the GEMM path)

    def c(self):
        s = NetShape(self.n_agents, self.obs_dim, self.hidden, self.n_actions)
        s.n_hidden = int(self.n_hidden)
        if self.sharing is not None:
            if len(self.sharing) != self.n_agents or self.n_agents > 16:
                raise ValueError("sharing indices: one per agent, at most 16 agents")
            s.n_networks = max(self.sharing) + 1
            for i, k in enumerate(self.sharing):
                s.net_of[i] = int(k)
        return s

    @property
    def n_blocks(self):
        return self.n_agents if self.sharing is None else max(self.sharing) + 1

    def nparams(self):
        s = self.c()
        if self.wide:
            return check(lib.marlhip_wide_nparams(ctypes.byref(s), self.n_actions), "wide_nparams")
        return check(lib.marlhip_net_nparams(ctypes.byref(s)), "net_nparams")


_WIDE_WS = {}


def wide_forward(spec: NetSpec, params, obs, n_out=None, agent_stride=None, row_stride=None, n_rows=None):
    """networks without a fused kernel (spec.wide): out[p][row][:] = MLP_p(obs row) through the GEMM path.  obs f32 [P][N][D]
    (or any layout given agent_stride / row_stride / n_rows) -> [P][N][n_out]"""
    _require_gpu()
    if n_rows is None:
        P, n_rows, D = obs.shape
        agent_stride, row_stride = n_rows * D, D
    n_out = spec.n_actions if n_out is None else int(n_out)
    s = spec.c()
    key = (spec.n_agents, spec.obs_dim, spec.hidden, spec.n_hidden, int(n_rows), params.device.index, torch.cuda.current_stream().cuda_stream)
    ws = _WIDE_WS.get(key)
    if ws is None:
        if len(_WIDE_WS) > 32:
            _WIDE_WS.clear()
        n = check(lib.marlhip_wide_forward_workspace_bytes(ctypes.byref(s), int(n_rows)), "wide_forward_workspace_bytes")
        ws = _WIDE_WS[key] = torch.empty(int(n), dtype=torch.uint8, device=params.device)
    out = torch.empty(spec.n_agents, int(n_rows), n_out, device=params.device)
    check(lib.marlhip_wide_forward(ctypes.byref(s), n_out, _ptr(params), _ptr(obs), int(agent_stride), int(row_stride), int(n_rows), _ptr(out),
                                   _ptr(ws), ws.numel(), _stream()), "wide_forward")
    return out


def dqn_act(spec: NetSpec, params, obs, epsilon, u=None, rand_actions=None, seed=0, episode=None, ep_length=None,
            actions=None, q_out=None):
    """Batched QNetwork.act (K2).  params f32 [P][nparams], obs f32 [P][N][D] -> actions i32 [P][N]."""
    _require_gpu()
    P, N, _ = obs.shape
    if actions is None:
        actions = torch.empty(P, N, dtype=torch.int32, device=obs.device)
    s = spec.c()
    check(lib.marlhip_dqn_act(ctypes.byref(s), _ptr(params), _ptr(obs), N, float(epsilon), _ptr(u), _ptr(rand_actions),
                              int(seed) & (2**64 - 1), _ptr(episode), _ptr(ep_length), _ptr(actions), _ptr(q_out), _stream()),
          "dqn_act")
    return actions


class DeviceReplay:
    """Episode-major replay in HBM (K3/K4) - ReplayBuffer of marlbase/dqn/train.py:19-124."""

    def __init__(self, capacity, n_agents, obs_dim, max_len, device="cuda"):
        _require_gpu()
        self.shape = ReplayShape(capacity, n_agents, obs_dim, max_len)
        self.capacity, self.P, self.D, self.T = capacity, n_agents, obs_dim, max_len
        dev = torch.device(device)
        self.device = dev
        self.obs = torch.zeros(capacity, n_agents, max_len + 1, obs_dim, dtype=torch.float32, device=dev)
        self.act = torch.zeros(capacity, n_agents, max_len, dtype=torch.uint8, device=dev)
        self.rew = torch.zeros(capacity, n_agents, max_len, dtype=torch.float32, device=dev)
        self.done = torch.zeros(capacity, max_len + 1, dtype=torch.uint8, device=dev)
        self.filled = torch.zeros(capacity, max_len, dtype=torch.uint8, device=dev)
        self.bufs = ReplayBuffers(self.obs.data_ptr(), self.act.data_ptr(), self.rew.data_ptr(), self.done.data_ptr(),
                                  self.filled.data_ptr())
        self._out = {}

    def init_episode(self, slot, obs, active=None):
        check(lib.marlhip_replay_init_episode(ctypes.byref(self.shape), ctypes.byref(self.bufs), _ptr(slot), _ptr(active),
                                              _ptr(obs), obs.shape[1], _stream()), "replay_init_episode")

    def add(self, slot, t, obs, actions, rewards, done, active=None):
        check(lib.marlhip_replay_add(ctypes.byref(self.shape), ctypes.byref(self.bufs), _ptr(slot), _ptr(t), _ptr(active),
                                     _ptr(obs), _ptr(actions), _ptr(rewards), _ptr(done), obs.shape[1], _stream()), "replay_add")

    def add_step(self, slot, t, env, actions, alive, proper=False):
        """marlhip_replay_add_step: ReplayBuffer.add of step t for the alive envs straight from the env-step outputs, then
        alive &= ~(done | truncated)"""
        check(lib.marlhip_replay_add_step(ctypes.byref(self.shape), ctypes.byref(self.bufs), _ptr(slot), int(t), int(bool(proper)), _ptr(alive),
                                          _ptr(env.obs), _ptr(actions), _ptr(env.rewards), _ptr(env.done), _ptr(env.truncated), env.N, _stream()),
              "replay_add_step")

    def _outputs(self, B):
        if B not in self._out:
            dev, P, T, D = self.device, self.P, self.T, self.D
            self._out[B] = (torch.empty(P, T + 1, B, D, dtype=torch.float32, device=dev),
                            torch.empty(P, T, B, dtype=torch.int64, device=dev),
                            torch.empty(P, T, B, dtype=torch.float32, device=dev),
                            torch.empty(T + 1, B, dtype=torch.float32, device=dev),
                            torch.empty(T, B, dtype=torch.float32, device=dev),
                            torch.empty(B, dtype=torch.int32, device=dev))
        return self._out[B]

    def sample(self, batch_size, length=None, idx=None, seed=0, counter=0, fresh=False):
        """idx (int32 device tensor) given: gather exactly those episodes; else draw them on the device
        from Philox(seed, counter) uniformly in [0, length).  Output tensors are re-used per batch size
        unless fresh=True."""
        outs = self._outputs(batch_size)
        if fresh:
            outs = tuple(torch.empty_like(o) for o in outs)
        obss, actions, rewards, dones, filled, idx_out = outs
        check(lib.marlhip_replay_sample(ctypes.byref(self.shape), ctypes.byref(self.bufs), _ptr(idx), batch_size,
                                        int(length or 0), int(seed) & (2**64 - 1), int(counter) & 0xFFFFFFFF, _ptr(idx_out),
                                        _ptr(obss), _ptr(actions), _ptr(rewards), _ptr(dones), _ptr(filled), _stream()),
              "replay_sample")
        return Batch(obss, actions, rewards, dones, filled, None)


OPTIMIZERS = {"Adam": 0, "SGD": 1, "RMSprop": 2, "AdamW": 3}  # marlhip_dqn_clip_step's ids (torch's default hyper-parameters)


def optimizer_id(opt):
    """cfg.optimizer (a name, or the torch.optim class itself) -> marlhip_dqn_clip_step's id"""
    name = opt if isinstance(opt, str) else getattr(opt, "__name__", str(opt))
    if name not in OPTIMIZERS:
        raise NotImplementedError(f"optimizer {name}: built are {sorted(OPTIMIZERS)} (torch.optim defaults, lr from the config)")
    return OPTIMIZERS[name]


def _clip_step(opt, n_tensor, params, grad, s1, s2, target, step, lr, betas, eps, clip, grad_scale, hard, tau, scratch, gnorm, what):
    if opt == 0:
        check(lib.marlhip_dqn_clip_adam(n_tensor, _ptr(params), _ptr(grad), _ptr(s1), _ptr(s2), _ptr(target), step, float(lr), float(betas[0]),
                                        float(betas[1]), float(eps), float(clip), float(grad_scale), int(bool(hard)), float(tau), _ptr(scratch),
                                        _ptr(gnorm), _stream()), what)
    else:
        check(lib.marlhip_dqn_clip_step(int(opt), n_tensor, _ptr(params), _ptr(grad), _ptr(s1), _ptr(s2), _ptr(target), step, float(lr), float(clip),
                                        float(grad_scale), int(bool(hard)), float(tau), _ptr(scratch), _ptr(gnorm), _stream()), what)


# the library's callbacks into the host's collective (include/marlhip.h: marlhip_exchange_fn / marlhip_exchange_f64_fn)
EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)


class Exchange:
    """a ctypes callback that all-reduces (SUM) ONE known device tensor through `reduce(tensor)` - torch.distributed.all_reduce on the
    current stream (backend nccl == RCCL over xGMI; gloo in the tests).  An exception inside the callback is kept and re-raised by
    `check()` after the library call returned (ctypes swallows exceptions in callbacks)."""

    def __init__(self, tensor, reduce):
        self.tensor, self.reduce, self.error = tensor, reduce, None

        def cb(ctx, buf, count, stream):
            try:
                if buf != self.tensor.data_ptr() or count != self.tensor.numel():
                    raise MarlHipError(f"exchange: the library handed over {count} elements at {buf:#x}, expected the registered tensor")
                self.reduce(self.tensor)
                return 0
            except BaseException as e:  # noqa: BLE001 - must not propagate through the C frames
                self.error = e
                return -1

        self._cb = EXCHANGE_FN(cb)
        self.ptr = ctypes.cast(self._cb, ctypes.c_void_p)

    def check(self):
        if self.error is not None:
            e, self.error = self.error, None
            raise e


class RunningReturnStats:
    """RunningMeanStd of marlbase/utils/standardise_stream.py, resident on the device: shape (n_agents,) for the independent
    learner; for VDNetwork / QMixNetwork `columns` = batch size - their RunningMeanStd(shape=(1,)) turns into one (mean, var) per
    batch column at its first update (dqn/model.py:256-264,415-422; include/marlhip.h, marlhip_ret_stats)."""

    def __init__(self, n, device, epsilon=1e-4, columns=0):
        self.columns = int(columns)
        self.mean = torch.zeros(n, dtype=torch.float32, device=device)
        self.var = torch.ones(n, dtype=torch.float32, device=device)
        self.count_t = torch.full((1,), epsilon, dtype=torch.float64, device=device)
        self.exchange = None

    @property
    def count(self):
        return float(self.count_t.item())

    def attach_exchange(self, reduce):
        """data-parallel training: the batch moments of the per-agent statistics are summed over the ranks (`reduce(tensor)` = an
        all-reduce SUM in place) before the running update, so every rank keeps the statistics of the GLOBAL batch
        (standardise_stream.py:15-20 on the concatenated batch).  Per-batch-column statistics stay local (marlhip_ret_stats)."""
        if self.columns == 0 and reduce is not None:
            self.moments = torch.zeros(2 * self.mean.numel() + 1, dtype=torch.float64, device=self.mean.device)
            self.exchange = Exchange(self.moments, reduce)
        return self

    def c(self):
        st = RetStatsStruct(self.mean.data_ptr(), self.var.data_ptr(), self.count_t.data_ptr(), self.columns)
        if self.exchange is not None:
            st.exchange, st.moments = self.exchange.ptr, self.moments.data_ptr()
        return st


class DqnUpdater:
    """K5-K8: loss + gradient, clip, Adam, target update over flat per-agent parameter blocks."""

    def __init__(self, spec: NetSpec, params, target, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, gamma=0.99, grad_clip=1.0,
                 double_q=True, standardise_returns=False, optimizer="Adam", split16=False):
        _require_gpu()
        self.optimizer = optimizer_id(optimizer)
        # OPT-IN deviation from the exact-f32 kernels (include/marlhip.h, marlhip_*_split16): products from fp16 halves on the double-rate
        # matrix pipe, fp32 accumulate.  IDQN (mode 0) only; everything else keeps the f32 entry points.
        self.split16 = bool(split16)
        if self.split16 and (standardise_returns or self.optimizer != 0 or spec.wide or spec.hidden != 64 or spec.obs_dim > 32):
            raise NotImplementedError("split16: the opt-in split-fp16 learner covers IDQN with Adam, layers [64, 64], observation width <= 32")
        self.spec, self.params, self.target = spec, params, target
        self.ret_stats = RunningReturnStats(spec.n_agents, params.device) if standardise_returns else None
        self.lr, self.betas, self.eps, self.gamma = lr, betas, eps, gamma
        self.grad_clip = float(grad_clip) if grad_clip else 0.0
        self.double_q = int(bool(double_q))
        dev = params.device
        self.exp_avg = torch.zeros_like(params)
        self.exp_avg_sq = torch.zeros_like(params)
        self.grad = torch.zeros_like(params)
        self.loss = torch.zeros(2, dtype=torch.float32, device=dev)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.scratch = torch.zeros((params.numel() + 63) // 64 + 1, dtype=torch.float32, device=dev)  # clip-norm partials (update_n: one per 64)
        self.step = 0
        self._ws = {}
        self._gru_ws_cache = {}

    def _workspace(self, T, B):
        key = (T, B)
        if key not in self._ws:
            s = self.spec.c()
            n = check(lib.marlhip_dqn_workspace_bytes(ctypes.byref(s), T, B), "dqn_workspace_bytes")
            self._ws[key] = torch.empty(max(int(n), 4), dtype=torch.uint8, device=self.params.device)
        return self._ws[key]

    def _stats_for(self, mode, B):
        """the running return statistics in the shape the learner keeps them: per agent (IDQN, mode 0) or per batch column (VDN /
        QMIX: the reference's (1,)-shaped RunningMeanStd becomes [B] at its first update and pins the batch size)"""
        st = self.ret_stats
        if mode == 0:
            if st.columns != 0:
                raise MarlHipError("standardise_returns: these statistics belong to a VDN / QMIX learner")
            return st
        if st.columns == 0 and st.count <= 1e-4:  # untouched per-agent placeholder of the constructor: take the column form
            st = self.ret_stats = RunningReturnStats(B, self.params.device, columns=B)
        if st.columns != B:
            raise MarlHipError(f"standardise_returns: the statistics hold {st.columns} batch columns, the batch has {B} "
                               "(VDNetwork / QMixNetwork: RunningMeanStd(shape=(1,)) pins the batch size at its first update)")
        return st

    def loss_grad(self, batch, mode=0):
        T, B = batch.filled.shape
        ws = self._workspace(T, B)
        bs = BatchStruct(batch.obss.data_ptr(), batch.actions.data_ptr(), batch.rewards.data_ptr(), batch.dones.data_ptr(),
                         batch.filled.data_ptr(), T, B, 0, 0, 0, 0, _mask_ptr(batch.action_mask, (self.spec.n_agents, T + 1, B, self.spec.n_actions)))
        s = self.spec.c()
        if self.ret_stats is not None:
            st = self._stats_for(mode, B).c()
            check(lib.marlhip_dqn_loss_grad_std(ctypes.byref(s), _ptr(self.params), _ptr(self.target), ctypes.byref(bs),
                                                float(self.gamma), self.double_q, int(mode), ctypes.byref(st), _ptr(ws), ws.numel(),
                                                _ptr(self.grad), _ptr(self.loss), _stream()), "dqn_loss_grad_std")
            return self.loss, self.grad
        if self.split16:
            if mode != 0:
                raise NotImplementedError("split16: IDQN (mode 0) only")
            check(lib.marlhip_dqn_loss_grad_split16(ctypes.byref(s), _ptr(self.params), _ptr(self.target), ctypes.byref(bs), float(self.gamma),
                                                    self.double_q, _ptr(ws), ws.numel(), _ptr(self.grad), _ptr(self.loss), _stream()),
                  "dqn_loss_grad_split16")
            return self.loss, self.grad
        check(lib.marlhip_dqn_loss_grad(ctypes.byref(s), _ptr(self.params), _ptr(self.target), ctypes.byref(bs), float(self.gamma),
                                        self.double_q, int(mode), _ptr(ws), ws.numel(), _ptr(self.grad), _ptr(self.loss), _stream()),
              "dqn_loss_grad")
        return self.loss, self.grad

    def loss_grad_replay(self, replay, batch_size, length=None, idx=None, seed=0, counter=0, idx_out=None, mode=0):
        """Sampling fused into the loss/grad kernel: rows are gathered from the replay in-kernel (no Batch)."""
        if self.split16:  # the split-fp16 gather lives in the n-updates call (FusedLearner); a single step materialises the Batch
            return self.loss_grad(replay.sample(batch_size, length=length, idx=idx, seed=seed, counter=counter), mode=mode)
        ws = self._workspace(replay.T, batch_size)
        s = self.spec.c()
        if self.ret_stats is not None:
            st = self._stats_for(mode, batch_size).c()
            check(lib.marlhip_dqn_loss_grad_std_replay(ctypes.byref(s), _ptr(self.params), _ptr(self.target), ctypes.byref(replay.shape),
                                                       ctypes.byref(replay.bufs), _ptr(idx), int(batch_size), int(length or 0),
                                                       int(seed) & (2**64 - 1), int(counter) & 0xFFFFFFFF, _ptr(idx_out),
                                                       float(self.gamma), self.double_q, int(mode), ctypes.byref(st), _ptr(ws), ws.numel(),
                                                       _ptr(self.grad), _ptr(self.loss), _stream()), "dqn_loss_grad_std_replay")
            return self.loss, self.grad
        check(lib.marlhip_dqn_loss_grad_replay(ctypes.byref(s), _ptr(self.params), _ptr(self.target), ctypes.byref(replay.shape),
                                               ctypes.byref(replay.bufs), _ptr(idx), int(batch_size), int(length or 0),
                                               int(seed) & (2**64 - 1), int(counter) & 0xFFFFFFFF, _ptr(idx_out),
                                               float(self.gamma), self.double_q, int(mode), _ptr(ws), ws.numel(), _ptr(self.grad),
                                               _ptr(self.loss), _stream()), "dqn_loss_grad_replay")
        return self.loss, self.grad

    def apply(self, hard_update=False, tau=0.0, grad_scale=1.0):
        self.step += 1
        _clip_step(self.optimizer, self.params.numel(), self.params, self.grad, self.exp_avg, self.exp_avg_sq, self.target, self.step, self.lr,
                   self.betas, self.eps, self.grad_clip, grad_scale, hard_update, tau, self.scratch, self.gnorm, "dqn_clip_step")



class WideDqnUpdater(DqnUpdater):
    """DqnUpdater for networks without a fused kernel (spec.wide): marlhip_wide_dqn_loss_grad (GEMM path + the TD stage of the
    recurrent learner, its standardise_returns stage included); sampling materialises the Batch first.  IDQN and VDN."""

    def _workspace(self, T, B):
        key = (T, B)
        if key not in self._ws:
            s = self.spec.c()
            n = check(lib.marlhip_wide_dqn_workspace_bytes(ctypes.byref(s), T, B), "wide_dqn_workspace_bytes")
            self._ws = {key: torch.empty(int(n), dtype=torch.uint8, device=self.params.device)}  # one shape alive: these are large
        return self._ws[key]

    def loss_grad(self, batch, mode=0):
        if mode not in (0, 1):
            raise NotImplementedError("QMIX with layers wider than 128: the mixer stage goes with the fused agent kernels")
        T, B = batch.filled.shape
        ws = self._workspace(T, B)
        bs = BatchStruct(batch.obss.data_ptr(), batch.actions.data_ptr(), batch.rewards.data_ptr(), batch.dones.data_ptr(),
                         batch.filled.data_ptr(), T, B, 0, 0, 0, 0, _mask_ptr(batch.action_mask, (self.spec.n_agents, T + 1, B, self.spec.n_actions)))
        s = self.spec.c()
        if self.ret_stats is not None:  # standardise_returns: per-agent statistics (IDQN) or VDNetwork's per-batch-column ones
            st = self._stats_for(mode, B).c()
            check(lib.marlhip_wide_dqn_loss_grad_std(ctypes.byref(s), _ptr(self.params), _ptr(self.target), ctypes.byref(bs), float(self.gamma),
                                                     self.double_q, ctypes.byref(st), _ptr(ws), ws.numel(), _ptr(self.grad), _ptr(self.loss),
                                                     _stream()), "wide_dqn_loss_grad_std")
            return self.loss, self.grad
        check(lib.marlhip_wide_dqn_loss_grad(ctypes.byref(s), _ptr(self.params), _ptr(self.target), ctypes.byref(bs), float(self.gamma),
                                             self.double_q, int(mode), _ptr(ws), ws.numel(), _ptr(self.grad), _ptr(self.loss), _stream()),
              "wide_dqn_loss_grad")
        return self.loss, self.grad

    def loss_grad_replay(self, replay, batch_size, length=None, idx=None, seed=0, counter=0, idx_out=None, mode=0):
        return self.loss_grad(replay.sample(batch_size, length=length, idx=idx, seed=seed, counter=counter), mode=mode)

class QmixUpdater(DqnUpdater):
    """QMixNetwork's learner step (marlbase/dqn/model.py:334-443): agent networks + monotonic mixer.  The mixer block is a
    second flat parameter vector (mixer.parameters() order) with its own Adam moments; `apply` clips the critic gradient
    only (QNetwork.update clips critic.parameters(), model.py:169-170) and steps both with the same Adam step count."""

    def __init__(self, spec: NetSpec, params, target, mixer, target_mixer, mixing=None, **kw):
        super().__init__(spec, params, target, **kw)
        mixing = dict(mixing or dict(embed_dim=64, hypernet_layers=2, hypernet_embed=32))
        self.mixing = (int(mixing["embed_dim"]), int(mixing["hypernet_layers"]), int(mixing["hypernet_embed"]))
        self.l1_fp16 = int(bool(mixing.get("fp16", False)))  # opt-in: the mixers' first layers on the fp16 MFMA (marlhip_qmix_mixer.l1_fp16)
        s = spec.c()
        n = check(lib.marlhip_qmix_nparams(ctypes.byref(s), *self.mixing), "qmix_nparams")
        if mixer.numel() != n or target_mixer.numel() != n:
            raise ValueError(f"mixer block has {mixer.numel()} parameters, the QMixer of this shape has {n}")
        self.mixer, self.target_mixer = mixer, target_mixer
        # critic and mixer gradients share one allocation so that data-parallel training all-reduces them in ONE message
        self.joint_grad = torch.zeros(params.numel() + n, dtype=torch.float32, device=params.device)
        self.grad = self.joint_grad[:params.numel()].view_as(params)
        self.mixer_grad = self.joint_grad[params.numel():]
        self.mixer_exp_avg = torch.zeros_like(mixer)
        self.mixer_exp_avg_sq = torch.zeros_like(mixer)
        self.mixer_scratch = torch.zeros((n + 255) // 256 + 1, dtype=torch.float32, device=mixer.device)

    def _workspace(self, T, B):
        key = (T, B)
        if key not in self._ws:
            s = self.spec.c()
            n = check(lib.marlhip_qmix_workspace_bytes_mx(ctypes.byref(s), ctypes.byref(self._mx_dims()), T, B), "qmix_workspace_bytes")
            self._ws[key] = torch.empty(max(int(n), 4), dtype=torch.uint8, device=self.params.device)
        return self._ws[key]

    def _mx_dims(self):
        """the mixing configuration alone (what the *_workspace_bytes_mx queries read)"""
        return QmixMixer(None, None, None, *self.mixing)

    def _mx(self, B=None):
        mx = QmixMixer(self.mixer.data_ptr(), self.target_mixer.data_ptr(), self.mixer_grad.data_ptr(), *self.mixing)
        mx.l1_fp16 = self.l1_fp16
        if self.ret_stats is not None:  # standardise_returns: per-batch-column statistics (dqn/model.py:415-422)
            self._st_c = self._stats_for(2, B).c()
            mx.ret_stats = ctypes.cast(ctypes.pointer(self._st_c), ctypes.c_void_p)
        return mx

    def loss_grad(self, batch, mode=2):
        T, B = batch.filled.shape
        ws = self._workspace(T, B)
        bs = BatchStruct(batch.obss.data_ptr(), batch.actions.data_ptr(), batch.rewards.data_ptr(), batch.dones.data_ptr(),
                         batch.filled.data_ptr(), T, B, 0, 0, 0, 0, _mask_ptr(batch.action_mask, (self.spec.n_agents, T + 1, B, self.spec.n_actions)))
        s, mx = self.spec.c(), self._mx(B)
        check(lib.marlhip_qmix_loss_grad(ctypes.byref(s), _ptr(self.params), _ptr(self.target), ctypes.byref(mx), ctypes.byref(bs),
                                         float(self.gamma), self.double_q, _ptr(ws), ws.numel(), _ptr(self.grad), _ptr(self.loss),
                                         _stream()), "qmix_loss_grad")
        return self.loss, self.grad

    def loss_grad_replay(self, replay, batch_size, length=None, idx=None, seed=0, counter=0, idx_out=None, mode=2):
        ws = self._workspace(replay.T, batch_size)
        s, mx = self.spec.c(), self._mx(batch_size)
        check(lib.marlhip_qmix_loss_grad_replay(ctypes.byref(s), _ptr(self.params), _ptr(self.target), ctypes.byref(mx),
                                                ctypes.byref(replay.shape), ctypes.byref(replay.bufs), _ptr(idx), int(batch_size),
                                                int(length or 0), int(seed) & (2**64 - 1), int(counter) & 0xFFFFFFFF,
                                                _ptr(idx_out), float(self.gamma), self.double_q, _ptr(ws), ws.numel(),
                                                _ptr(self.grad), _ptr(self.loss), _stream()), "qmix_loss_grad_replay")
        return self.loss, self.grad

    def apply(self, hard_update=False, tau=0.0, grad_scale=1.0):
        super().apply(hard_update, tau, grad_scale)  # critic: clip + Adam + target; advances self.step
        _clip_step(self.optimizer, self.mixer.numel(), self.mixer, self.mixer_grad, self.mixer_exp_avg, self.mixer_exp_avg_sq, self.target_mixer,
                   self.step, self.lr, self.betas, self.eps, 0.0, grad_scale, hard_update, tau, self.mixer_scratch, None, "dqn_clip_step(mixer)")


def ac_forward_rows(spec: NetSpec, params, obs, agent_stride, row_stride, n_rows, value_net=False):
    """out[p][row][:] = MLP_p(obs row); value_net: 1 = the critic shape (one output), 2 = the centralised critic (P*D inputs,
    every agent reads the same concatenated row)."""
    _require_gpu()
    s = spec.c()
    out = torch.empty(spec.n_agents, n_rows, 1 if value_net else spec.n_actions, device=params.device)
    check(lib.marlhip_ac_forward_rows(ctypes.byref(s), int(value_net), _ptr(params), _ptr(obs), int(agent_stride),
                                      int(row_stride), int(n_rows), _ptr(out), *_fwd_ws(spec, params.device), _stream()), "ac_forward_rows")
    return out


class AcUpdater:
    """A2CNetwork.update / PPONetwork.update on the device (marlbase/ac/model.py:189-352): loss + gradient of actor and
    critic from one rollout Batch in the ac/train.py layout, then clip over ALL parameters + Adam on the joint block.
    `block` is ONE flat fp32 tensor [P*n_actor + P*n_critic]: actor blocks first (parameters() order), then critic."""

    def __init__(self, spec: NetSpec, block, target_critic, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, gamma=0.99, n_steps=5,
                 entropy_coef=0.001, value_loss_coef=0.5, grad_clip=False, ppo_clip=0.2, standardise_returns=False,
                 centralised_critic=False, recurrent=False, optimizer="Adam", critic_sharing="actor", critic_n_hidden=None, mixed_rnn=None):
        """mixed_rnn: actor.use_rnn != critic.use_rnn (ac/model.py:45-97 builds the families from their own flags): "actor" = recurrent actors
        next to feed-forward critics, "critic" = the reverse (marlhip_mixed_*; `recurrent` must then be False).
        critic_sharing: the critics' agent -> network map when critic.parameter_sharing differs from actor.parameter_sharing
        (ac/model.py:45-97): "actor" (default) = spec.sharing for both, None = one critic per agent, or a tuple of network indices.
        critic_n_hidden: the critics' number of hidden layers when critic.layers is of another length than actor.layers (GEMM-path shapes)"""
        _require_gpu()
        self.optimizer = optimizer_id(optimizer)
        self.recurrent = bool(recurrent)  # use_rnn actors and critics: the marlhip_gru_* entry points, recurrent block layout
        if mixed_rnn not in (None, "actor", "critic") or (mixed_rnn and (recurrent or centralised_critic or spec.wide)):
            raise ValueError("mixed_rnn: 'actor' or 'critic', with independent fused-width critics and recurrent=False")
        self.mixed_rnn = mixed_rnn
        self.any_recurrent = self.recurrent or mixed_rnn is not None  # (no kept forward pass, no deferred critics, no CU-share stream)
        if mixed_rnn:
            arnn = int(mixed_rnn == "actor")
            self._fn = tuple((lambda s_, *a, _f=f: _f(s_, arnn, *a)) for f in (lib.marlhip_mixed_a2c_loss_grad, lib.marlhip_mixed_ppo_prepare,
                                                                              lib.marlhip_mixed_ppo_loss_grad))
        else:
            self._fn = ((lib.marlhip_gru_a2c_loss_grad, lib.marlhip_gru_ppo_prepare, lib.marlhip_gru_ppo_loss_grad) if self.recurrent else
                        (lib.marlhip_a2c_loss_grad, lib.marlhip_ppo_prepare, lib.marlhip_ppo_loss_grad))
        self.centralised = int(bool(centralised_critic))
        self.ret_stats = RunningReturnStats(spec.n_agents, block.device) if standardise_returns else None
        s = spec.c()
        self.spec = spec
        if self.recurrent:
            self.n_actor = check(lib.marlhip_gru_nparams(ctypes.byref(s)), "gru_nparams")
            sc = spec.c()  # the critics' own number of stacked GRU layers (critic.layers is its own list: ac/model.py:45-97)
            self.critic_n_hidden = 0
            if critic_n_hidden is not None and int(critic_n_hidden) != int(spec.n_hidden):
                self.critic_n_hidden = sc.n_hidden = int(critic_n_hidden)
            self.n_critic = check(lib.marlhip_gru_ac_critic_nparams(ctypes.byref(sc), int(bool(centralised_critic))), "gru_ac_critic_nparams")
        elif mixed_rnn:  # each block in its own family's layout; the recurrent family may be a stack (its own depth), the other has two layers
            self.critic_n_hidden = 0
            if mixed_rnn == "actor":  # spec.n_hidden = len(actor.layers)
                two = spec.c()
                two.n_hidden = 2
                self.n_actor = check(lib.marlhip_gru_nparams(ctypes.byref(s)), "gru_nparams")
                self.n_critic = check(lib.marlhip_ac_critic_nparams(ctypes.byref(two), 0), "ac_critic_nparams")
            else:  # spec.n_hidden = 2 (the actors'); critic_n_hidden = len(critic.layers)
                sc = spec.c()
                if critic_n_hidden is not None and int(critic_n_hidden) != 2:
                    self.critic_n_hidden = sc.n_hidden = int(critic_n_hidden)
                self.n_actor = spec.nparams()
                self.n_critic = check(lib.marlhip_gru_ac_critic_nparams(ctypes.byref(sc), 0), "gru_ac_critic_nparams")
        else:
            self.n_actor = spec.nparams()
            sc = spec.c()
            self.critic_n_hidden = 0
            if critic_n_hidden is not None and int(critic_n_hidden) != int(spec.n_hidden):
                if not spec.wide:
                    raise ValueError("critics of another depth than the actors: the shape must be a GEMM-path one (NetSpec.wide)")
                self.critic_n_hidden = sc.n_hidden = int(critic_n_hidden)
            self.n_critic = check(lib.marlhip_ac_critic_nparams(ctypes.byref(sc), self.centralised), "ac_critic_nparams")
        P = spec.n_blocks
        if isinstance(critic_sharing, str):
            self.critic_sharing = spec.sharing
        else:
            self.critic_sharing = None if critic_sharing is None else tuple(int(k) for k in critic_sharing)
        Kc = spec.n_agents if self.critic_sharing is None else max(self.critic_sharing) + 1
        if block.numel() != P * self.n_actor + Kc * self.n_critic or target_critic.numel() != Kc * self.n_critic:
            raise ValueError("actor-critic parameter block has the wrong size for this shape")
        self.block, self.target_critic = block, target_critic
        self.actor = block[:P * self.n_actor].view(P, self.n_actor)
        self.critic = block[P * self.n_actor:].view(Kc, self.n_critic)
        self.grad = torch.zeros_like(block)
        self.actor_grad = self.grad[:P * self.n_actor].view(P, self.n_actor)
        self.critic_grad = self.grad[P * self.n_actor:].view(Kc, self.n_critic)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(block), torch.zeros_like(block)
        self.scratch = torch.zeros((block.numel() + 255) // 256 + 1, dtype=torch.float32, device=block.device)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=block.device)
        self.metrics = torch.zeros(5, dtype=torch.float32, device=block.device)
        rs = self.ret_stats
        self.cfg = AcConfig(int(n_steps), float(entropy_coef), float(value_loss_coef), float(ppo_clip), float(gamma),
                            rs.mean.data_ptr() if rs else None, rs.var.data_ptr() if rs else None,
                            rs.count_t.data_ptr() if rs else None, self.centralised, None)
        self.cfg.critic_n_hidden = getattr(self, "critic_n_hidden", 0)
        if self.critic_sharing != spec.sharing:  # two agent -> network maps: the critics' rides in the config
            cmap = self.critic_sharing if self.critic_sharing is not None else tuple(range(spec.n_agents))
            if len(cmap) != spec.n_agents or spec.n_agents > 16:
                raise ValueError("critic sharing indices: one per agent, at most 16 agents")
            self.cfg.critic_n_networks = max(cmap) + 1
            for i, k in enumerate(cmap):
                self.cfg.critic_net_of[i] = int(k)
        if self.recurrent:
            # recurrent actors + critics: the critics' sequence passes overlap the actors' on this second stream (joined inside the call)
            self._side = torch.cuda.Stream(device=block.device)
            self.cfg.side_stream = self._side.cuda_stream
        self.lr, self.betas, self.eps = lr, betas, eps
        self.grad_clip = float(grad_clip) if grad_clip else 0.0
        self.step = 0
        self._ws = {}
        self._kept = None  # (T, B, batch obs pointer, rollout generation) of the rollout whose actor forward pass sits in the workspace (ac_collect(keep_for=self))
        # the critics' half of an update on a stream of its own (a2c_loss_grad(defer_critic=True)): the stream, the batch tensors its
        # backward pass still reads, and the event everything that touches the critic blocks next waits for
        self._critic_stream = None
        self._no_defer = False  # set when the device refuses a compute-unit-masked stream (_side_stream)
        self._critic_pending = None
        self._critic_event = None
        self._inflight = None

    # ---- the actors' forward pass kept by the rollout (marlhip_*_ac_collect_keep; include/marlhip.h) ------------------------------
    def can_keep(self, n_envs, actor_params):
        """fused feed-forward actors, envs in whole blocks of 16, and the rollout sampled with THIS updater's actor block"""
        return (not self.any_recurrent and not self.spec.wide and n_envs % 16 == 0 and not _NO_KEEP
                and actor_params.data_ptr() == self.actor.data_ptr())

    # ---- the critics' half of an A2C update next to the following rollout (marlhip_ac_config.defer_critic_backward) -----------------
    def can_defer(self, n_envs=None, force=False):
        """Hard conditions: no joint clip (the optimiser step is then elementwise: the actors' step does not need the critics' gradient),
        feed-forward networks, a caller that is not on the default stream.  Where it PAYS (skipped with force): the critics run on a stream
        that owns half of the compute units, next to the rollout instead of in front of it - that is a gain while the rollout leaves that
        half idle (one workgroup per block of 16 envs: at most CUs / 2 blocks) and the critics' backward pass at half speed is not longer
        than the rollout (measured on the warehouse, 2048 envs x 500 steps, 128-128: independent critics 3.7 ms against a 7.3 ms rollout,
        52.2 -> 59.8 M env-steps/s; the 284-input centralised critics 7 ms, 38.1 -> 31.1 M: a critic row may cost 1.5 x an actor row)."""
        if self.any_recurrent or self.grad_clip or _NO_OVERLAP or self._no_defer:
            return False
        if n_envs is None:
            return True
        # (the compute-unit mask comes with a stream of the legacy blocking kind - hipExtStreamCreateWithCUMask takes no flags - and those
        # synchronise implicitly with the default stream: a caller on the default stream would wait for the critics after all)
        if torch.cuda.current_stream(self.block.device) == torch.cuda.default_stream(self.block.device):
            return False
        if force or _FORCE_OVERLAP:
            return True
        S = self.spec
        cus = torch.cuda.get_device_properties(self.block.device).multi_processor_count
        row = lambda d, a: d * S.hidden + S.hidden * S.hidden + S.hidden * a  # noqa: E731 - multiply-adds of one row through D-H-H-A
        dc = S.n_agents * S.obs_dim if self.centralised else S.obs_dim
        return not S.wide and (int(n_envs) + 15) // 16 <= cus // 2 and row(dc, 1) <= 1.5 * row(S.obs_dim, S.n_actions)

    def _side_stream(self):
        """the stream that owns half of the compute units, or None when the runtime refuses one (a virtualised / partitioned device, an
        older runtime: hipExtStreamCreateWithCUMask fails) - remembered, warned about once, and the update then stays on the caller's
        stream (the same launches, the same bits)"""
        if self._critic_stream is None and not self._no_defer:
            try:
                self._critic_stream = _half_chip_stream(self.block.device)
            except Exception as e:  # noqa: BLE001 - MarlHipError from check(), or whatever torch raises wrapping the handle
                import logging

                logging.getLogger(__name__).warning("marlhip: no compute-unit-masked stream on this device (%s); the critics' half of an A2C "
                                                    "update stays on the caller's stream", e)
                self._no_defer = True
        return self._critic_stream

    def probe_defer(self):
        """True when the critics' half of an update could run on a stream of its own here (hard conditions of can_defer + the stream
        exists).  Data-parallel jobs vote over this at set-up (A2CNetwork.attach_grad_sync): every rank defers or none does."""
        return self.can_defer() and self._side_stream() is not None

    def sync_critic(self):
        """order the current stream behind the critics' deferred work (no-op when none is in flight) and release the batch it read"""
        if self._critic_pending is not None:
            self.finish_critic()
        if self._critic_event is not None:
            torch.cuda.current_stream(self.block.device).wait_event(self._critic_event)
            self._critic_event = None
            self._inflight = None

    def critic_stream(self):
        """context manager: torch ops on the critics' blocks that belong behind the deferred step (the target update); the current stream
        when nothing is deferred"""
        return torch.cuda.stream(self._critic_stream if self._critic_pending is not None else torch.cuda.current_stream(self.block.device))

    def finish_critic(self):
        """after the critics' step and target update have been queued: mark the point later readers wait for"""
        if self._critic_pending is not None:
            self._critic_event = torch.cuda.Event()
            self._critic_event.record(self._critic_stream)
            self._inflight, self._critic_pending = self._critic_pending, None

    def attach_exchange(self, reduce):
        """data-parallel training with standardise_returns: batch moments summed over the ranks (marlhip_ac_config.ret_exchange)"""
        if self.ret_stats is not None and reduce is not None:
            self.ret_stats.attach_exchange(reduce)
            self.cfg.ret_exchange = self.ret_stats.exchange.ptr
            self.cfg.ret_moments = self.ret_stats.moments.data_ptr()

    def _workspace(self, T, B):
        if (T, B) not in self._ws:
            s = self.spec.c()
            if self.mixed_rnn:
                n = check(lib.marlhip_mixed_ac_workspace_bytes_lc(ctypes.byref(s), int(self.mixed_rnn == "actor"), self.critic_n_hidden, T, B), "mixed_ac_workspace_bytes")
            elif self.recurrent:
                n = check(lib.marlhip_gru_ac_workspace_bytes_lc(ctypes.byref(s), self.centralised, self.critic_n_hidden, T, B), "gru_ac_workspace_bytes")
            else:
                n = check(lib.marlhip_ac_workspace_bytes_lc(ctypes.byref(s), self.centralised, self.critic_n_hidden, T, B), "ac_workspace_bytes")
            self._ws[(T, B)] = torch.empty(int(n), dtype=torch.uint8, device=self.block.device)
        return self._ws[(T, B)]

    def _batch(self, batch):
        """ac/train.py Batch (obss [T+1,N,P*D], actions i64 [T,N,P], rewards [T,N,P], dones [T+1,N], filled [T,N])"""
        T, N = batch.filled.shape
        P, D = self.spec.n_agents, self.spec.obs_dim
        dones = batch.dones if batch.dones.dtype == torch.float32 else batch.dones.float()  # model.py:198
        keep = (batch.obss.contiguous(), batch.actions.contiguous(), batch.rewards.contiguous(), dones.contiguous(),
                batch.filled.contiguous())
        masks = getattr(batch, "action_masks", None)  # ac/train.py:53-63: [T+1][N][P][A] f32 or None
        if masks is not None:
            masks = masks.to(torch.float32).contiguous()
            keep = keep + (masks,)
        bs = BatchStruct(keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(), T, N,
                         D, P * D, 1, P, _mask_ptr(masks, (T + 1, N, P, self.spec.n_actions)))
        return bs, keep, T, N

    def a2c_loss_grad(self, batch, kept=True, defer_critic=False):
        """kept: let the step read the actors' logits and hidden layers the collector left (ac_collect(keep_for=self)) instead of running the
        pass again - taken only when the batch is that rollout's (same T x N, same observation tensor) and the parameters have not moved
        since (apply() voids the record); False: always recompute.
        defer_critic: the critics' backward pass goes to a stream of its own (where can_defer() allows: no joint clip, a rollout that leaves
        half of the chip idle, a caller that is NOT on the default stream) - the apply() that follows then
        steps the actors on the current stream and the critics there, and the caller ends the update with `with critic_stream(): <target
        update>; finish_critic()`.  critic_grad, the critic blocks and the batch tensors must not be touched from the current stream
        before sync_critic() (the next a2c_loss_grad / ppo_* call does it)."""
        self.sync_critic()
        bs, keep, T, N = self._batch(batch)
        ws, s = self._workspace(T, N), self.spec.c()
        defer = bool(defer_critic) and self.can_defer(N, force=defer_critic == "force") and self._side_stream() is not None
        if defer:
            self.cfg.side_stream = self._critic_stream.cuda_stream
            self.cfg.defer_critic_backward = 1
        try:
            with self._kept_scope(kept, bs, keep, T, N):
                check(self._fn[0](ctypes.byref(s), _ptr(self.actor), _ptr(self.critic), _ptr(self.target_critic),
                                                ctypes.byref(bs), ctypes.byref(self.cfg), _ptr(ws), ws.numel(), _ptr(self.actor_grad),
                                                _ptr(self.critic_grad), _ptr(self.metrics), _stream()), "a2c_loss_grad")
        finally:
            if defer:
                self.cfg.side_stream = None
                self.cfg.defer_critic_backward = 0
        if defer:
            self._critic_pending = keep  # the batch tensors the critics' backward pass reads stay alive until its event has been waited for
        return self.metrics

    def _kept_scope(self, kept, bs, keep, T, N):
        """marlhip_ac_config.actor_forward_kept for ONE library call: set when the batch is the kept rollout's and the parameters are still
        the ones it was sampled with; `last_step_used_kept_forward` says what the call did"""
        use = (bool(kept) and self._kept is not None and self._kept == (T, N, keep[0].data_ptr(), _ROLLOUT_GEN.get(keep[0].data_ptr()))
               and not bs.action_mask)
        up = self

        class _Scope:
            def __enter__(self):
                up.cfg.actor_forward_kept = int(use)
                up.last_step_used_kept_forward = bool(use)

            def __exit__(self, *exc):
                up.cfg.actor_forward_kept = 0
                return False

        return _Scope()

    def ppo_prepare(self, batch, kept=True):
        """kept: as a2c_loss_grad's - the old log-probs come from the logits the collector sampled the actions with (the same bits)"""
        self.sync_critic()
        bs, keep, T, N = self._batch(batch)
        ws, s = self._workspace(T, N), self.spec.c()
        with self._kept_scope(kept, bs, keep, T, N):
            check(self._fn[1](ctypes.byref(s), _ptr(self.actor), _ptr(self.critic), _ptr(self.target_critic),
                                          ctypes.byref(bs), ctypes.byref(self.cfg), _ptr(ws), ws.numel(), _stream()), "ppo_prepare")

    def ppo_loss_grad(self, batch, kept=True):
        """kept: the FIRST epoch of a rollout runs on the parameters it was sampled with - its actor forward pass is the collector's; apply()
        voids the record, the later epochs run the pass themselves"""
        self.sync_critic()
        bs, keep, T, N = self._batch(batch)
        ws, s = self._workspace(T, N), self.spec.c()
        with self._kept_scope(kept, bs, keep, T, N):
            check(self._fn[2](ctypes.byref(s), _ptr(self.actor), _ptr(self.critic), ctypes.byref(bs), ctypes.byref(self.cfg),
                                            _ptr(ws), ws.numel(), _ptr(self.actor_grad), _ptr(self.critic_grad), _ptr(self.metrics),
                                            _stream()), "ppo_loss_grad")
        return self.metrics

    def apply(self, grad_scale=1.0):
        """clip_grad_norm_(self.parameters(), grad_clip) + optimizer.step() (model.py:227-231): one norm over actor + critic"""
        self.step += 1
        self._kept = None  # the parameters move: a kept forward pass is no longer this network's
        if self._critic_pending is not None:
            # the critics' gradient is still being formed on their stream: without a clip the step is elementwise, so the actors' slice
            # steps here (the next rollout needs nothing else) and the critics' slice there, behind its backward pass - the same bits as
            # one launch over the joint block
            na = self.actor.numel()
            _clip_step(self.optimizer, na, self.block[:na], self.grad[:na], self.exp_avg[:na], self.exp_avg_sq[:na], None, self.step, self.lr,
                       self.betas, self.eps, 0.0, grad_scale, False, 0.0, self.scratch, None, "dqn_clip_step(actor)")
            with torch.cuda.stream(self._critic_stream):
                _clip_step(self.optimizer, self.block.numel() - na, self.block[na:], self.grad[na:], self.exp_avg[na:], self.exp_avg_sq[na:], None,
                           self.step, self.lr, self.betas, self.eps, 0.0, grad_scale, False, 0.0, self.scratch, None, "dqn_clip_step(critic)")
            return
        _clip_step(self.optimizer, self.block.numel(), self.block, self.grad, self.exp_avg, self.exp_avg_sq, None, self.step, self.lr, self.betas,
                   self.eps, self.grad_clip, grad_scale, False, 0.0, self.scratch, self.gnorm if self.grad_clip else None,  # (no clip: no norm launch)
                   "dqn_clip_step(actor+critic)")


def idqn_collect(cfg, spec: NetSpec, params, epsilon, round_idx, replay: DeviceReplay, slot_base, fin_return,
                 fin_length, write_replay=True, clear_stale=False, use_proper_termination=False):
    """Fused collector: one launch = one round of N episodes (reset -> T x (act, step, add))."""
    _require_gpu()
    s = spec.c()
    fn = lib.marlhip_rware_idqn_collect if is_rware(cfg) else lib.marlhip_idqn_collect
    check(fn(ctypes.byref(cfg), ctypes.byref(s), _ptr(params), float(epsilon), int(round_idx) & 0xFFFFFFFF,
                                   ctypes.byref(replay.shape), ctypes.byref(replay.bufs), int(slot_base), int(bool(write_replay)),
                                   int(bool(clear_stale)), int(bool(use_proper_termination)), _ptr(fin_return), _ptr(fin_length),
                                   *_fwd_ws(spec, params.device), _stream()), "idqn_collect")


class FusedLearner:
    """n updates per host call (marlhip_idqn_update_n): sample -> loss/grad -> clip+Adam+target, with the
    reference's update / target-update counters kept in host integers shared with the QNetwork object."""

    def __init__(self, updater: DqnUpdater, replay: DeviceReplay, batch, target_update_interval_or_tau, mode=0,
                 materialise_batch=False):
        self.up, self.replay, self.B = updater, replay, int(batch)
        outs = replay._outputs(self.B)
        ws = updater._workspace(replay.T, self.B)
        self.c = IdqnLearner(
            net=updater.spec.c(), rs=replay.shape, rb=replay.bufs, params=updater.params.data_ptr(),
            target=updater.target.data_ptr(), exp_avg=updater.exp_avg.data_ptr(), exp_avg_sq=updater.exp_avg_sq.data_ptr(),
            grad=updater.grad.data_ptr(), loss=updater.loss.data_ptr(), scratch=updater.scratch.data_ptr(),
            gnorm=updater.gnorm.data_ptr(), workspace=ws.data_ptr(), workspace_bytes=ws.numel(),
            obss=outs[0].data_ptr(), actions=outs[1].data_ptr(), rewards=outs[2].data_ptr(), dones=outs[3].data_ptr(),
            filled=outs[4].data_ptr(), idx=outs[5].data_ptr(), batch=self.B, double_q=updater.double_q, mode=int(mode),
            materialise_batch=int(bool(materialise_batch)),
            gamma=float(updater.gamma), max_norm=float(updater.grad_clip), lr=float(updater.lr), beta1=float(updater.betas[0]),
            beta2=float(updater.betas[1]), eps=float(updater.eps),
            target_update_interval_or_tau=float(target_update_interval_or_tau))
        self._keep = (outs, ws)

    def run(self, n_updates, length, seed, counter0, updates, last_target_update, grad_sync=None, world=1):
        """returns the advanced (updates, last_target_update); the Adam step lives in the DqnUpdater.  `grad_sync(grad)` (an
        in-place all-reduce SUM over `world` ranks): the data-parallel form, marlhip_idqn_update_n_dist - the loop over the updates
        stays in the library, the exchange is its only host hop, clip + Adam take 1 / world and the post-reduce norm."""
        step = ctypes.c_int64(self.up.step)
        upd = ctypes.c_int64(int(updates))
        last = ctypes.c_int64(int(last_target_update))
        if self.up.split16:
            if grad_sync is not None or self.c.mode != 0:
                raise NotImplementedError("split16: the opt-in learner is single-process IDQN")
            check(lib.marlhip_idqn_update_n_split16(ctypes.byref(self.c), int(n_updates), int(length), int(seed) & (2**64 - 1),
                                                    int(counter0) & 0xFFFFFFFF, ctypes.byref(step), ctypes.byref(upd), ctypes.byref(last),
                                                    _stream()), "idqn_update_n_split16")
        elif grad_sync is None:
            check(lib.marlhip_idqn_update_n(ctypes.byref(self.c), int(n_updates), int(length), int(seed) & (2**64 - 1),
                                            int(counter0) & 0xFFFFFFFF, ctypes.byref(step), ctypes.byref(upd), ctypes.byref(last),
                                            _stream()), "idqn_update_n")
        elif getattr(grad_sync, "c_fn", None) is not None:
            # the in-library exchange (parallel.P2PExchange -> marlhip_p2p_allreduce): a C function pointer, the loop never leaves the library
            check(lib.marlhip_idqn_update_n_dist(ctypes.byref(self.c), int(n_updates), int(length), int(seed) & (2**64 - 1),
                                                 int(counter0) & 0xFFFFFFFF, ctypes.byref(step), ctypes.byref(upd), ctypes.byref(last),
                                                 grad_sync.c_fn, grad_sync.c_ctx, int(world), _stream()), "idqn_update_n_dist")
        else:
            ex = getattr(self, "_exchange", None)
            if ex is None or ex.reduce is not grad_sync:
                ex = self._exchange = Exchange(self.up.grad, grad_sync)
            rc = lib.marlhip_idqn_update_n_dist(ctypes.byref(self.c), int(n_updates), int(length), int(seed) & (2**64 - 1),
                                                int(counter0) & 0xFFFFFFFF, ctypes.byref(step), ctypes.byref(upd), ctypes.byref(last),
                                                ex.ptr, None, int(world), _stream())
            ex.check()
            check(rc, "idqn_update_n_dist")
        self.up.step = step.value
        return upd.value, last.value


class FusedQmixLearner:
    """n QMIX updates per host call (marlhip_qmix_update_n): the per-update host loop of the trainer - loss/grad with the in-kernel
    gather, the joint [critic | mixer] gradient exchange, the two optimiser steps, target copies - behind one library call, with the
    same exchange hook as FusedLearner (a C function pointer for the in-library exchange, a ctypes callback otherwise)."""

    def __init__(self, updater, replay: DeviceReplay, batch, target_update_interval_or_tau):
        self.up, self.replay, self.B = updater, replay, int(batch)
        outs = replay._outputs(self.B)
        ws = updater._workspace(replay.T, self.B)
        base = IdqnLearner(
            net=updater.spec.c(), rs=replay.shape, rb=replay.bufs, params=updater.params.data_ptr(),
            target=updater.target.data_ptr(), exp_avg=updater.exp_avg.data_ptr(), exp_avg_sq=updater.exp_avg_sq.data_ptr(),
            grad=updater.grad.data_ptr(), loss=updater.loss.data_ptr(), scratch=updater.scratch.data_ptr(),
            gnorm=updater.gnorm.data_ptr(), workspace=ws.data_ptr(), workspace_bytes=ws.numel(),
            obss=outs[0].data_ptr(), actions=outs[1].data_ptr(), rewards=outs[2].data_ptr(), dones=outs[3].data_ptr(),
            filled=outs[4].data_ptr(), idx=outs[5].data_ptr(), batch=self.B, double_q=updater.double_q, mode=2, materialise_batch=0,
            gamma=float(updater.gamma), max_norm=float(updater.grad_clip), lr=float(updater.lr), beta1=float(updater.betas[0]),
            beta2=float(updater.betas[1]), eps=float(updater.eps), target_update_interval_or_tau=float(target_update_interval_or_tau))
        mx = updater._mx(self.B)
        # the mixer struct is copied BY VALUE below; with standardise_returns its ret_stats field points at the ctypes struct _mx() built
        # for this call (updater._st_c, replaced by any later _mx()) and at the statistics tensors behind it: hold both for the C loop
        st_c = getattr(updater, "_st_c", None) if updater.ret_stats is not None else None
        self.c = QmixLearner(base=base, mixer=mx, mixer_rw=updater.mixer.data_ptr(), target_mixer_rw=updater.target_mixer.data_ptr(),
                             mixer_exp_avg=updater.mixer_exp_avg.data_ptr(), mixer_exp_avg_sq=updater.mixer_exp_avg_sq.data_ptr(),
                             mixer_scratch=updater.mixer_scratch.data_ptr(), optimizer=int(updater.optimizer))
        self._keep = (outs, ws, mx, st_c, updater.ret_stats, getattr(updater, "_stats", None))

    def run(self, n_updates, length, seed, counter0, updates, last_target_update, grad_sync=None, world=1):
        step = ctypes.c_int64(self.up.step)
        upd = ctypes.c_int64(int(updates))
        last = ctypes.c_int64(int(last_target_update))
        args = (ctypes.byref(self.c), int(n_updates), int(length), int(seed) & (2**64 - 1), int(counter0) & 0xFFFFFFFF, ctypes.byref(step),
                ctypes.byref(upd), ctypes.byref(last))
        if grad_sync is None:
            check(lib.marlhip_qmix_update_n(*args, None, None, 1, _stream()), "qmix_update_n")
        elif getattr(grad_sync, "c_fn", None) is not None:
            check(lib.marlhip_qmix_update_n(*args, grad_sync.c_fn, grad_sync.c_ctx, int(world), _stream()), "qmix_update_n")
        else:
            ex = getattr(self, "_exchange", None)
            if ex is None or ex.reduce is not grad_sync:
                ex = self._exchange = Exchange(self.up.joint_grad, grad_sync)
            rc = lib.marlhip_qmix_update_n(*args, ex.ptr, None, int(world), _stream())
            ex.check()
            check(rc, "qmix_update_n")
        self.up.step = step.value
        return upd.value, last.value


_ROLLOUT_GEN = {}  # batch observation tensor (device pointer) -> serial number of the last rollout written into it (ac_collect)
_ROLLOUT_COUNTER = __import__("itertools").count(1)


def ac_collect(cfg, spec: NetSpec, actor_params, round_idx, max_len, use_proper_termination, b_obs, b_act, b_rew,
               b_done, b_filled, fin_return, fin_length, t_max, keep_for=None):
    """Fused actor-critic rollout collector (marlbase/ac/train.py:24-119): one launch = one collection call.
    keep_for: the AcUpdater whose A2C step will run on this rollout - when the rollout is sampled with ITS actor block, the collector leaves
    the actors' logits and hidden layers in the updater's workspace (marlhip_*_ac_collect_keep) and `a2c_loss_grad` on these batch tensors
    reads them instead of running the pass again (AcUpdater.can_keep says for which shapes).  Returns True when the pass was kept."""
    _require_gpu()
    s = spec.c()
    # every rollout written into these batch tensors gets a new generation number: a kept record only matches the LAST rollout that wrote
    # them, whoever collected it (another updater, keep_for=None, another actor block - ADVICE r5)
    gen = _ROLLOUT_GEN[b_obs.data_ptr()] = next(_ROLLOUT_COUNTER)
    if len(_ROLLOUT_GEN) > 4096:  # (fresh batch tensors per rollout reuse the allocator's blocks; this only bounds a pathological caller)
        _ROLLOUT_GEN.clear()
        _ROLLOUT_GEN[b_obs.data_ptr()] = gen
    args = (ctypes.byref(cfg), ctypes.byref(s), _ptr(actor_params), int(round_idx) & 0xFFFFFFFF, int(max_len),
            int(bool(use_proper_termination)), _ptr(b_obs), _ptr(b_act), _ptr(b_rew), _ptr(b_done),
            _ptr(b_filled), _ptr(fin_return), _ptr(fin_length), _ptr(t_max), *_fwd_ws(spec, actor_params.device))
    if keep_for is not None and keep_for.can_keep(cfg.n_envs, actor_params):
        ws = keep_for._workspace(int(max_len), int(cfg.n_envs))
        fn = lib.marlhip_rware_ac_collect_keep if is_rware(cfg) else lib.marlhip_ac_collect_keep
        keep_for._kept = None
        check(fn(*args, keep_for.centralised, _ptr(ws), ws.numel(), _stream()), "ac_collect_keep")
        keep_for._kept = (int(max_len), int(cfg.n_envs), b_obs.data_ptr(), gen)
        return True
    if keep_for is not None:
        keep_for._kept = None  # this rollout leaves no record: one of an earlier rollout into the same batch tensors must not match it
    fn = lib.marlhip_rware_ac_collect if is_rware(cfg) else lib.marlhip_ac_collect
    check(fn(*args, _stream()), "ac_collect")
    return False


def ac_store_step(env, t, proper, running, acts, b_obs, b_act, b_rew, b_done, b_fill, fin_ret, fin_len, later=None):
    """marlhip_ac_store_step: one step's bookkeeping of a modular rollout after env.step() (ac/train.py:90-110) in one library call;
    `later` = (count int32[1], returns [cap][P], meta int32 [cap][3]) or None"""
    cnt, lret, lmeta = later if later is not None else (None, None, None)
    check(lib.marlhip_ac_store_step(env.N, env.P, env.D, int(t), int(bool(proper)), _ptr(running), _ptr(env.obs), _ptr(acts), _ptr(env.rewards),
                                    _ptr(env.done), _ptr(env.truncated), _ptr(env.fin_return), _ptr(env.fin_length), _ptr(b_obs[t + 1]), _ptr(b_act[t]),
                                    _ptr(b_rew[t]), _ptr(b_done[t + 1]), _ptr(b_fill[t]), _ptr(fin_ret), _ptr(fin_len), _ptr(cnt),
                                    int(lret.shape[0]) if lret is not None else 0, _ptr(lret), _ptr(lmeta), _stream()), "ac_store_step")


def ac_collect_later_episodes(cfg, spec: NetSpec, actor_params, round_idx, max_len, env_ids, t_start, t_stop, cap=8):
    """The second pass of a rollout (marlhip_ac_collect_later_episodes): the envs `env_ids` (int32 device tensor) whose first episode
    ended at `t_start` < t_stop keep stepping, as the reference's auto-resetting vector env does until the last env has finished
    (ac/train.py:71,101-110).  Returns (returns [n][cap][P], meta [n][cap][2] = (length, finishing step), count [n])."""
    _require_gpu()
    n, dev = int(env_ids.numel()), actor_params.device
    ret = torch.zeros(n, cap, spec.n_agents, device=dev)
    meta = torch.zeros(n, cap, 2, dtype=torch.int32, device=dev)
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    s = spec.c()
    fn = lib.marlhip_rware_ac_collect_later_episodes if is_rware(cfg) else lib.marlhip_ac_collect_later_episodes
    check(fn(ctypes.byref(cfg), ctypes.byref(s), _ptr(actor_params), int(round_idx) & 0xFFFFFFFF, int(max_len), _ptr(env_ids), _ptr(t_start), n,
             int(t_stop), int(cap), _ptr(ret), _ptr(meta), _ptr(cnt), *_fwd_ws(spec, dev), _stream()), "ac_collect_later_episodes")
    return ret, meta, cnt


def _gru_fwd_ws(spec, steps, batch, device):
    """(pointer, bytes) for marlhip_gru_forward / marlhip_gru_ac_forward: one layer -> the shared pack workspace (_fwd_ws); a stack of GRU layers
    chains them through scratch behind the packs (marlhip_gru_forward_workspace_bytes), one buffer per (shape, depth, steps, batch, stream)"""
    if int(spec.n_hidden) <= 2:
        return _fwd_ws(spec, device)
    key = ("gru", spec.n_agents, spec.obs_dim, spec.hidden, int(spec.n_hidden), int(steps), int(batch), torch.device(device).index, torch.cuda.current_stream().cuda_stream)
    ws = _FWD_WS.get(key)
    if ws is None:
        s = spec.c()
        n = check(lib.marlhip_gru_forward_workspace_bytes(ctypes.byref(s), int(steps), int(batch)), "gru_forward_workspace_bytes")
        ws = _FWD_WS[key] = torch.empty(n, dtype=torch.uint8, device=device)
    return ctypes.c_void_p(ws.data_ptr()), ws.numel()


def gru_forward(spec: NetSpec, params, obs, h_in=None, want_h=False, record=None):
    """recurrent Q-networks (use_rnn): obs f32 [P][S][B][D] on the device -> q [P][S][B][A] (and the final hidden state
    [P][B][H] when want_h); h_in [P][B][H] or None (zeros).  L = spec.n_hidden - 1 > 1 stacked GRU layers: hidden states [L][P][B][H].
    `record`: float tensor of marlhip_gru_record_floats for BPTT."""
    _require_gpu()
    P, S, B, D = obs.shape
    L = max(int(spec.n_hidden), 2) - 1
    assert obs.is_contiguous() and obs.dtype == torch.float32 and D == spec.obs_dim and P == spec.n_agents
    hshape = (P, B, spec.hidden) if L == 1 else (L, P, B, spec.hidden)
    if h_in is not None:
        assert tuple(h_in.shape) == hshape and h_in.is_contiguous() and h_in.dtype == torch.float32, (tuple(h_in.shape), hshape)
    q = torch.empty(P, S, B, spec.n_actions, device=obs.device)
    h_out = torch.empty(*hshape, device=obs.device) if want_h else None
    s = spec.c()
    check(lib.marlhip_gru_forward(ctypes.byref(s), _ptr(params), _ptr(obs), S, B, _ptr(h_in), _ptr(h_out), _ptr(q), _ptr(record),
                                  *_gru_fwd_ws(spec, S, B, params.device), _stream()), "gru_forward")
    return (q, h_out) if want_h else q


def gru_nparams(spec: NetSpec):
    s = spec.c()
    return check(lib.marlhip_gru_nparams(ctypes.byref(s)), "gru_nparams")


def gru_loss_grad(spec: NetSpec, params, target, batch, gamma=0.99, double_q=True, mode=0, grad=None, loss=None, ws_cache=None):
    """loss / gradient of the recurrent DQN-family learners (marlhip_gru_loss_grad); batch = hip.Batch on the device.
    `ws_cache`: a dict owned by the caller (GruUpdater keeps one per instance) holding the workspace per (T, B); without
    it the workspace lives for this call only."""
    _require_gpu()
    T, B = batch.filled.shape
    s = spec.c()
    n = check(lib.marlhip_gru_workspace_bytes(ctypes.byref(s), T, B), "gru_workspace_bytes")
    key = ("gru", T, B)
    ws = ws_cache.get(key) if ws_cache is not None else None
    if ws is None or ws.numel() != n or ws.device != params.device:
        ws = torch.empty(n, dtype=torch.uint8, device=params.device)
        if ws_cache is not None:
            ws_cache[key] = ws
    grad = torch.empty_like(params) if grad is None else grad
    loss = torch.empty(2, device=params.device) if loss is None else loss
    bs = BatchStruct(batch.obss.data_ptr(), batch.actions.data_ptr(), batch.rewards.data_ptr(), batch.dones.data_ptr(),
                     batch.filled.data_ptr(), T, B, 0, 0, 0, 0, _mask_ptr(batch.action_mask, (spec.n_agents, T + 1, B, spec.n_actions)))
    check(lib.marlhip_gru_loss_grad(ctypes.byref(s), _ptr(params), _ptr(target), ctypes.byref(bs), float(gamma), int(bool(double_q)), int(mode),
                                    _ptr(ws), ws.numel(), _ptr(grad), _ptr(loss), _stream()), "gru_loss_grad")
    return loss, grad


class GruUpdater(DqnUpdater):
    """DqnUpdater for the recurrent networks: same buffers / clip+Adam / target update, marlhip_gru_loss_grad for the step."""

    def loss_grad(self, batch, mode=0):
        if self.ret_stats is not None:  # standardise_returns: per-agent statistics (IDQN) or VDNetwork's per-batch-column ones
            T, B = batch.filled.shape
            s, st = self.spec.c(), self._stats_for(mode, B).c()
            n = check(lib.marlhip_gru_workspace_bytes(ctypes.byref(s), T, B), "gru_workspace_bytes")
            if self._ws.get("gru_std_n") != n:
                self._ws = {"gru_std_n": n, "buf": torch.empty(n, dtype=torch.uint8, device=self.params.device)}
            ws = self._ws["buf"]
            bs = BatchStruct(batch.obss.data_ptr(), batch.actions.data_ptr(), batch.rewards.data_ptr(), batch.dones.data_ptr(),
                             batch.filled.data_ptr(), T, B, 0, 0, 0, 0, _mask_ptr(batch.action_mask, (self.spec.n_agents, T + 1, B, self.spec.n_actions)))
            check(lib.marlhip_gru_loss_grad_std(ctypes.byref(s), _ptr(self.params), _ptr(self.target), ctypes.byref(bs), float(self.gamma),
                                                self.double_q, ctypes.byref(st), _ptr(ws), ws.numel(), _ptr(self.grad), _ptr(self.loss), _stream()),
                  "gru_loss_grad_std")
            return self.loss, self.grad
        return gru_loss_grad(self.spec, self.params, self.target, batch, gamma=self.gamma, double_q=self.double_q, mode=mode,
                             grad=self.grad, loss=self.loss, ws_cache=self._gru_ws_cache)

    def loss_grad_replay(self, replay, batch_size, length=None, idx=None, seed=0, counter=0, idx_out=None, mode=0):
        """the recurrent learner reads a materialised Batch: sample kernel first (no in-kernel gather)"""
        return self.loss_grad(replay.sample(batch_size, length=length, idx=idx, seed=seed, counter=counter), mode=mode)


def act_from_q(q, epsilon, seed, episode, ep_length, actions=None):
    """joint epsilon-greedy of QNetwork.act (dqn/model.py:105-115) from given values q [P][N][A]: ONE Philox uniform per env decides
    random-vs-greedy for the whole joint action, greedy = first maximum; same noise words as marlhip_dqn_act / the fused collector"""
    _require_gpu()
    P, N, A = q.shape
    actions = torch.empty(P, N, dtype=torch.int32, device=q.device) if actions is None else actions
    check(lib.marlhip_act_from_q(P, N, A, _ptr(q.contiguous()), float(epsilon), int(seed) & (2**64 - 1), _ptr(episode), _ptr(ep_length),
                                 _ptr(actions), _stream()), "act_from_q")
    return actions


class GruQmixUpdater(QmixUpdater):
    """QmixUpdater with recurrent agent networks: marlhip_gru_qmix_loss_grad for the step, everything else inherited
    (joint critic | mixer gradient buffer, critic-only clipping, one Adam step count)."""

    def _gru_ws(self, T, B):
        key = ("gru", T, B)
        if key not in self._ws:
            s = self.spec.c()
            n = check(lib.marlhip_gru_qmix_workspace_bytes_mx(ctypes.byref(s), ctypes.byref(self._mx_dims()), T, B), "gru_qmix_workspace_bytes")
            self._ws.clear()
            self._ws[key] = torch.empty(int(n), dtype=torch.uint8, device=self.params.device)
        return self._ws[key]

    def loss_grad(self, batch, mode=2):
        T, B = batch.filled.shape
        ws = self._gru_ws(T, B)
        bs = BatchStruct(batch.obss.data_ptr(), batch.actions.data_ptr(), batch.rewards.data_ptr(), batch.dones.data_ptr(),
                         batch.filled.data_ptr(), T, B, 0, 0, 0, 0, _mask_ptr(batch.action_mask, (self.spec.n_agents, T + 1, B, self.spec.n_actions)))
        s, mx = self.spec.c(), self._mx(B)
        check(lib.marlhip_gru_qmix_loss_grad(ctypes.byref(s), _ptr(self.params), _ptr(self.target), ctypes.byref(mx), ctypes.byref(bs),
                                             float(self.gamma), self.double_q, _ptr(ws), ws.numel(), _ptr(self.grad), _ptr(self.loss),
                                             _stream()), "gru_qmix_loss_grad")
        return self.loss, self.grad

    def loss_grad_replay(self, replay, batch_size, length=None, idx=None, seed=0, counter=0, idx_out=None, mode=2):
        return self.loss_grad(replay.sample(batch_size, length=length, idx=idx, seed=seed, counter=counter), mode=mode)


class WideQmixUpdater(QmixUpdater):
    """QmixUpdater with agent networks on the GEMM path (spec.wide): marlhip_wide_qmix_loss_grad for the step, everything else inherited"""

    def _wide_ws(self, T, B):
        key = ("wide", T, B)
        if key not in self._ws:
            s = self.spec.c()
            n = check(lib.marlhip_wide_qmix_workspace_bytes_mx(ctypes.byref(s), ctypes.byref(self._mx_dims()), T, B), "wide_qmix_workspace_bytes")
            self._ws.clear()
            self._ws[key] = torch.empty(int(n), dtype=torch.uint8, device=self.params.device)
        return self._ws[key]

    def loss_grad(self, batch, mode=2):
        T, B = batch.filled.shape
        ws = self._wide_ws(T, B)
        bs = BatchStruct(batch.obss.data_ptr(), batch.actions.data_ptr(), batch.rewards.data_ptr(), batch.dones.data_ptr(),
                         batch.filled.data_ptr(), T, B, 0, 0, 0, 0, _mask_ptr(batch.action_mask, (self.spec.n_agents, T + 1, B, self.spec.n_actions)))
        s, mx = self.spec.c(), self._mx(B)
        check(lib.marlhip_wide_qmix_loss_grad(ctypes.byref(s), _ptr(self.params), _ptr(self.target), ctypes.byref(mx), ctypes.byref(bs),
                                              float(self.gamma), self.double_q, _ptr(ws), ws.numel(), _ptr(self.grad), _ptr(self.loss),
                                              _stream()), "wide_qmix_loss_grad")
        return self.loss, self.grad

    def loss_grad_replay(self, replay, batch_size, length=None, idx=None, seed=0, counter=0, idx_out=None, mode=2):
        return self.loss_grad(replay.sample(batch_size, length=length, idx=idx, seed=seed, counter=counter), mode=mode)


def gru_ac_forward(spec: NetSpec, params, obs, agent_stride, row_stride, steps, batch, value_net=False, h_in=None, want_h=False):
    """sequence forward of recurrent actors (logits [P][steps][B][A]) or critics (values [P][steps][B][1]); rows of agent p at
    obs + p * agent_stride + (t * B + b) * row_stride; hidden state [P][B][H] in / out ([L][P][B][H] for L = spec.n_hidden - 1 > 1 stacked layers)"""
    _require_gpu()
    P = spec.n_agents
    L = max(int(spec.n_hidden), 2) - 1
    hshape = (P, batch, spec.hidden) if L == 1 else (L, P, batch, spec.hidden)
    if h_in is not None:
        assert tuple(h_in.shape) == hshape and h_in.is_contiguous() and h_in.dtype == torch.float32, (tuple(h_in.shape), hshape)
    out = torch.empty(P, steps, batch, 1 if value_net else spec.n_actions, device=params.device)
    h_out = torch.empty(*hshape, device=params.device) if want_h else None
    s = spec.c()
    check(lib.marlhip_gru_ac_forward(ctypes.byref(s), int(value_net), _ptr(params), _ptr(obs), int(agent_stride), int(row_stride), int(steps),
                                     int(batch), _ptr(h_in), _ptr(h_out), _ptr(out), *_gru_fwd_ws(spec, steps, batch, params.device), _stream()), "gru_ac_forward")
    return (out, h_out) if want_h else out


def sample_from_logits(logits, seed, episode, t):
    """logits f32 [P][N][A] -> sampled actions i64 [P][N] (Philox inverse-CDF draw of the fused rollout collector)"""
    _require_gpu()
    P, N, A = logits.shape
    actions = torch.empty(P, N, dtype=torch.int64, device=logits.device)
    check(lib.marlhip_sample_from_logits(P, N, A, _ptr(logits.contiguous()), int(seed) & (2**64 - 1), _ptr(episode), int(t), _ptr(actions),
                                         _stream()), "sample_from_logits")
    return actions
