#!/usr/bin/env python
"""bench.py - env-steps/s of the IDQN hot path on Foraging-8x8-2p-3f (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]

`--gpus N` with N > 1 and no torchrun environment re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU,
backend nccl == RCCL); launched under torch.distributed.run directly it checks WORLD_SIZE == N.

One "step" = one ROUND of the hot path on this rank's N_env batched envs:
    fused collector (reset -> T x (act, env.step, replay add))             1 launch
    U x ( replay sample-gather of B episodes -> loss/grad -> [RCCL all-reduce] -> clip+Adam+target )
Cadence (the reference's 1 update of 32 episodes per collected episode cannot be
kept sequentially at 10M steps/s): the default keeps the reference's REPLAY RATIO - 32 sampled
episodes per collected episode - but batches the gradient steps: U = 32 updates of B = N_env
episodes per round.  `--cadence reference` runs the reference's own cadence (N_env sequential
updates of 32 episodes per round); `--cadence env-only` runs the collector alone.  The timed region
contains every kernel of the path incl. optimizer and target updates; inputs are resident in HBM.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: fused loss/grad, f32 MFMA-bound) and
`cpu_baseline` (the oracle port of the reference path on the host cores; N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENV_NAME = "lbforaging:Foraging-8x8-2p-3f-v3"
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs", type=int, default=4096, help="batched envs per GPU")
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--time-limit", type=int, default=25)
    ap.add_argument("--cadence", default="ratio", choices=["ratio", "reference", "env-only"])
    ap.add_argument("--update-batch", type=int, default=0, help="episodes per update (default: envs for ratio, 32 for reference)")
    ap.add_argument("--updates-per-round", type=int, default=0)
    ap.add_argument("--replay-rounds", type=int, default=4, help="replay capacity in rounds of `envs` episodes")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--env-name", default=ENV_NAME, help="other BASELINE.json configs, e.g. lbforaging:Foraging-15x15-4p-5f-v3 or rware:rware-tiny-4ag-v2 (with --time-limit 500)")
    ap.add_argument("--algo", default="idqn", choices=["idqn", "vdn", "qmix", "ia2c", "ippo", "maa2c", "mappo"])
    ap.add_argument("--rnn", action="store_true", help="recurrent Q-networks (algorithm.model.use_rnn=True; idqn / vdn, hidden 64)")
    ap.add_argument("--mixer-fp16", action="store_true", help="qmix: the opt-in fp16 first mixer layers (BASELINE config 5; a deviation from the fp32 reference)")
    ap.add_argument("--hparams", default="reference", choices=["tuned", "reference"],
                    help="idqn at --cadence ratio: `reference` (default, the headline line) = idqn.yaml's lr 3e-4 + hard copy every 200 updates; "
                         "`tuned` = the optimiser settings under which the batched cadence learns fastest (lr 3e-3 + Polyak 0.1; U >= 128: lr 1e-3 + "
                         "hard copy every 50; profiles/r02_learning_parity.md) - a `modes` row of the default line.  Same loss/grad kernels either "
                         "way (Polyak adds the target blend to the epilogue); named in config.lr / target_update_interval_or_tau")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = --envs envs and an update batch of B episodes PER GPU (effective batch G*B); strong = --envs envs and B "
                         "episodes in TOTAL, split evenly over the GPUs (the 1-GPU job's global batch and env count)")
    ap.add_argument("--split16", action="store_true",
                    help="idqn, hidden 64: the OPT-IN split-fp16 learner (products from fp16 halves on the double-rate MFMA, fp32 accumulate; "
                         "marlhip_idqn_update_n_split16) - a deviation from the exact-f32 default, reported as its own row")
    ap.add_argument("--no-modes", action="store_true", help="skip the secondary rows (`modes`: env-only, reference cadence, hidden 128) the default line carries")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-all-cores", type=int, default=0, help=argparse.SUPPRESS)  # internal: the forked all-cores leg of cpu_baseline
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--pretrain-rounds", type=int, default=0,
                    help="idqn / vdn / qmix: this many UNTIMED training rounds first (epsilon annealed 1 -> --eps-fixed over their first 60 %%), then the "
                         "warm-up and timed rounds at --eps-fixed: the trained-policy rows of `modes` (short episodes)")
    ap.add_argument("--eps-fixed", type=float, default=None)
    ap.add_argument("--clear-stale", action="store_true",
                    help="idqn / vdn / qmix: the collector zeroes the `filled` tail of a replay slot it reuses (marlhip_idqn_collect clear_stale) - a "
                         "DEVIATION from the reference's ReplayBuffer, which never resets `filled` (dqn/train.py:65-89: once the ring has wrapped, the rows "
                         "behind a shorter episode's end are the previous occupant's, still filled = True, and are trained on)")
    return ap.parse_args()


def _cpu_all_cores(seconds, hidden, workers, make_env, loop):
    """`workers` forked copies of `loop` (cpu_baseline hands over oracle/ref_learner.reference_idqn_loop), one thread each; (total env-steps, slowest window in s, copies
    that reported) or None.  The children never touch the GPU and leave through os._exit (no torch / HIP exit handlers in a fork)."""
    import select

    pipes, pids = [], []
    for _ in range(workers):
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            code = 1
            try:
                os.close(r)
                _, n_steps, _, dt, _ = loop(seconds, hidden, ENV_NAME, 25, make_env)
                os.write(w, f"{n_steps} {dt}\n".encode())
                code = 0
            finally:
                os._exit(code)
        os.close(w)
        pipes.append(r)
        pids.append(pid)
    deadline = time.perf_counter() + seconds + 60.0
    got, open_fds = [], set(pipes)
    while open_fds and time.perf_counter() < deadline:
        ready, _, _ = select.select(list(open_fds), [], [], 1.0)
        for fd in ready:
            data = os.read(fd, 256)
            if data:
                a, b = data.decode().split()
                got.append((int(a), float(b)))
            open_fds.discard(fd)
    for fd in pipes:
        os.close(fd)
    for pid in pids:  # exactly the processes started above
        try:
            done, _ = os.waitpid(pid, os.WNOHANG)
            if done == 0:
                os.kill(pid, 9)
                os.waitpid(pid, 0)
        except OSError:
            pass
    if not got:
        return None
    return sum(a for a, _ in got), max(b for _, b in got), len(got)


def cpu_baseline(seconds, hidden, all_cores_only=0):
    """The reference's CPU path on the host cores, reference cadence (1 update of 32 episodes per episode once 32 episodes are
    stored), 1 thread as marlbase/run.py:29, bounded sample, FPS as loggers.py:70.  `kind: "reference"`: the reference's OWN
    QNetwork / ReplayBuffer / _epsilon_schedule / _collect_trajectory, unmodified, from oracle/_ref (oracle/make_ref.py copies them
    there in the build container; git-ignored, it travels to the GPU box like a built .so) around oracle/lbf.py (lbforaging itself is
    not installable).  Without oracle/_ref: `kind: "port"`, the torch-CPU restatement oracle/dqn_port.py."""
    import numpy as np
    import torch

    from oracle import dqn_port as dp
    from oracle import ref_learner
    from oracle.lbf import MarlbaseEnv

    if ref_learner.available():
        make_env = lambda: MarlbaseEnv(ENV_NAME, 25, rng=np.random.default_rng(0))  # noqa: E731
        if all_cores_only:  # the child process of the all-cores leg below: fork the copies from a process that never touched the GPU
            return _cpu_all_cores(seconds, hidden, all_cores_only, make_env, ref_learner.reference_idqn_loop)
        v, n_steps, n_upd, dt, root = ref_learner.reference_idqn_loop(seconds, hidden, ENV_NAME, 25, make_env)
        where = os.path.relpath(root, ROOT) if root.startswith(ROOT) else root
        one = {"value": v, "unit": "env-steps/s", "cores": 1, "kind": "reference", "host_cores": os.cpu_count(),
               "sample": f"{n_steps} env-steps / {n_upd} updates of the reference's own marlbase.dqn QNetwork + ReplayBuffer + _collect_trajectory "
                         f"({where}) on oracle/lbf.py (python LBF env), IDQN {hidden}-{hidden}, reference cadence, 1 thread, in {dt:.1f} s; "
                         f"host has {os.cpu_count()} cores"}
        # the whole host, MEASURED: one independent 1-thread copy of the same loop per core (how marlbase is run on a CPU node: one
        # process per seed, marlbase/run.py:29), all started together, aggregate = total env-steps / the slowest copy's window
        try:
            workers = min(len(os.sched_getaffinity(0)), 256)
        except AttributeError:
            workers = min(os.cpu_count() or 1, 256)
        quota = None  # the container's CPU allowance when the cgroup states one ("max": none) - more copies than that only time-share
        for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            try:
                quota = open(path).read().strip()
                break
            except OSError:
                pass
        try:
            q = quota.split()
            period = float(q[1]) if len(q) > 1 else float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q[0] not in ("max", "-1"):
                workers = max(1, min(workers, int(-(-float(q[0]) // period))))
        except (AttributeError, IndexError, ValueError, OSError):
            pass
        agg = None
        if workers > 1:  # in a fresh interpreter without a HIP context (forking this one, with the runtime initialised, is not safe)
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-all-cores", str(workers), "--cpu-seconds", str(0.75 * seconds),
                                    "--hidden", str(hidden)], capture_output=True, text=True, timeout=seconds + 180,
                                   env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
                lines = [l for l in r.stdout.splitlines() if l.startswith("[")]
                agg = tuple(json.loads(lines[-1])) if lines else None
            except (subprocess.SubprocessError, OSError, ValueError):
                agg = None
        if agg is None:
            one["whole_node_estimate"] = v * (os.cpu_count() or 1)
            one["whole_node_estimate_note"] = "NOT measured: the 1-thread figure times the host's core count (independent runs, marlbase/run.py:29)"
            return one
        total_steps, slowest, n_ok = agg
        return {"value": total_steps / slowest, "unit": "env-steps/s", "cores": n_ok, "kind": "reference", "host_cores": os.cpu_count(),
                "cgroup_cpu_quota": quota,
                "one_thread": {"value": v, "sample": one["sample"]},
                "sample": f"{n_ok} concurrent 1-thread copies (one per CPU this container may use: host cores {os.cpu_count()}, cgroup quota {quota}) of the reference's own marlbase.dqn QNetwork + ReplayBuffer + "
                          f"_collect_trajectory ({where}) on oracle/lbf.py, IDQN {hidden}-{hidden}, reference cadence: {total_steps} env-steps in "
                          f"{slowest:.1f} s (the slowest copy's window); one copy alone: {v:.0f} env-steps/s"}

    torch.set_num_threads(1)
    P, D, A, T = 2, 15, 6, 25
    env = MarlbaseEnv(ENV_NAME, T, rng=np.random.default_rng(0))
    learner = dp.Learner(dp.init_params(P, D, hidden, A, seed=0), D, hidden, A)
    rb = dp.ReplayBuffer(10000, P, D, T)
    rng = np.random.default_rng(1)
    eps_sched = dp.epsilon_schedule("linear", 0.5, 1.0, 0.05, 6.5, 100000)
    steps = updates = 0
    # fill 32 episodes first (untimed) so the timed window is the steady state after training_start
    t0 = None
    while True:
        if t0 is None and rb.can_sample(32):
            t0, s0 = time.perf_counter(), steps
        if t0 is not None and time.perf_counter() - t0 >= seconds:
            break
        obs, _ = env.reset()
        rb.init_episode(obs)
        done = False
        eps = eps_sched(steps)
        while not done:
            o = torch.tensor(np.stack(obs)).unsqueeze(1)
            acts, _ = dp.act(learner.flat().detach(), o, eps, torch.tensor([rng.random()], dtype=torch.float32),
                             torch.tensor(rng.integers(0, A, (P, 1))), D, hidden, A)
            acts = [int(a) for a in acts[:, 0]]
            obs, rew, d, tr, info = env.step(acts)
            done = d or tr
            rb.add(obs, acts, rew, done)
            steps += 1
        if rb.can_sample(32):
            learner.update(rb.sample(32))
            updates += 1
    dt = time.perf_counter() - t0
    v = (steps - s0) / dt
    return {"value": v, "unit": "env-steps/s", "cores": 1, "kind": "port", "host_cores": os.cpu_count(),
            "whole_node_estimate": v * (os.cpu_count() or 1),  # independent 1-thread runs on every host core (marlbase/run.py:29)
            "whole_node_estimate_note": "NOT measured: the 1-thread figure times the host's core count",
            "sample": f"{steps - s0} env-steps / {updates} updates of oracle/lbf.py + oracle/dqn_port.py "
                      f"(python LBF env + torch-CPU IDQN {hidden}-{hidden}, reference cadence: 1 update of 32 episodes "
                      f"per episode, 1 thread) in {dt:.1f} s; host has {os.cpu_count()} cores"}


# ---- algorithmic FLOPs of one learner update (what `roofline.achieved` divides by the measured stage time) -------------------------
def mlp_fwd_flops(D, H, A):
    """one row through Linear(D,H)-ReLU-Linear(H,H)-ReLU-Linear(H,A) (utils/models.py:34-48)"""
    return 2.0 * (D * H + H * H + H * A)


def gru_fwd_flops(D, H, A):
    """one row-step through Linear(D,H)-ReLU-GRU(H,H)-Linear(H,A) (utils/models.py:51-116): W_ih and W_hh are [3H][H] each"""
    return 2.0 * (D * H + 6 * H * H + H * A)


def qmix_mixer_fwd_flops(P, SD):
    """QMixer.forward per (t, b) row with mixing = {embed 64, hypernet 2 x 32} (dqn/model.py:313-331): the four state-fed first
    layers as one [192 x SD] product, hyper_w_1.2 [32 -> 64 P], hyper_w_final.2 [32 -> 64], V.2 [64 -> 1]"""
    return 2.0 * (192 * SD + 32 * 64 * P + 32 * 64 + 64)


def first_layer_dx_flops(D, H):
    """the one product `backward = 2 x forward` counts that no implementation runs: the gradient w.r.t. a network's INPUT rows
    (2 D H per row).  Small for the 15-wide LBF rows (18 % of a forward), most of a 312-wide centralised critic's (71 %): the
    roofline objects carry `frac` by the usual convention (comparable across rounds) and `frac_needed` without this product."""
    return 2.0 * D * H


def dqn_update_flops(algo, rnn, P, D, A, H, T, B):
    """critic forward on T+1 observations, target forward on T (T+1 for recurrent nets: the sequence starts at t = 0), backward =
    2 x forward on the T transitions (T+1 steps of BPTT for recurrent nets); QMIX adds online + target mixer forward and the
    online mixer's backward (2 x) on the T * B rows."""
    if rnn:
        agents = gru_fwd_flops(D, H, A) * P * B * 4 * (T + 1)
    else:
        agents = mlp_fwd_flops(D, H, A) * P * B * ((T + 1) + T + 2 * T)
    mixer = 4.0 * qmix_mixer_fwd_flops(P, P * D) * T * B if algo == "qmix" else 0.0
    unneeded = first_layer_dx_flops(D, H) * P * B * (T + 1 if rnn else T)
    return agents + mixer, {"agent_networks": agents, "mixer": mixer, "first_layer_input_gradient (counted, never run)": unneeded}


def ac_update_flops(rnn, P, D, A, H, T, N, central, epochs=1, actor_forward_kept=False, critic_backward_deferred=False):
    """target-critic forward on T+1 rows, critic and actor forward + backward (3 x forward) on T rows; recurrent nets walk
    T+1 / T steps the same way.  PPO (`epochs` > 1): the prepare pass (target critic + old log-probs = one actor forward) once,
    then critic + actor forward / backward per epoch - the timer brackets one launch group, so this returns the per-call mean.
    actor_forward_kept (A2C on the fused collectors): the rollout left the actors' logits and hidden layers for the step
    (marlhip_*_ac_collect_keep), which then runs the actors' backward only - the stage under the timer does 2 x, not 3 x, their forward;
    PPO: the prepare pass and the first epoch read it (the later epochs run on moved parameters).
    critic_backward_deferred (A2C, one process, no joint clip): the critics' backward pass runs on a stream of its own next to the following
    rollout (update_async(overlap=True)) - it is not inside the timed launch group, so its 2 x forward is not counted for it either."""
    f = gru_fwd_flops if rnn else mlp_fwd_flops
    fa, fc = f(D, H, A), f(P * D if central else D, H, 1)
    k = 1 if actor_forward_kept else 0
    if epochs > 1:  # launch groups under the timer per rollout: 1 prepare + `epochs` epoch steps (kept: neither the prepare pass nor the first epoch runs the actors' forward)
        return P * N * ((fc * (T + 1) + (1 - k) * fa * T) + (epochs * (3 * fc + 3 * fa) - k * fa) * T) / (1 + epochs)
    return P * N * (fc * (T + 1) + (1 if critic_backward_deferred else 3) * fc * T + (3 - k) * fa * T)


def ac_unneeded_flops(P, D, H, T, N, central, epochs=1, critic_backward_deferred=False):
    """the first-layer input-gradient products inside ac_update_flops (see first_layer_dx_flops), per launch group like it"""
    dx = (first_layer_dx_flops(D, H) + (0.0 if critic_backward_deferred else first_layer_dx_flops(P * D if central else D, H))) * P * N * T
    return dx * epochs / (1 + epochs) if epochs > 1 else dx


TRAFFIC_SOURCES = {"": ("dqn_update_kernels.h", "mlp.h"), "H128": ("dqn_update_tp.h", "mlp.h"), "split16": ("dqn_update_h16.h", "mlp.h")}


def kernel_source_hash(files):
    """sha256 (16 hex digits) over the named kernel sources under codebase_amd/csrc, in the given order: what a committed PMC profile is
    keyed by, so that a traffic figure is only quoted for the kernel text it was measured on"""
    import hashlib

    h = hashlib.sha256()
    for f in files:
        h.update(f.encode() + b"\0")
        h.update(open(os.path.join(ROOT, "codebase_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def traffic_from_profile(key, variant=""):
    """HBM bytes per launch of the named workload's dominant kernel from the committed rocprofv3 PMC passes (bench.py cannot run
    rocprofv3 on itself).  Each workload entry of profiles/rNN_pmc_traffic.json carries `source_hash` = kernel_source_hash of the
    kernel's sources at the profiled head; when the sources in this tree hash differently the figure is NOT quoted: traffic null and
    the reason in `traffic_source`."""
    reason = "no committed PMC profile holds this workload"
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"):  # the newest profile that holds this exact workload
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            ent = pmc["workloads"].get(key)
            if ent is None:
                continue
            files = tuple(ent.get("source_files") or TRAFFIC_SOURCES[variant])
            now = kernel_source_hash(files)
            if ent.get("source_hash") != now:
                reason = (f"profiles/{name} was measured on kernel sources {ent.get('source_hash')} ({', '.join(files)} at head {pmc.get('head')}); "
                          f"this tree's hash is {now}: not quoted")
                continue
            return ent["traffic_bytes"], {"file": "profiles/" + name, "head": pmc.get("head"), "kernel": ent.get("kernel"),
                                          "source_files": list(files), "source_hash": now,
                                          "algorithmic_bytes": ent.get("algorithmic_bytes"),
                                          "note": "rocprofv3 PMC passes of this workload; the kernel sources of this tree hash to the profiled ones (bench.py cannot profile itself)"}
        except (OSError, KeyError, ValueError):
            continue
    return None, {"reason": reason}


def _ranks_field(world, dist, args, sync=None):
    """Ranks that took part in the gradient exchange, as torch.distributed reports them (backend: nccl == RCCL), and WHICH exchange
    carried the gradients: "p2p" = the library's own (csrc/p2p.hip: IPC-shared buffers, rank-ordered sum in the reduce launch),
    "collective" = torch.distributed's all-reduce.  (A p2p exchange that ran into its peer timeout fails the run before this point:
    GradSync.check() on every rank, right behind the timed region.)"""
    out = {"world_size": dist.get_world_size() if dist is not None else 1, "backend": args.backend if dist is not None else None}
    if dist is not None:
        out["exchange"] = "p2p (in-library)" if getattr(sync, "p2p", None) is not None else "collective (torch.distributed.all_reduce)"
        if hasattr(sync, "describe"):
            out.update(sync.describe())
    return out


def bench_ac(args, rank, world, dist, steps=None, warmup=None):
    """IA2C / IPPO (marlbase/ac): one step = one rollout of every env (fused collector, 1 launch) + one update
    (A2C) or num_epochs updates (PPO) on that rollout.  env-steps = transitions actually stored (sum of `filled`);
    the reference's own counter advances by t_max * parallel_envs per rollout (ac/train.py:226) - reported next to it."""
    import torch

    from codebase_amd import hip as h
    from codebase_amd._lib import lib
    from codebase_amd.ac.model import A2CNetwork, PPONetwork
    from codebase_amd.ac.train import Batch
    from codebase_amd.parallel import GradSync, rank_env_seed
    from codebase_amd.utils.envs import _space_pair

    N, T, H = args.envs, args.time_limit, args.hidden
    n_steps_timed = args.steps if steps is None else steps
    n_warmup = args.warmup if warmup is None else warmup
    cfg = h.env_config(args.env_name, N, T, seed=rank_env_seed(args.seed, rank))
    P, (D, A) = cfg.n_agents, h.env_dims(cfg)
    torch.manual_seed(args.seed)
    obs_space, act_space = _space_pair(cfg)
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=False, n_steps=5, entropy_coef=0.001, value_loss_coef=0.5,
                 standardise_returns=False, target_update_interval_or_tau=200, num_epochs=4, ppo_clip=0.2)  # ia2c.yaml / ippo.yaml
    net = dict(layers=[H, H], parameter_sharing=False, use_orthogonal_init=True, use_rnn=bool(args.rnn))
    central = args.algo in ("maa2c", "mappo")  # critic.centralised (maa2c.yaml / mappo.yaml)
    model = (PPONetwork if args.algo in ("ippo", "mappo") else A2CNetwork)(obs_space, act_space, hyper, net,
                                                                           dict(net, centralised=central), "cuda")
    dev = model.device
    # two sets of batch tensors, used in turn: A2C's critics finish their half of update r (backward pass, step, target update) on a stream
    # of their own while rollout r + 1 is being written (update_async(overlap=True)) - the drivers allocate a fresh batch per rollout
    # (ac/train.py:36-49), the bench keeps its two
    bufs = [dict(obs=torch.empty(T + 1, N, P * D, device=dev), act=torch.empty(T, N, P, dtype=torch.int64, device=dev),
                 rew=torch.empty(T, N, P, device=dev), done=torch.empty(T + 1, N, dtype=torch.uint8, device=dev),
                 donef=torch.empty(T + 1, N, device=dev), fill=torch.empty(T, N, device=dev)) for _ in range(2)]
    fin_ret = torch.zeros(P, N, device=dev)
    fin_len = torch.zeros(N, dtype=torch.int32, device=dev)
    t_max = torch.zeros(1, dtype=torch.int32, device=dev)
    steps_dev = torch.zeros((), dtype=torch.int64, device=dev)
    ref_steps = torch.zeros((), dtype=torch.int64, device=dev)
    # the feed-forward rounds count with ONE small elementwise launch each (per-env first-episode lengths and the round's longest episode
    # added into accumulators, reduced once behind the timed region): a per-round reduction to a scalar is bench bookkeeping that cost
    # 17 + 10 us of a 460 us IA2C round
    len_acc = torch.zeros(N, dtype=torch.int64, device=dev)
    tmax_acc = torch.zeros(1, dtype=torch.int64, device=dev)
    # N > 1: the joint gradient on one lane, and a second lane for the critics' slice when their half of the update runs beside the next
    # rollout (A2CNetwork.update_async(overlap=True); PPO defers nothing)
    sync_grad = GradSync(dist, max_floats=model.updater.grad.numel(),
                         side_floats=model.updater.critic_grad.numel() if hasattr(model, "attach_grad_sync") and args.algo in ("ia2c", "maa2c") else 0) if dist is not None else None
    state = {"round": 0, "step": 0}

    if args.rnn:  # recurrent actors: the rollout runs through the modular entry points (hidden state carried between steps)
        from codebase_amd.ac.train import _collect_trajectories_recurrent
        from codebase_amd.utils.envs import HipForagingVecEnv

        vec = HipForagingVecEnv(cfg)

    # a stream of the bench's own: the critics' half of an A2C update can only run next to the following rollout when the caller is not on
    # the default stream (AcUpdater.can_defer)
    own_stream = torch.cuda.Stream(device=dev)

    def one_round():
        with torch.cuda.stream(own_stream):
            _one_round()

    def _one_round():
        if args.rnn:
            tmax, batch, _, _, _ = _collect_trajectories_recurrent(vec, model, T, False, state["round"])
            model.update_async(batch._replace(dones=batch.dones.float()), state["step"], grad_sync=sync_grad, world=world)
            steps_dev.add_(batch.filled.sum().to(torch.int64))
            ref_steps.add_(tmax * N)
            state["round"] += 1
            state["step"] += T * N
            return
        b = bufs[state["round"] & 1]
        state["kept"] = h.ac_collect(cfg, model.spec, model.actor_params, state["round"], T, False, b["obs"], b["act"], b["rew"], b["done"], b["fill"],
                                     fin_ret, fin_len, t_max, keep_for=model.updater if getattr(model, "keeps_actor_forward", False) else None)
        b["donef"].copy_(b["done"])  # batch.dones.float() (ac/model.py:198)
        model.update_async(Batch(b["obs"], b["act"], b["rew"], b["donef"], b["fill"], None), state["step"], grad_sync=sync_grad, world=world, overlap=True)
        state["deferred"] = model.updater._critic_event is not None
        len_acc.add_(fin_len)  # sum == b_fill.sum(): every env stores exactly its first episode (the collector's contract; checked once below)
        tmax_acc.add_(t_max)
        state["round"] += 1
        state["step"] += T * N  # host-side stand-in for the reference's step counter (target update cadence only)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if sync_grad is not None and hasattr(model, "attach_grad_sync"):
        with torch.cuda.stream(own_stream):
            state["split_exchange"] = model.attach_grad_sync(sync_grad)  # a vote over the ranks: all defer or none
    for _ in range(n_warmup):
        one_round()
    sync()
    if not args.rnn and n_warmup > 0:  # the counter above counts what the batch holds
        assert int(fin_len.sum().item()) == int(bufs[(state["round"] - 1) & 1]["fill"].sum().item()), "stored transitions != sum of first-episode lengths"
    steps_dev.zero_()
    ref_steps.zero_()
    len_acc.zero_()
    tmax_acc.zero_()
    if not args.no_kernel_timing:
        lib.marlhip_timing_enable(1)
    t0 = time.perf_counter()
    for _ in range(n_steps_timed):
        one_round()
    sync()
    dt = time.perf_counter() - t0
    if not args.rnn:
        steps_dev.add_(len_acc.sum())
        ref_steps.add_(tmax_acc[0] * N)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
        dist.all_reduce(steps_dev)
        dist.all_reduce(ref_steps)
    env_steps = int(steps_dev.item())
    timing = {}
    if not args.no_kernel_timing:
        for kid, kname in ((0, "ac_update (fwd rows x3, elementwise, bwd rows x2)"), (1, "ac_collect_kernel")):
            n, ms = ctypes.c_int64(0), ctypes.c_double(0.0)
            lib.marlhip_timing_read(kid, ctypes.byref(n), ctypes.byref(ms))
            if n.value:
                timing[kname] = {"launches": n.value, "avg_us": 1e3 * ms.value / n.value, "total_ms": ms.value}
        lib.marlhip_timing_enable(0)
    exchange = None
    if sync_grad is not None:
        sync_grad.check()
        exchange = sync_grad.describe()
        exchange["critics_slice_on_their_own_lane"] = bool(state.get("split_exchange", False))
    digest = None
    if getattr(args, "want_digest", False):
        import hashlib

        model.updater.sync_critic()
        torch.cuda.synchronize()
        digest = hashlib.sha256(model.updater.block.cpu().numpy().tobytes()).hexdigest()[:16]
    if sync_grad is not None:
        sync_grad.close()
    if rank != 0:
        return None
    name = args.env_name.split(":")[-1].replace("-v3", "").replace("-v2", "")
    roofline = None
    upd = timing.get("ac_update (fwd rows x3, elementwise, bwd rows x2)")
    col = timing.get("ac_collect_kernel")
    if upd:
        kept = bool(state.get("kept", False))  # the collector left the actors' forward pass for the step (hip.ac_collect(keep_for=...))
        deferred = bool(state.get("deferred", False))  # the critics' backward pass ran next to the following rollout, outside the timed launch group
        flops = ac_update_flops(bool(args.rnn), P, D, A, H, T, N, central, epochs=hyper["num_epochs"] if args.algo in ("ippo", "mappo") else 1,
                                actor_forward_kept=kept, critic_backward_deferred=deferred)
        ach = flops / (upd["avg_us"] * 1e-6) / 1e12
        needed = flops - ac_unneeded_flops(P, D, H, T, N, central, epochs=hyper["num_epochs"] if args.algo in ("ippo", "mappo") else 1,
                                           critic_backward_deferred=deferred)
        traffic, tsrc = traffic_from_profile(f"{args.algo}:{args.env_name}:N{N}:H{H}:T{T}")  # (whole update stage incl. the critics' backward pass, per rollout)
        roofline = {"kernel": "ac_update stage (forward rows, elementwise, backward rows)" + ("; the actors' forward pass is the collector's own, kept for the step" if kept else ""),
                    "actor_forward_kept": kept, "critic_backward_overlaps_next_rollout": deferred, "bound": "mfma", "achieved": ach,
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": tsrc,
                    "flops_per_launch": flops, "avg_launch_us": upd["avg_us"],
                    "flops_needed_per_launch": needed, "frac_needed": ach * needed / flops / PEAK_F32_MFMA_TFLOPS,
                    "dominant_stage_by_time": "ac_update" if not col or upd["total_ms"] >= col["total_ms"] else "ac_collect_kernel"}
        if deferred:
            roofline["side_stream"] = h.side_stream_description(dev)
        # the whole round against its wall time - every product the round runs, wherever it runs: the rollout's acting forward pass, the
        # update stage above AND the critics' deferred backward pass (which the stage's own `frac` leaves out of both FLOPs and time)
        epochs = hyper["num_epochs"] if args.algo in ("ippo", "mappo") else 1
        f_all = (ac_update_flops(bool(args.rnn), P, D, A, H, T, N, central, epochs=epochs, actor_forward_kept=kept) * (1 + epochs if epochs > 1 else 1)
                 + (gru_fwd_flops if args.rnn else mlp_fwd_flops)(D, H, A) * P * (env_steps / max(n_steps_timed * world, 1)))
        round_s = dt / n_steps_timed
        roofline["whole_round"] = {"flops_per_round_per_gpu": f_all, "round_ms": 1e3 * round_s,
                                   "frac": f_all / round_s / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                   "note": "rollout forward + update stage + the critics' backward pass (deferred or not) over the round's wall time"}
        if col and col["total_ms"] > upd["total_ms"]:  # the rollout dominates (long episodes, few envs): its acting forward next to it
            cf = (gru_fwd_flops if args.rnn else mlp_fwd_flops)(D, H, A) * P * (env_steps / n_steps_timed)
            roofline["collector"] = {"kernel": "ac_collect_kernel", "bound": "mfma (latency-bound in practice: one wave per 16 envs walks the episode)",
                                     "flops_per_launch": cf, "avg_launch_us": col["avg_us"],
                                     "frac": cf / (col["avg_us"] * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS}
    out = {
        "metric": f"env-steps/sec (whole node) {args.algo.upper()} {name}", "value": env_steps / dt, "unit": "env-steps/s",
        "n_gpus": world, "rccl_ranks": _ranks_field(world, dist, args, sync_grad), "steps": n_steps_timed, "warmup": n_warmup, "ms_per_step": 1e3 * dt / n_steps_timed,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (Philox-seeded env layouts, orthogonal-init weights)",
        "config": {"workload": f"{args.algo.upper()} on {name}, {N} batched HIP envs per GPU, actor/critic " + (f"GRU-{H} networks (use_rnn), " if args.rnn else f"2-layer-{H} MLPs, ")
                               + f"time_limit {T}, one update per rollout", "envs_per_gpu": N, "env_steps_timed": env_steps,
                   # the batch stores every env's FIRST episode of a rollout: stored transitions / (rollouts x envs x ranks)
                   "mean_episode_length": env_steps / max(n_steps_timed * N * world, 1),
                   "policy_regime": "untrained actors (orthogonal init, a few Adam steps of warm-up): episodes run to the time limit unless the env ends them",
                   "reference_step_counter": int(ref_steps.item()),
                   "parallelism": f"dp{world} (envs sharded per GPU, RCCL grad all-reduce per update)" if world > 1 else "1 GPU"},
        "kernels": timing, "roofline": roofline,
    }
    if exchange is not None:
        out["rccl_ranks"].update(exchange)
    if digest is not None:
        out["params_sha16_rank0"] = digest
    return out


def _self_launch(n):
    """`python bench.py --gpus N` outside torchrun: become `python -m torch.distributed.run ... bench.py <same args>`
    (exec, so the driver's clock and exit code see the N-rank job; rank 0 prints the one JSON line)."""
    import socket

    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    args = parse()
    if args.cpu_all_cores:  # internal: cpu_baseline's all-cores leg in its own interpreter (no GPU use in this process)
        print(json.dumps(cpu_baseline(args.cpu_seconds, args.hidden, all_cores_only=args.cpu_all_cores)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(args.gpus)
    import torch

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or plain `python bench.py --gpus {args.gpus}`, which launches the ranks itself)")
    if os.environ.get("MARLHIP_BENCH_DRYRUN"):
        # launch-plumbing check without a GPU (tests/test_bench_launch.py): rendezvous over gloo, one all-reduce, no hot path
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.ones(1)
            dist.all_reduce(t)
            assert dist.get_world_size() == args.gpus
            ranks = int(t[0])
            dist.destroy_process_group()
        else:
            ranks = 1
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": world, "ranks_in_allreduce": ranks}), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the marlhip hot path has no CPU fallback")
    if world > 1 and not os.environ.get("MARLHIP_BENCH_ONE_DEVICE") and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} needs {world} visible devices, found {torch.cuda.device_count()}")
    # MARLHIP_BENCH_BACKEND=gloo + MARLHIP_BENCH_ONE_DEVICE=1 let the N>1 code path be exercised on a 1-GPU box
    backend = os.environ.get("MARLHIP_BENCH_BACKEND", "nccl")  # nccl == RCCL on ROCm
    dev_index = 0 if os.environ.get("MARLHIP_BENCH_ONE_DEVICE") else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1 or os.environ.get("MARLHIP_BENCH_FORCE_DIST"):  # FORCE_DIST: the N > 1 code path (RCCL all-reduce per update) with one rank, to time its host side
        import torch.distributed as dist

        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": torch.device("cuda", dev_index)} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    args.backend = backend

    if args.algo in ("ia2c", "ippo", "maa2c", "mappo"):
        out = bench_ac(args, rank, world, dist)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    default_line = (args.algo == "idqn" and args.cadence == "ratio" and args.hidden == 64 and not args.rnn and args.env_name == ENV_NAME
                    and not args.update_batch and not args.updates_per_round and not args.split16 and not args.pretrain_rounds)
    multi = world > 1 and default_line and not args.no_modes  # the first multi-GPU run describes itself
    if multi:
        args.want_digest = True
    out = bench_dqn(args, rank, world, dist, args.steps, args.warmup)
    if multi:
        rows = multi_gpu_rows(args, rank, world, dist, out)
        if rank == 0:
            out["rccl_ranks"]["exchange_rows"] = rows["exchange_rows"]
            out["modes"] = rows["modes"]
    if rank == 0:
        if world == 1 and default_line and not args.no_modes:
            out["modes"] = secondary_modes(args)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.hidden)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def multi_gpu_rows(args, rank, world, dist, headline):
    """N > 1 only, every rank runs the same sequence: (i) the headline workload a second time in the same process group with the OTHER
    exchange (MARLHIP_P2P=0: torch.distributed's all-reduce, RCCL under backend nccl), so that one line carries both - value, the
    per-update exchange time from the library's timers (id 5), what carried the gradients and why; for two ranks the rank-ordered
    sum IS the collective's sum, so the two rows' final parameters must hash the same; (ii) BASELINE configs 4 and 5 at their
    per-GPU shard (2048 / 8192 envs per rank) with their own exchanges (config 4: the critics' slice on the second lane)."""
    import copy
    import gc

    import torch

    def summary(r):
        k = (r.get("kernels") or {})
        ex = next((v for n, v in k.items() if n.startswith("gradient_exchange")), None)
        rr = r["rccl_ranks"]
        return {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"], "warmup": r["warmup"],
                "exchange": rr.get("exchange"), "self_test": rr.get("self_test"), "fallback_reason": rr.get("fallback_reason"),
                "per_update_exchange_us": ex["avg_us"] if ex else None, "exchanges_timed": ex["launches"] if ex else None,
                "params_sha16_rank0": r.get("params_sha16_rank0")}

    rows = {"exchange_rows": {}, "modes": {}}
    if rank == 0:
        rows["exchange_rows"]["default (MARLHIP_P2P unset): " + headline["rccl_ranks"].get("exchange", "?")] = summary(headline)
    if os.environ.get("MARLHIP_BENCH_EXTRAS", "1") == "0":  # the headline alone (a driver that wants the shortest possible N > 1 run)
        return rows
    keep = os.environ.get("MARLHIP_P2P")
    os.environ["MARLHIP_P2P"] = "0"
    try:
        a = copy.copy(args)
        r = bench_dqn(a, rank, world, dist, args.steps, args.warmup)
        if rank == 0:
            rows["exchange_rows"]["MARLHIP_P2P=0: " + r["rccl_ranks"].get("exchange", "?")] = summary(r)
    except Exception as e:  # noqa: BLE001 - a secondary row never takes the headline down (an error every rank raises alike; a rank-local one cannot be caught here)
        if rank == 0:
            rows["exchange_rows"]["MARLHIP_P2P=0: error"] = {"error": f"{type(e).__name__}: {e}"}
    finally:
        if keep is None:
            os.environ.pop("MARLHIP_P2P", None)
        else:
            os.environ["MARLHIP_P2P"] = keep
    if rank == 0:
        shas = [v.get("params_sha16_rank0") for v in rows["exchange_rows"].values()]
        rows["exchange_rows"]["same_final_parameters"] = (len(set(shas)) == 1 and None not in shas) if world == 2 else None  # (N > 2: the collective's summation order is its own)
    for name, over, steps, warmup in (
            ("BASELINE config 4 (per-GPU shard): IA2C rware-tiny-4ag, 2048 envs per rank, 128-128", dict(algo="ia2c", env_name="rware:rware-tiny-4ag-v2", envs=2048, hidden=128, time_limit=500), 16, 2),
            ("BASELINE config 5 (per-GPU shard): QMIX Foraging-15x15-8p-5f, 8192 envs per rank, 128-128, fp32 mixer", dict(algo="qmix", env_name="lbforaging:Foraging-15x15-8p-5f-v3", envs=8192, hidden=128), 2, 1)):
        a = copy.copy(args)
        for k, v in over.items():
            setattr(a, k, v)
        if os.environ.get("MARLHIP_BENCH_SMALL_ROWS"):  # the one-device plumbing test: the same rows at a size two ranks sharing a GPU finish in seconds
            a.envs, a.time_limit = (256, 60) if a.algo == "ia2c" else (256, 25)
            steps, warmup = 3, 1
        try:
            r = bench_ac(a, rank, world, dist, steps, warmup) if a.algo == "ia2c" else bench_dqn(a, rank, world, dist, steps, warmup)
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                rows["modes"][name] = {"error": f"{type(e).__name__}: {e}"}
            gc.collect()
            torch.cuda.empty_cache()
            continue
        if rank == 0:
            rf = r.get("roofline") or {}
            row = summary(r)
            row.update({"workload": r["config"]["workload"], "critic_backward_overlaps_next_rollout": rf.get("critic_backward_overlaps_next_rollout"),
                        "critics_slice_on_their_own_lane": r["rccl_ranks"].get("critics_slice_on_their_own_lane"),
                        "roofline": {k: rf.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_us", "whole_round")}})
            rows["modes"][name] = row
        gc.collect()
        torch.cuda.empty_cache()
    return rows


def secondary_modes(args):
    """BASELINE.md 3.5's other modes and the reference-default network, timed by the same process right after the headline region so
    that the driver's bench line witnesses them too: env-only (collector alone), cadence=reference (the reference's one update of 32
    episodes per collected episode, sequentially), the headline cadence with the reference's default 128-128 networks
    (configs/algorithm/idqn.yaml:8-10), a trained-policy row (short episodes) and BASELINE.json's configs 3 - 5 at their per-GPU shard.
    Short regions (tens to hundreds of ms; the trained-policy row trains ~5 s first); `value`, `config` and `roofline` of the line are untouched."""
    import copy

    rows = {}

    def row(r, steps, warmup):
        rf, c = r.get("roofline") or {}, r["config"]
        return {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": steps, "warmup": warmup, "dtype": r["dtype"],
                "workload": c["workload"], "updates_per_round": c.get("updates_per_round"), "update_batch_episodes": c.get("update_batch_episodes"),
                "lr": c.get("lr"), "target_update_interval_or_tau": c.get("target_update_interval_or_tau"),
                "mean_episode_length": c.get("mean_episode_length"), "mean_episode_return_last_round": c.get("mean_episode_return_last_round"),
                "epsilon_timed_rounds": c.get("epsilon_timed_rounds"),
                "mean_filled_rows_per_sampled_episode": c.get("mean_filled_rows_per_sampled_episode"), "replay_clear_stale": c.get("replay_clear_stale"),
                "roofline": dict({k: rf.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_needed", "avg_launch_us", "traffic", "whole_round")},
                                 traffic_source=(rf.get("traffic_source") or {}).get("file") or (rf.get("traffic_source") or {}).get("reason"))}

    for name, over, steps, warmup in (("hparams=tuned (lr 3e-3, Polyak 0.1: what the batched cadence learns fastest with), cadence=ratio", dict(hparams="tuned"), 10, 3),
                                      ("env-only", dict(cadence="env-only"), 100, 5),
                                      ("cadence=reference", dict(cadence="reference"), 3, 1),
                                      ("hidden=128 (reference default net), cadence=ratio", dict(hidden=128), 5, 2),
                                      ("split16 OPT-IN learner (fp16 hi/lo products, fp32 accumulate; NOT the default), cadence=ratio", dict(split16=True), 10, 3),
                                      # how env-steps/s moves with the episode length: the same loop after ~5 s of training with the
                                      # tuned optimiser settings, timed at epsilon 0.05 - episodes end when the food is gone, update cost per sampled
                                      # episode stays what it was
                                      ("trained policy (hparams=tuned, 1500 rounds of training first, epsilon 0.05): short episodes, cadence=ratio",
                                       dict(hparams="tuned", pretrain_rounds=1500, eps_fixed=0.05,
                                            regime_note="after 1500 untimed training rounds (epsilon 1 -> 0.05), timed at epsilon 0.05"), 20, 2),
                                      # round 6: WHY the row above pays for 25 rows per sampled episode - the reference's
                                      # ReplayBuffer never resets `filled` (dqn/train.py:65-89), so once the ring has wrapped a slot's rows behind a
                                      # shorter episode's end are its previous occupant's, still filled = True, and the reference trains on them
                                      # (`mean_filled_rows_per_sampled_episode` = 25.0 there).  With clear_stale (an opt-in DEVIATION: the collector zeroes
                                      # the tail of a slot it reuses) the filled-aware plan (csrc/update_plan.h) walks only what an episode stored
                                      ("trained policy as above, replay with clear_stale (OPT-IN deviation from the reference's ReplayBuffer: no stale filled tails), filled-aware update plans",
                                       dict(hparams="tuned", pretrain_rounds=1500, eps_fixed=0.05, clear_stale=True,
                                            regime_note="after 1500 untimed training rounds (epsilon 1 -> 0.05), timed at epsilon 0.05; replay slots cleared on reuse"), 20, 2)):
        a = copy.copy(args)
        for k, v in over.items():
            setattr(a, k, v)
        rows[name] = row(bench_dqn(a, 0, 1, None, steps, warmup), steps, warmup)
    # BASELINE.json configs 3 - 5 at their per-GPU shard (configs 4 and 5 are quoted on 8 GPUs: 16384 / 8 and 65536 / 8 envs), so that the
    # driver's line witnesses them; config 3 names no width: both the headline's 64 and the reference default 128
    for name, over, steps, warmup in (
            ("BASELINE config 3: VDN Foraging-15x15-4p-5f, 8192 envs, 64-64", dict(algo="vdn", env_name="lbforaging:Foraging-15x15-4p-5f-v3", envs=8192, hidden=64), 3, 1),
            ("BASELINE config 3: VDN Foraging-15x15-4p-5f, 8192 envs, 128-128", dict(algo="vdn", env_name="lbforaging:Foraging-15x15-4p-5f-v3", envs=8192, hidden=128), 2, 1),
            ("BASELINE config 4 (per-GPU shard): IA2C rware-tiny-4ag, 2048 envs, 128-128", dict(algo="ia2c", env_name="rware:rware-tiny-4ag-v2", envs=2048, hidden=128, time_limit=500), 16, 2),  # (16 rounds: the last update's critics finish inside the timed region, beside no rollout - 7 ms over the K rounds)
            ("BASELINE config 5 (per-GPU shard): QMIX Foraging-15x15-8p-5f, 8192 envs, 128-128, fp32 mixer", dict(algo="qmix", env_name="lbforaging:Foraging-15x15-8p-5f-v3", envs=8192, hidden=128), 2, 1),
            ("BASELINE config 5 (per-GPU shard): QMIX Foraging-15x15-8p-5f, 8192 envs, 128-128, OPT-IN fp16 first mixer layers", dict(algo="qmix", env_name="lbforaging:Foraging-15x15-8p-5f-v3", envs=8192, hidden=128, mixer_fp16=True), 2, 1)):
        a = copy.copy(args)
        for k, v in over.items():
            setattr(a, k, v)
        try:
            r = bench_ac(a, 0, 1, None, steps, warmup) if a.algo == "ia2c" else bench_dqn(a, 0, 1, None, steps, warmup)
            rows[name] = row(r, steps, warmup)
        except Exception as e:  # noqa: BLE001 - a secondary row never takes the headline down
            rows[name] = {"error": f"{type(e).__name__}: {e}"}
        import gc

        import torch

        gc.collect()
        torch.cuda.empty_cache()  # the 8-agent workspaces are GBs: hand them back before the next row
    return rows


def bench_dqn(args, rank, world, dist, steps, warmup):
    """IDQN / VDN / QMIX: `warmup` untimed rounds, then `steps` timed rounds bracketed by barrier + synchronize, max over ranks;
    returns the JSON object of the line on rank 0 (None elsewhere)"""
    import torch

    from codebase_amd import hip as h
    from codebase_amd._lib import lib
    from codebase_amd.dqn.model import QNetwork
    from codebase_amd.dqn.train import VectorisedIDQN, _epsilon_schedule
    from codebase_amd.utils.envs import _space_pair

    N, T, H = args.envs, args.time_limit, args.hidden
    if args.scaling == "strong" and world > 1:  # the 1-GPU job's env count and global batch, split evenly over the ranks
        if N % world:
            raise SystemExit(f"bench.py --scaling strong: --envs {N} is not a multiple of --gpus {world}")
        N //= world
    from codebase_amd.parallel import rank_env_seed

    cfg = h.env_config(args.env_name, N, T, seed=rank_env_seed(args.seed, rank), cooperative=args.algo != "idqn")
    P, (D, A) = cfg.n_agents, h.env_dims(cfg)
    strong = args.scaling == "strong" and world > 1
    if args.cadence == "ratio":
        B = (args.update_batch // world if strong else args.update_batch) or N
        U = args.updates_per_round or max(1, (32 * N) // B)
    elif args.cadence == "reference":
        B = max((args.update_batch or 32) // (world if strong else 1), 1)
        U = args.updates_per_round or N * (world if strong else 1)  # one update per collected episode of the whole job
    else:
        B, U = 1, 0
    torch.manual_seed(args.seed)  # identical initial weights on every rank (orthogonal init, utils/models.py:8-11)
    obs_space, act_space = _space_pair(cfg)
    hyper = dict(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, double_q=True, standardise_returns=False,
                 target_update_interval_or_tau=200)  # marlbase/configs/algorithm/idqn.yaml:16-37
    if args.cadence == "ratio" and args.algo == "idqn" and args.hparams == "tuned":
        # batched gradient steps want a larger step and a faster target (profiles/r02_learning_parity.md: with the reference's lr /
        # target interval the batched cadence barely learns); U = 32: lr 3e-3 + Polyak 0.1; U >= 128: lr 1e-3 + hard copy every 50 updates
        hyper.update(dict(lr=3e-3, target_update_interval_or_tau=0.1) if U < 128 else dict(lr=1e-3, target_update_interval_or_tau=50))
    if getattr(args, "split16", False):
        hyper["split16"] = True
    from codebase_amd.dqn.model import QMixNetwork, VDNetwork

    if args.algo == "qmix":  # marlbase/configs/algorithm/qmix.yaml:14-17
        model = QMixNetwork(obs_space, act_space, hyper, [H, H], False, bool(args.rnn), True,
                            dict(embed_dim=64, hypernet_layers=2, hypernet_embed=32, fp16=bool(args.mixer_fp16)), "cuda")
    else:
        model = (VDNetwork if args.algo == "vdn" else QNetwork)(obs_space, act_space, hyper, [H, H], False, bool(args.rnn), True, "cuda")
    cap = args.replay_rounds * N
    trainer = VectorisedIDQN(cfg, model, cap, T, B, U, seed=args.seed, dist=dist, clear_stale=bool(getattr(args, "clear_stale", False)))
    eps_sched = _epsilon_schedule("linear", 0.5, 1.0, 0.05, 6.5, 100_000_000)
    # `modes` row "trained policy": `pretrain_rounds` untimed rounds of the same loop with epsilon annealed 1 -> eps_fixed over their first
    # 60 %, then the warm-up and the timed rounds at eps_fixed (secondary_modes sets both; the default line has neither)
    pre, eps_fixed = int(getattr(args, "pretrain_rounds", 0) or 0), getattr(args, "eps_fixed", None)
    if pre and eps_fixed is None:
        eps_fixed = 0.05

    def eps_at(rnd):
        if pre:
            return max(eps_fixed, 1.0 - (1.0 - eps_fixed) * rnd / (0.6 * pre))
        return eps_sched(rnd * N * T)

    def one_round():
        trainer.round(eps_at(trainer.rounds), train=U > 0)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(pre + warmup):
        one_round()
    sync()
    trainer.reset_env_steps()
    if not args.no_kernel_timing:
        lib.marlhip_timing_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        one_round()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
    steps_dev = trainer.env_steps
    if dist is not None:
        dist.all_reduce(steps_dev)
    env_steps = int(steps_dev.item())
    mean_return = float(trainer.fin_return.sum(0).mean().item())  # the last round's episodes, summed over the agents (this rank's)
    # rows the learner must compute per sampled episode = sum(filled) of the last update / B: the episode's own transitions PLUS, once the ring
    # has wrapped, the stale tail of the slot's longer previous occupants (the reference never resets `filled`: dqn/train.py:65-89)
    sampled_rows = float(trainer.last_loss[1].item()) / B if (U and trainer.last_loss is not None) else None

    timing = {}
    if not args.no_kernel_timing:
        for kid, kname in ((0, "dqn_lossgrad_kernel"), (1, "idqn_collect_kernel"), (2, "replay_sample_kernel"),
                           (4, "qmix_mixer_stage"), (5, "gradient_exchange (reduce launch with the in-library exchange inside, or the exchange call)")):
            n, ms = ctypes.c_int64(0), ctypes.c_double(0.0)
            lib.marlhip_timing_read(kid, ctypes.byref(n), ctypes.byref(ms))
            if n.value:
                timing[kname] = {"launches": n.value, "avg_us": 1e3 * ms.value / n.value, "total_ms": ms.value}
        lib.marlhip_timing_enable(0)

    exchange_sync = getattr(trainer, "_sync", None)
    ranks_field = _ranks_field(world, dist, args, exchange_sync)
    digest = None
    if getattr(args, "want_digest", False):
        import hashlib

        torch.cuda.synchronize()
        digest = hashlib.sha256(model.params.cpu().numpy().tobytes()).hexdigest()[:16]
    if exchange_sync is not None:
        exchange_sync.check()  # every rank: a timed-out in-library exchange fails the run on all of them
        exchange_sync.close()  # (freed behind a job-wide barrier: the next row of a multi-row run sets up its own)
    mean_len = env_steps / max(steps * N * world, 1)  # every round stores N episodes per rank
    eps_timed = (eps_at(trainer.rounds - steps), eps_at(trainer.rounds - 1))
    del trainer, model
    if rank != 0:
        return None

    roofline = None
    lg, col = timing.get("dqn_lossgrad_kernel"), timing.get("idqn_collect_kernel")
    if U and lg:
        # the timer brackets the loss/grad launch group of one update: agent forward(s) [+ mixer stage] + agent backward
        flops, parts = dqn_update_flops(args.algo, bool(args.rnn), P, D, A, H, T, B)
        avg_s = lg["avg_us"] * 1e-6
        ach = flops / avg_s / 1e12
        key = f"{args.algo}:{args.env_name}:N{N}:H{H}:B{B}:T{T}:rnn{int(bool(args.rnn))}" + (":split16" if getattr(args, "split16", False) else "")
        traffic, tsrc = traffic_from_profile(key, "split16" if getattr(args, "split16", False) else ("H128" if H > 64 else ""))
        kname = "dqn_lossgrad_h16_kernel (split-fp16 products, fp32 accumulate)" if getattr(args, "split16", False) else ("gru_seq_fwd2 + gru_td + gru_seq_bwd + gru_wgrad" if args.rnn else
                 ("dqn_lossgrad_kernel" if H <= 64 else "tp_fwd_kernel + tp_mix_kernel + tp_bwd_kernel")) + (" + qmix mixer stage" if args.algo == "qmix" else "")
        roofline = {"kernel": kname, "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": tsrc, "flops_per_launch": flops,
                    "flops_parts": parts, "avg_launch_us": lg["avg_us"],
                    "frac_needed": ach * (1.0 - parts["first_layer_input_gradient (counted, never run)"] / flops) / PEAK_F32_MFMA_TFLOPS,
                    "dominant_stage_by_time": "loss/grad" if not col or lg["total_ms"] >= col["total_ms"] else "idqn_collect_kernel"}
        if getattr(args, "split16", False):
            # the same ALGORITHMIC f32 FLOPs over the f32 MFMA peak (comparable with the default row); on the pipe it really runs on it
            # issues 3 fp16 products per f32 product against the 2.5 PFLOP/s dense fp16 peak
            roofline["note"] = "frac = algorithmic f32 FLOP/s over the f32-input MFMA peak (f32-equivalent, comparable with the default kernel's)"
            roofline["fp16_pipe"] = {"issued_flops_per_launch": 3.0 * flops, "peak_TFLOPs": 2516.6, "frac": 3.0 * ach / 2516.6}
        if col and col["total_ms"] > lg["total_ms"]:
            cf = (gru_fwd_flops if args.rnn else mlp_fwd_flops)(D, H, A) * P * (env_steps / steps)
            roofline["collector"] = {"kernel": "idqn_collect_kernel", "bound": "mfma (latency-bound in practice)", "flops_per_launch": cf,
                                     "avg_launch_us": col["avg_us"], "frac": cf / (col["avg_us"] * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS}
    elif col:
        # env-only: the collector's HBM traffic is the replay write, 4*P*D + P + 4*P + 2 bytes per env-step (+ row 0)
        per_step = 4 * P * D + P + 4 * P + 2
        byts = per_step * env_steps / steps + 4 * P * D * N
        avg_s = col["avg_us"] * 1e-6
        ach = byts / avg_s / 1e9
        roofline = {"kernel": "idqn_collect_kernel", "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": ach / PEAK_HBM_GBS, "traffic": None, "bytes_per_launch": byts, "avg_launch_us": col["avg_us"]}

    out = {
        "metric": f"env-steps/sec (whole node) {args.algo.upper()} {args.env_name.split(':')[-1].replace('-v3', '')}",
        "value": env_steps / dt,
        "unit": "env-steps/s",
        "n_gpus": world,
        "rccl_ranks": ranks_field,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": 1e3 * dt / steps,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32 via fp16 hi/lo split (2^-21 per product), fp32 accumulate - OPT-IN, not the default" if getattr(args, "split16", False) else "f32" if not (args.algo == "qmix" and args.mixer_fp16) else "f32 (mixer first layers: fp16 inputs on MFMA, fp32 accumulate - opt-in)",
        "data": "synthetic (Philox-seeded env layouts, orthogonal-init weights)",
        "config": {
            "workload": f"{args.algo.upper()} on {args.env_name.split(':')[-1].replace('-v3', '')}, {N} batched HIP envs per GPU, "
                        + (f"GRU-{H} networks (use_rnn), " if args.rnn else f"2-layer-{H} MLP, ") + f"time_limit {T}",
            "cadence": args.cadence,
            "cadence_note": ("the reference's replay ratio (32 sampled episodes per collected episode), gradient steps batched: U updates of B "
                             "episodes per round; return-vs-env-steps against the reference cadence: profiles/r02_learning_parity.md, profiles/r03_learning_parity.md")
            if args.cadence == "ratio" else ("the reference's own cadence: one update of 32 episodes per collected episode, sequentially"
                                             if args.cadence == "reference" else "collection only"),
            "envs_per_gpu": N,
            "updates_per_round": U,
            "update_batch_episodes": B,
            "sampled_episodes_per_collected_episode": (U * B) / N if N else 0,
            "replay_capacity_episodes": cap,
            "lr": hyper["lr"], "target_update_interval_or_tau": hyper["target_update_interval_or_tau"],
            "parallelism": (f"dp{world} (envs + replay sharded per GPU, RCCL grad all-reduce per update; "
                            + (f"strong scaling: {N * world} envs and a global update batch of {B * world} episodes split over the GPUs)" if strong else
                               f"weak scaling: effective update batch {world} x {B} episodes)")) if world > 1 else "1 GPU",
            "hparams": args.hparams if args.algo == "idqn" and args.cadence == "ratio" else "reference",
            #  idqn.yaml's lr 3e-4 / hard copy every 200 updates are tuned for the reference's one update per collected
            # episode; at the batched cadence they barely move the return inside a bench run (`mean_episode_return_last_round` = random play).
            # The `tuned` row of `modes` runs the same kernels at the same throughput and learns (0.84 after 1.2e8 env-steps:
            # profiles/r03_learning_parity.md); cadence=reference is the row whose learning curve matches the reference's per env-step
            "learns_at_these_hparams": (("yes (profiles/r03_learning_parity.md)" if args.hparams == "tuned" else
                                         "barely at this cadence - see the `tuned` row of `modes` (same kernels, same throughput)")
                                        if args.algo == "idqn" and args.cadence == "ratio" else None),
            "learner": "split16 (opt-in: f32 products from fp16 halves)" if getattr(args, "split16", False) else "f32",
            "mixer_first_layers": ("fp16 (opt-in)" if args.mixer_fp16 else "f32") if args.algo == "qmix" else None,
            "env_steps_timed": env_steps,
            # update cost is per SAMPLED episode (padding rows are computed), collection cost per step: env-steps/s moves with the episode
            # length the policy produces.  The timed rounds run at the epsilon below (a fresh run's schedule: close to 1, random policy,
            # every episode runs to the time limit unless the env ends it) - `modes` carries a trained-policy row next to it
            "mean_episode_length": mean_len,
            "mean_filled_rows_per_sampled_episode": sampled_rows,
            "replay_clear_stale": bool(getattr(args, "clear_stale", False)),
            "mean_episode_return_last_round": mean_return,
            "epsilon_timed_rounds": {"first": eps_timed[0], "last": eps_timed[1]},
            "policy_regime": getattr(args, "regime_note", "fresh run: epsilon-greedy near epsilon = 1 on orthogonal-init networks"),
        },
        "kernels": timing,
        "roofline": roofline,
    }
    if digest is not None:
        out["params_sha16_rank0"] = digest
    return out


if __name__ == "__main__":
    main()
