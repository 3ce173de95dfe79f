"""The reference's own learner classes, imported from oracle/_ref (oracle/make_ref.py) - or straight from the reference checkout
when it is present - behind the four stubs SURVEY.md 8c lists.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the `cpu_baseline`
leg of bench.py and tests use it; it is never the thing measured as the product."""
import contextlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isfile(os.path.join(HERE, "_ref", "marlbase", "dqn", "model.py")) or os.path.isdir("/root/reference/marlbase")


def import_reference():
    """(marlbase.dqn.model, marlbase.dqn.train) of the reference, unmodified"""
    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")

    def flatdim(space):
        if isinstance(space, (tuple, list)):
            return sum(flatdim(s) for s in space)
        return int(space.n) if hasattr(space, "n") else int(np.prod(space.shape))

    spaces.flatdim = flatdim
    gym.spaces = spaces
    sys.modules.setdefault("gymnasium", gym)
    sys.modules.setdefault("gymnasium.spaces", spaces)
    sys.modules.setdefault("hydra", types.ModuleType("hydra"))
    oc = types.ModuleType("omegaconf")
    oc.DictConfig = dict
    sys.modules.setdefault("omegaconf", oc)
    sys.modules.setdefault("imageio", types.ModuleType("imageio"))
    root = os.path.join(HERE, "_ref")
    if not os.path.isfile(os.path.join(root, "marlbase", "dqn", "model.py")):
        root = "/root/reference"
    sys.dont_write_bytecode = True
    if root not in sys.path:
        sys.path.insert(0, root)
    from marlbase.dqn import model as ref_model
    from marlbase.dqn import train as ref_train

    return ref_model, ref_train, root


class Cfg(dict):
    __getattr__ = dict.__getitem__


class Box:
    def __init__(self, d):
        self.shape = (d,)


class Discrete:
    """the two things the reference asks of an action space: `.n` (flatdim) and `.sample()` (QNetwork.act, dqn/model.py:113)"""

    def __init__(self, n, rng):
        self.n, self._rng = n, rng

    def sample(self):
        return int(self._rng.integers(0, self.n))


class TupleSpace(list):
    def sample(self):
        return [s.sample() for s in self]


def reference_idqn_loop(seconds, hidden, env_name, time_limit, make_env):
    """the reference's training loop (dqn/train.py:298-312) on its own QNetwork + ReplayBuffer + _epsilon_schedule, the env being
    `make_env()` (oracle/lbf.py under the reference's wrapper semantics: lbforaging itself is not installable here); 1 thread as
    run.py:29; returns (env-steps per second over a window of `seconds` after training_start's 32 stored episodes, steps, updates, root)"""
    import time

    import torch

    ref_model, ref_train, root = import_reference()
    torch.set_num_threads(1)
    env = make_env()
    P, D, A, T = 2, 15, 6, time_limit
    rng = np.random.default_rng(1)
    obs_space = TupleSpace(Box(D) for _ in range(P))
    act_space = TupleSpace(Discrete(A, rng) for _ in range(P))
    cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=1.0, target_update_interval_or_tau=200, double_q=True,
              standardise_returns=False)  # configs/algorithm/idqn.yaml
    with contextlib.redirect_stdout(io.StringIO()):  # QNetwork.__init__ prints itself (dqn/model.py:86)
        model = ref_model.QNetwork(obs_space, act_space, cfg, [hidden, hidden], False, False, True, "cpu")
    rb = ref_train.ReplayBuffer(10000, P, obs_space, act_space, T, "cpu")
    eps_sched = ref_train._epsilon_schedule("linear", 0.5, 1.0, 0.05, 6.5, 100000)
    steps = updates = 0
    t0 = None
    while True:
        if t0 is None and rb.can_sample(32):
            t0, s0 = time.perf_counter(), steps
        if t0 is not None and time.perf_counter() - t0 >= seconds:
            break
        t, _ = ref_train._collect_trajectory(env, model, rb, eps_sched(steps), False)  # dqn/train.py:202-237, unmodified
        steps += t
        if rb.can_sample(32):
            model.update(rb.sample(32))
            updates += 1
    dt = time.perf_counter() - t0
    return (steps - s0) / dt, steps - s0, updates, dt, root
