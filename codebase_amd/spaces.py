"""The slice of gymnasium.spaces the reference touches (gymnasium itself is not a dependency of
this package): Tuple / Box / Discrete, `flatdim` (marlbase/dqn/model.py:32-33), `.sample()`
(dqn/model.py:113), `space[i].shape` (dqn/train.py:42).  When gymnasium is installed its own
`flatdim` accepts these objects' duck type as well (`.n` / `.shape`)."""
import numpy as np


class Discrete:
    def __init__(self, n, seed=None):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return int(self._rng.integers(0, self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n

    def __repr__(self):
        return f"Discrete({self.n})"


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape if shape is not None else np.shape(low)).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.low.shape).copy()
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.shape}, {self.dtype})"


class Tuple(tuple):
    def __new__(cls, spaces):
        return super().__new__(cls, tuple(spaces))

    @property
    def spaces(self):
        return tuple(self)

    def sample(self):
        return tuple(s.sample() for s in self)

    def __repr__(self):
        return "Tuple(" + ", ".join(repr(s) for s in self) + ")"


def flatdim(space):
    if isinstance(space, Tuple):
        return sum(flatdim(s) for s in space)
    if hasattr(space, "n"):
        return int(space.n)
    return int(np.prod(space.shape))
