"""Minimal stand-in for the gym.spaces classes the reference exposes through observation_space / action_space
(DrivingEnvironment.py:170-232, RoboCupEnvironment.py:338-342).  gym is not a dependency of this package; consumers
only use `.spaces`, `.shape`, `.nvec`, `.n`, `.low/.high` and `flatdim` (DynEnv/models/models.py:628,637)."""
from collections import OrderedDict

import numpy as np


class Space(object):
    shape = None
    dtype = None


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.low = np.full(self.shape, low, dtype=dtype)
        self.high = np.full(self.shape, high, dtype=dtype)
        self.dtype = np.dtype(dtype)

    def __repr__(self):
        return "Box(%s, %s, %s)" % (self.low.min(), self.high.max(), self.shape)


class MultiBinary(Space):
    def __init__(self, n):
        self.n = n
        self.shape = (n,)
        self.dtype = np.dtype(np.int8)

    def __repr__(self):
        return "MultiBinary(%d)" % self.n


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self.dtype = np.dtype(np.int64)

    def sample(self, rng=np.random):
        return (rng.random_sample(self.nvec.shape) * self.nvec).astype(np.int64)

    def __repr__(self):
        return "MultiDiscrete(%s)" % self.nvec.tolist()


class Dict(Space):
    def __init__(self, spaces):
        self.spaces = OrderedDict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def __repr__(self):
        return "Dict(%s)" % ", ".join("%s:%r" % kv for kv in self.spaces.items())


class Tuple(Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __len__(self):
        return len(self.spaces)

    def __repr__(self):
        return "Tuple(%s)" % ", ".join(repr(s) for s in self.spaces)


def flatdim(space):
    if isinstance(space, Box):
        return int(np.prod(space.shape))
    if isinstance(space, MultiBinary):
        return int(space.n)
    if isinstance(space, MultiDiscrete):
        return int(np.sum(space.nvec))
    if isinstance(space, (Tuple,)):
        return int(sum(flatdim(s) for s in space.spaces))
    if isinstance(space, Dict):
        return int(sum(flatdim(s) for s in space.spaces.values()))
    raise NotImplementedError(type(space))
