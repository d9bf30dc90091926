"""Minimal stand-ins for gymnasium.spaces.Box / Dict (gymnasium is not installed in the build image).
Only what the reference's env constructors use (envs/robot_env.py:87-100)."""
import numpy as np


class Box:
    def __init__(self, low, high, shape, dtype=np.float32):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        if np.all(np.isfinite(self.low)) and np.all(np.isfinite(self.high)):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)
        return self._rng.standard_normal(self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


class Dict(dict):
    def __init__(self, spaces):
        super().__init__(spaces)
        self.spaces = self

    def sample(self):
        return {k: s.sample() for k, s in self.items()}

    def seed(self, seed=None):
        for i, s in enumerate(self.values()):
            s.seed(None if seed is None else seed + i)

    def contains(self, x):
        return isinstance(x, dict) and x.keys() == self.keys() and all(s.contains(x[k]) for k, s in self.items())


def batch_space(space, n):
    if isinstance(space, Dict):
        return Dict({k: batch_space(s, n) for k, s in space.items()})
    b = Box(0, 0, (n,) + space.shape, space.dtype)
    b.low = np.broadcast_to(space.low, b.shape).copy()
    b.high = np.broadcast_to(space.high, b.shape).copy()
    return b
