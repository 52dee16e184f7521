"""Minimal stand-in for gym.spaces.Box (gym is not a dependency of this package).

Mirrors what stable-baselines3 reads from a VecEnv's spaces: shape, dtype, low, high, sample, contains.
The action boxes are the ones stored in the reference checkpoints."""
from dataclasses import dataclass, field

import numpy as np


@dataclass
class Box:
    low: np.ndarray
    high: np.ndarray
    shape: tuple = field(default=None)
    dtype: type = np.float32

    def __post_init__(self):
        self.low = np.asarray(self.low, dtype=self.dtype)
        self.high = np.asarray(self.high, dtype=self.dtype)
        if self.shape is None:
            self.shape = self.low.shape
        self.low = np.broadcast_to(self.low, self.shape).copy()
        self.high = np.broadcast_to(self.high, self.shape).copy()

    def sample(self, rng=None):
        rng = rng or np.random.default_rng()
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __eq__(self, other):
        return isinstance(other, Box) and self.shape == other.shape and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high)
