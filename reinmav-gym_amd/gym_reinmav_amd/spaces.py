"""``Box`` space: gym's / gymnasium's when importable, otherwise a minimal stand-in with the same
attributes (low, high, shape, dtype, sample, contains) so the package works without gym installed."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - neither package is in the build image
    from gym.spaces import Box  # type: ignore
except Exception:  # pragma: no cover
    try:
        from gymnasium.spaces import Box  # type: ignore
    except Exception:

        class Box:  # noqa: D401 - duck type of gym.spaces.Box
            def __init__(self, low, high, shape=None, dtype=np.float32):
                self.dtype = np.dtype(dtype)
                self.shape = tuple(shape) if shape is not None else np.shape(low)
                self.low = np.full(self.shape, low, dtype=self.dtype)
                self.high = np.full(self.shape, high, dtype=self.dtype)
                self._rng = np.random.RandomState()

            def seed(self, seed=None):
                self._rng = np.random.RandomState(seed)
                return [seed]

            def sample(self):
                return self._rng.uniform(self.low, self.high).astype(self.dtype)

            def contains(self, x):
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

            def __repr__(self):
                return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"
