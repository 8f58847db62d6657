"""``gym.spaces.Box`` when gym is installed, else a minimal stand-in with the attributes the reference
touches (/root/reference/environment.py:221-222, SubProcVecEnv.py:43-70: ``shape``, ``dtype``, ``low``, ``high``)."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - gym is absent in the build image
    from gym.spaces import Box  # type: ignore
except Exception:  # noqa: BLE001

    class Box:  # type: ignore[no-redef]
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:
                shape = np.shape(low)
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

        def sample(self):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

        def __eq__(self, other):
            return isinstance(other, Box) and self.shape == other.shape and np.allclose(self.low, other.low) \
                and np.allclose(self.high, other.high)
