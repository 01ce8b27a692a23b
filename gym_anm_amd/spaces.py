"""``Box`` space: gymnasium's when it is installed, otherwise a minimal stand-in with the members
the reference relies on (``low``, ``high``, ``shape``, ``dtype``, ``contains``, ``sample``,
``seed``; anm_env.py:231,302,357,493)."""

from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the image
    from gymnasium.spaces import Box  # type: ignore
    from gymnasium import Env as GymEnv  # type: ignore

    HAVE_GYMNASIUM = True
except Exception:  # gymnasium is not installed in the build image
    HAVE_GYMNASIUM = False

    class Box:  # noqa: D401
        def __init__(self, low, high, shape=None, dtype=np.float64, seed=None):
            self.dtype = np.dtype(dtype)
            self.low = np.asarray(low, dtype=self.dtype)
            self.high = np.asarray(high, dtype=self.dtype)
            if shape is not None:
                self.low = np.broadcast_to(self.low, shape).copy()
                self.high = np.broadcast_to(self.high, shape).copy()
            self.shape = self.low.shape
            self._np_random = np.random.default_rng(seed)

        def seed(self, seed=None):
            self._np_random = np.random.default_rng(seed)
            return seed

        def contains(self, x):
            x = np.asarray(x)
            return bool(x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1e6)
            hi = np.where(np.isfinite(self.high), self.high, 1e6)
            return self._np_random.uniform(lo, hi).astype(self.dtype)

        def __repr__(self):
            return "Box(%s, %s, %s, %s)" % (self.low, self.high, self.shape, self.dtype)

    class GymEnv:  # the part of gymnasium.Env the reference uses: seeding in reset()
        metadata = {}
        np_random = None

        def reset(self, *, seed=None, options=None):
            if seed is not None or self.np_random is None:
                self.np_random = np.random.default_rng(seed)
