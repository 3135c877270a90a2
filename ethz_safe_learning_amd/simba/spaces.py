"""Minimal stand-in for gym.spaces.Box (gym is not installed in this image); only what the planner path reads:
low, high, shape, dtype, is_bounded(), sample() (reference simba/policies/mpc_policy.py:17-18,45-57)."""
import numpy as np


class Box(object):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        low = np.asarray(low, dtype=dtype)
        high = np.asarray(high, dtype=dtype)
        if shape is not None:
            low = np.broadcast_to(low, shape).astype(dtype).copy()
            high = np.broadcast_to(high, shape).astype(dtype).copy()
        assert low.shape == high.shape
        self.low, self.high, self.shape, self.dtype = low, high, low.shape, np.dtype(dtype)

    def is_bounded(self):
        return bool(np.all(np.isfinite(self.low)) and np.all(np.isfinite(self.high)))

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)

    def __repr__(self):
        return 'Box%s' % (self.shape,)
