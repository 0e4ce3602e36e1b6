"""Minimal space containers.  The reference takes them from `gym` (envs/mpe/uav_dcc.py:4,
multiagent/environment.py:2); the hot path only ever reads `.shape`, `.low/.high`, `.dtype` and the
class name ("Box" selects the continuous action head, algos/algo_utils/act.py:22).  No gym needed."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape)
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape)

    def sample(self):
        return np.random.uniform(np.maximum(self.low, -1e3), np.minimum(self.high, 1e3)).astype(self.dtype)

    def __repr__(self):
        return "Box(%s, %s)" % (self.shape, self.dtype)
