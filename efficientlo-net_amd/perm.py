"""Source of the window visiting orders.

The reference draws `tf.random_shuffle(tf.range(kH*kW))` inside every operator
(utils/pointnet_util.py:45,104,193,270): a fresh permutation per sess.run.  The
grouping ops take the permutation as an INPUT, so results are a pure function
of (inputs, permutation) and the caller owns the randomness.  A PermSource
hands out one int32 device tensor per (scope, tag, KT):

  mode "fixed"  -- drawn once from the seed, then reused (inference, HIP-graph replay)
  mode "fresh"  -- re-drawn IN PLACE by reshuffle() (training; buffers keep their
                   address, so a captured HIP graph sees the new order)
  a callable    -- test hook: fn(scope, tag, KT) -> int array, e.g. the oracle's
"""
import contextlib

import numpy as np
import torch


class PermSource:
    def __init__(self, seed=0, fn=None):
        self.seed = seed
        self.fn = fn
        self._bufs = {}
        self._draws = 0
        self.generation = 0       # bumped by reshuffle(): the decoded orders a captured inference graph points at are dropped

    def _draw(self, scope, tag, KT):
        if self.fn is not None:
            return np.asarray(self.fn(scope, tag, KT), dtype=np.int32)
        rng = np.random.default_rng([self.seed, self._draws, KT])
        self._draws += 1
        return rng.permutation(KT).astype(np.int32)

    def get(self, scope, tag, KT, device):
        key = (scope, tag, KT, str(device))
        if key not in self._bufs:
            self._bufs[key] = torch.from_numpy(self._draw(scope, tag, KT).copy()).to(device)
        return self._bufs[key]

    def reshuffle(self):
        """Draw a new order into every existing buffer (same storage)."""
        for (scope, tag, KT, _dev), buf in self._bufs.items():
            buf.copy_(torch.from_numpy(self._draw(scope, tag, KT).copy()), non_blocking=True)
        self.generation += 1


_current = [PermSource()]


@contextlib.contextmanager
def default_perm_source(src):
    _current.append(src)
    try:
        yield src
    finally:
        _current.pop()


def random_shuffle(scope, tag, KT, device):
    return _current[-1].get(scope, tag, KT, device)
