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
  pooled        -- enable_pool(R, lanes): every order tensor becomes a slice of ONE flat buffer per lane, R pre-drawn
                   versions of it sit in a device pool, and refresh() -- one tiny launch, captured at the head of a lane's
                   hipGraph -- copies the next version in (and decodes it): every REPLAY walks fresh orders, like
                   every sess.run of the reference, without recapture
"""
import contextlib

import numpy as np
import torch


class PermSource:
    def __init__(self, seed=0, fn=None):
        self.seed = seed
        self.fn = fn
        self._bufs = {}
        self._kernel = {}         # key -> (kernel_h, kernel_w): what the pooled refresh needs to decode an order
        self._pool = None
        self.active_lane = 0      # which lane's buffers get() hands out in pooled mode (model.capture sets it per lane)
        self.tail_armed = False   # model.capture: the forward being recorded ends with "load the next replay's orders"
        self._draws = 0
        self.generation = 0       # bumped by reshuffle(): the decoded orders a captured inference graph points at are dropped

    def _draw(self, scope, tag, KT):
        if self.fn is not None:
            return np.asarray(self.fn(scope, tag, KT), dtype=np.int32)
        rng = np.random.default_rng([self.seed, self._draws, KT])
        self._draws += 1
        return rng.permutation(KT).astype(np.int32)

    def get(self, scope, tag, KT, device, kernel_size=None):
        key = (scope, tag, KT, str(device))
        if kernel_size is not None:
            self._kernel[key] = (int(kernel_size[0]), int(kernel_size[1]))
        if self._pool is not None:
            if key not in self._pool["slices"]:
                raise RuntimeError("order %r was not part of the forward the pool was built from (enable_pool after a warm-up forward)" % (key,))
            off = self._pool["slices"][key]
            return self._pool["flat"][self.active_lane][off:off + KT]
        if key not in self._bufs:
            self._bufs[key] = torch.from_numpy(self._draw(scope, tag, KT).copy()).to(device)
        return self._bufs[key]

    # -- fresh orders per replay of a captured graph ---------------------------------------------------------------
    def enable_pool(self, versions, lanes, device):
        """Re-home every order tensor handed out so far (one warm-up forward must have run) into per-lane flat buffers
        and pre-draw `versions` contents.  Bumps `generation` (graphs captured before are stale)."""
        keys = [k for k in self._bufs if k[3] == str(device)]
        missing = [k for k in keys if k not in self._kernel]
        if missing:
            raise RuntimeError("orders without a recorded window shape: %r" % missing[:3])
        slices, off = {}, 0
        for k in keys:
            slices[k] = off
            off += (k[2] + 3) & ~3                     # 16-byte aligned slices
        total = off
        rows = np.zeros((versions, total), np.int32)
        entry_of = np.zeros(total, np.int32)
        table = np.zeros((len(keys), 4), np.int32)
        for e, k in enumerate(keys):
            kH, kW = self._kernel[k]
            table[e] = (slices[k], k[2], kH, kW)
            entry_of[slices[k]:slices[k] + ((k[2] + 3) & ~3)] = e
            for v in range(versions):
                rows[v, slices[k]:slices[k] + k[2]] = self._draw(k[0], k[1], k[2])
        dev = torch.device(device)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self._pool = {"slices": slices, "total": total, "versions": versions, "rows": t(rows), "host_rows": rows,
                      "entry_of": t(entry_of), "table": t(table), "n": len(keys),
                      "flat": [torch.zeros(total, dtype=torch.int32, device=dev) for _ in range(lanes)],
                      "decoded": [torch.zeros(total, dtype=torch.int32, device=dev) for _ in range(lanes)],
                      "cursor": [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(lanes)]}
        self.generation += 1
        for lane in range(lanes):                      # every lane starts on version 0 (its cursor then points at 1)
            self.refresh(lane)

    def refresh_args(self, lane=None):
        """elo_perm_refresh_args of a lane's buffers."""
        from . import _lib as L
        P = self._pool
        lane = self.active_lane if lane is None else lane
        return L.PermRefreshArgs(P["rows"].data_ptr(), P["versions"], P["total"], P["cursor"][lane].data_ptr(),
                                 P["flat"][lane].data_ptr(), P["decoded"][lane].data_ptr(), P["entry_of"].data_ptr(),
                                 P["table"].data_ptr(), P["n"])

    def refresh(self, lane=None):
        """The next pooled version into lane's buffers (one launch on the current stream; capturable)."""
        from . import _lib as L
        L.call("elo_perm_refresh", self.refresh_args(lane), self._pool["rows"])

    def pooled_decoded(self, order):
        """The decoded (dh, dw) form of a pooled order tensor (a slice of the lane's decoded buffer), or None."""
        if self._pool is None:
            return None
        for lane, flat in enumerate(self._pool["flat"]):
            off = (order.data_ptr() - flat.data_ptr()) // 4
            if 0 <= off < self._pool["total"] and order.device == flat.device:
                return self._pool["decoded"][lane][off:off + order.numel()]
        return None

    def pooled_version(self, version):
        """{(scope, tag, KT): int32 array} of pooled version `version` (host copy): what replay number `version` of a lane
        (counted from the lane's first replay after capture, modulo the pool size) walks."""
        P = self._pool
        row = P["host_rows"][version % P["versions"]]
        return {(k[0], k[1], k[2]): row[off:off + k[2]].copy() for k, off in P["slices"].items()}

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


def random_shuffle(scope, tag, KT, device, kernel_size=None):
    return _current[-1].get(scope, tag, KT, device, kernel_size)


def pooled_decoded(order):
    return _current[-1].pooled_decoded(order)


def tail_refresh_args():
    """The side job of a captured forward's last launch (the l0 pose head): elo_perm_refresh_args of the active lane while
    model.capture records a graph with fresh_orders, else None."""
    src = _current[-1]
    return src.refresh_args() if (src._pool is not None and src.tail_armed) else None
