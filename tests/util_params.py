"""Shared helpers for the operator / model parity tests."""
import zlib

import numpy as np
import torch


def shuffle_fn(scope, tag, KT):
    """Deterministic visiting order per (scope, tag, KT): shared by the oracle and the product."""
    seed = zlib.crc32(("%s|%s|%d" % (scope, tag, KT)).encode())
    return np.random.default_rng(seed).permutation(KT).astype(np.int32)


def randomise(store, seed=0):
    """Give every variable a non-trivial value (BN statistics included) so that folding errors show."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda shape: torch.randn(shape, generator=g)
    u = lambda shape: torch.rand(shape, generator=g)
    with torch.no_grad():
        for name, p in list(store.params.items()) + list(store.buffers.items()):
            leaf = name.rsplit("/", 1)[-1]
            if leaf == "weights":
                continue                      # keep the xavier draw
            if leaf in ("gamma", "moving_variance"):
                p.copy_((0.5 + u(p.shape)).to(p.device))
            else:                             # biases, beta, moving_mean
                p.copy_((0.1 * r(p.shape)).to(p.device))
    store.invalidate()


def export(store):
    return {k: v.detach().cpu().numpy() for k, v in store.state_dict().items()}


def close(got, want, atol=1e-4, rtol=1e-4):
    got = got.detach().cpu().numpy() if hasattr(got, "detach") else np.asarray(got)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert (err <= tol).all(), "max err %.3e at %s (want %.6f got %.6f)" % (
        err.max(), np.unravel_index(err.argmax(), err.shape), want.flat[err.argmax()], got.flat[err.argmax()])
