"""CPU, authoring container only: the C restatement against the reference's own
kernel bodies built for the host (oracle/_ref, see oracle/build_ref.sh), on a
seeded random sweep that includes heavy exact-tie inputs.  Skipped where
oracle/_ref has not been built (it cannot be built without /root/reference)."""
import math

import numpy as np
import pytest

from oracle import grouping as G

pytestmark = pytest.mark.skipif(not G.have_ref(), reason="oracle/_ref not built")


@pytest.mark.parametrize("seed", range(40))
def test_restatement_equals_reference_build(seed):
    rng = np.random.default_rng(seed)
    B = int(rng.integers(1, 3))
    H, W = int(rng.integers(1, 12)), int(rng.integers(6, 40))
    sh, sw = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    H2, W2 = math.ceil(H / sh), math.ceil(W / sw)
    kH = int(rng.integers(1, 8))
    kW = int(rng.integers(1, min(2 * W2, 13)))
    K = int(rng.integers(1, 20))
    lattice = seed % 2 == 0
    def cloud(h, w):
        x = rng.normal(0, 2.0, (B, h, w, 3))
        if lattice:
            x = np.round(x)          # integer lattice: many equal distances and (0,0,0) points
        x[rng.random((B, h, w)) < 0.15] = 0
        return x.astype(np.float32)
    xyz1, xyz2 = cloud(H, W), cloud(H2, W2)
    N = int(rng.integers(1, 30))
    idx = np.stack([rng.integers(0, H, (B, N)), rng.integers(0, W, (B, N))], -1).astype(np.int32)
    perm = rng.permutation(kH * kW).astype(np.int32)
    dist = float(rng.choice([0.5, 2.0, 5.0, 1000.0]))
    fc = int(rng.integers(0, 2))
    for fn in (G.fused_conv_random_k, G.fused_conv_select_k):
        a = fn(xyz1, xyz2, idx, perm, H, W, N, kH, kW, K, fc, dist, sh, sw, impl="oracle")
        b = fn(xyz1, xyz2, idx, perm, H, W, N, kH, kW, K, fc, dist, sh, sw, impl="ref")
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
