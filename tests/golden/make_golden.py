"""Generate tests/golden/grouping_*.npz and large_digests.json.

Runs ONLY in the authoring container: it executes the REFERENCE's grouping
kernels (oracle/_ref/libelo_ref.so, built from /root/reference by
oracle/build_ref.sh) on seeded inputs and stores inputs + the four reference
outputs.  The fixtures are data; no reference source is stored.

    python tests/golden/make_golden.py

Cases follow SURVEY.md section 8(c): grids {(4,7),(8,32),(16,225)}, strides
{(1,1),(1,2),(2,2),(4,8)}, windows {1x5,3x5,5x9,7x15,11x41}, K in
{4,6,8,16,32}, flag_copy {0,1}, distances {0.5,4,1000}, hole rates {0,0.1,1},
permutations identity / reversed / shuffled, plus integer-lattice clouds that
force exact distance ties (select-k swap order) and the reference demo (KAT-1,
fused_conv_random_k.py:100-127).
"""
import hashlib
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import grouping as G  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def smooth_cloud(rng, B, H, W, hole_rate, lattice):
    """Range-image-like cloud; lattice=True snaps to a 0.5 m grid => exact ties."""
    h = np.arange(H)[:, None] / max(H - 1, 1)
    w = np.arange(W)[None, :] / W
    az = 2 * np.pi * w
    r = 6.0 + 2.0 * np.sin(3 * az) + 1.5 * h
    xyz = np.stack([r * np.cos(az), r * np.sin(az), np.broadcast_to(-1.5 * h + 0.3, (H, W))], -1)
    xyz = np.broadcast_to(xyz, (B, H, W, 3)) + rng.normal(0, 0.05, (B, H, W, 3))
    if lattice:
        xyz = np.round(xyz * 2.0) / 2.0
    xyz = xyz.astype(np.float32)
    xyz[rng.random((B, H, W)) < hole_rate] = 0.0
    return xyz


def build_case(c):
    rng = np.random.default_rng(c["seed"])
    B, H, W = c["B"], c["H"], c["W"]
    sh, sw = c["stride"]
    H2, W2 = math.ceil(H / sh), math.ceil(W / sw)
    xyz1 = smooth_cloud(rng, B, H, W, c["holes"], c["lattice"])
    if (sh, sw) == (1, 1) and c["same"]:
        xyz2 = xyz1.copy()
    else:
        xyz2 = smooth_cloud(rng, B, H2, W2, c["holes"], c["lattice"])
    if c["centres"] == "all":
        hh, ww = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    else:  # strided subset, like down_conv
        hh, ww = np.meshgrid(np.arange(0, H, 2), np.arange(0, W, 3), indexing="ij")
    idx = np.stack([hh, ww], -1).reshape(1, -1, 2).astype(np.int32)
    idx = np.ascontiguousarray(np.broadcast_to(idx, (B,) + idx.shape[1:]))
    kH, kW = c["window"]
    KT = kH * kW
    if c["perm"] == "identity":
        perm = np.arange(KT)
    elif c["perm"] == "reversed":
        perm = np.arange(KT)[::-1]
    else:
        perm = rng.permutation(KT)
    return xyz1, xyz2, idx, np.ascontiguousarray(perm, dtype=np.int32)


def cases():
    rng = np.random.default_rng(20250103)
    grids = [(4, 7), (8, 32), (16, 225)]
    strides = [(1, 1), (1, 2), (2, 2), (4, 8)]
    windows = [(1, 5), (3, 5), (5, 9), (7, 15), (11, 41)]
    out = []
    n = 0
    for op in ("random", "select"):
        for gi, (H, W) in enumerate(grids):
            for wi, (kH, kW) in enumerate(windows):
                for rep in range(2):
                    stride = strides[int(rng.integers(len(strides)))]
                    W2 = math.ceil(W / stride[1])
                    if kW // 2 > W2:          # single wrap only (fused_conv_g.cu:89-97)
                        stride = (1, 1)
                    if kW // 2 > W:           # precondition violated => reference reads out of bounds
                        continue
                    out.append(dict(
                        name="%s_%03d" % (op, n), op=op, B=int(rng.integers(1, 3)), H=H, W=W,
                        stride=stride, window=(kH, kW), K=int(rng.choice([4, 6, 8, 16, 32])),
                        flag_copy=int(rng.random() < 0.25),
                        distance=float(rng.choice([0.5, 4.0, 1000.0])),
                        holes=float(rng.choice([0.0, 0.1, 0.1, 1.0] if rep else [0.0, 0.1])),
                        lattice=bool(rng.random() < 0.4), same=bool(rng.random() < 0.5),
                        centres=str(rng.choice(["all", "strided"])),
                        perm=str(rng.choice(["identity", "reversed", "shuffle"])),
                        seed=1000 + n))
                    n += 1
    return out


def run(c, arrays):
    xyz1, xyz2, idx, perm = arrays
    fn = G.fused_conv_random_k if c["op"] == "random" else G.fused_conv_select_k
    kH, kW = c["window"]
    return fn(xyz1, xyz2, idx, perm, c["H"], c["W"], idx.shape[1], kH, kW, c["K"], c["flag_copy"],
              c["distance"], c["stride"][0], c["stride"][1], impl="ref")


def digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    assert G.have_ref(), "build oracle/_ref first (make -C oracle)"
    meta = []
    blobs = {}
    for c in cases():
        arrays = build_case(c)
        sel, valid, indis, mask = run(c, arrays)
        k = c["name"]
        blobs[k + "/xyz1"], blobs[k + "/xyz2"], blobs[k + "/idx_n2"], blobs[k + "/random_hw"] = arrays
        blobs[k + "/sel"] = sel
        # valid_* are prefix-ones (Appendix A.2): store the two counts, not KT floats per centre
        assert (np.diff(valid[..., 0], axis=-1) <= 0).all() and (np.diff(indis[..., 0], axis=-1) <= 0).all()
        blobs[k + "/n_valid"] = valid[..., 0].sum(-1).astype(np.int32)
        blobs[k + "/n_indis"] = indis[..., 0].sum(-1).astype(np.int32)
        blobs[k + "/mask"] = mask
        meta.append(c)
    # KAT-1: the reference's own __main__ demo shape with the identity permutation
    H, W = 4, 7
    pc = np.tile(np.arange(H * W, dtype=np.float32).reshape(1, H, W, 1), (1, 1, 1, 3))
    idx = np.array([[[0, 0], [0, 1]]], np.int32)
    perm = np.arange(5, dtype=np.int32)
    for name, op, fc, dist in (("kat1_random", "random", 0, 200.0), ("kat1_select", "select", 0, 200.0),
                               ("kat1_select_copy", "select", 1, 0.5)):
        c = dict(name=name, op=op, B=1, H=H, W=W, stride=(1, 1), window=(1, 5), K=8, flag_copy=fc,
                 distance=dist)
        sel, valid, indis, mask = run(c, (pc, pc, idx, perm))
        blobs[name + "/xyz1"], blobs[name + "/xyz2"] = pc, pc
        blobs[name + "/idx_n2"], blobs[name + "/random_hw"] = idx, perm
        blobs[name + "/sel"], blobs[name + "/mask"] = sel, mask
        blobs[name + "/n_valid"] = valid[..., 0].sum(-1).astype(np.int32)
        blobs[name + "/n_indis"] = indis[..., 0].sum(-1).astype(np.int32)
        meta.append(c)
    np.savez_compressed(os.path.join(HERE, "grouping_cases.npz"), **blobs)
    with open(os.path.join(HERE, "grouping_cases.json"), "w") as f:
        json.dump(meta, f, indent=1)

    # large cases: digests only (inputs come from the seeded synth generator)
    import importlib
    synth = importlib.import_module("efficientlo-net_amd.synth")
    large = []
    for (H, W, op, win, K, dist, seed) in [
            (64, 1800, "random", (9, 15), 16, 0.5, 7),      # BASELINE config 1
            (64, 1800, "select", (5, 15), 32, 1000.0, 7),
            (128, 2048, "random", (9, 15), 16, 0.5, 11),    # config 5 resolution
            (128, 2048, "select", (3, 9), 8, 1000.0, 11)]:
        f1, f2 = synth.frame_pair(1, H, W, seed=seed)
        idx = synth.hw_index(1, H, W)
        perm = np.random.default_rng(seed).permutation(win[0] * win[1]).astype(np.int32)
        c = dict(op=op, H=H, W=W, stride=(1, 1), window=win, K=K, flag_copy=0, distance=dist)
        sel, valid, indis, mask = run(c, (f1, f2, idx, perm))
        large.append(dict(H=H, W=W, op=op, window=win, K=K, distance=dist, seed=seed,
                          inputs_sha256=digest(f1, f2, idx, perm), sel_sha256=digest(sel),
                          mask_sha256=digest(mask),
                          counts_sha256=digest(valid.sum(2).astype(np.int32), indis.sum(2).astype(np.int32))))
    with open(os.path.join(HERE, "large_digests.json"), "w") as f:
        json.dump(large, f, indent=1)
    print("cases:", len(meta), "large:", len(large))


if __name__ == "__main__":
    main()
