"""Seeded synthetic KITTI-shaped range images (no file IO).

The reference is fed real KITTI scans (kitti_dataset.py:38-103 ->
model_util.py:181-292 projection); there is no network or dataset here, so
bench/tests use this generator (SURVEY.md section 8(d) "Synthetic inputs"):
pixel (h, w) -> azimuth/elevation of a Velodyne-64-like sensor, range from a
smooth wall + ground plane, Gaussian noise, random holes, and the 35 m crop of
model_util.py:380-383.  Empty pixels are (0,0,0), exactly what the reference's
projection leaves in unhit cells.
"""
import math

import numpy as np

FOV_UP_DEG = 2.0          # model_util.py:193
FOV_DOWN_DEG = -24.8      # model_util.py:192


def range_image(H=64, W=1800, seed=0, yaw=0.0, shift=(0.0, 0.0, 0.0), hole_rate=0.05,
                noise=0.02, crop=35.0, dtype=np.float32, profile="dense", scene_seed=None):
    """One (H, W, 3) xyz range image.

    profile "dense" (default): the smooth scene with `hole_rate` uniform holes -- a 95 %-filled grid, the best case for
    every kernel regime.  profile "kitti": the DENSITY of a projected HDL-64 scan after the 35 m crop (kitti_dataset.py:38-103
    -> model_util.py:380-383, :181-292), which fills about half of 64 x 1800: returns thin out with range (hole probability
    0.25 + 0.9 (r / crop)^2), the beams that look above the horizon mostly see sky (85 % empty), two whole beam rows are dead
    (ring dropouts), one azimuth sector of 7 degrees has no return at all (an absorbing / too-near object), and a second, far
    wall segment lies beyond the crop -- together ~55 % of the grid is empty, in runs and blocks rather than as salt and
    pepper, so windows without any valid neighbour, all-masked softmaxes and empty cells are common at every level.
    `scene_seed`: the structure (dead rows, sector) is drawn from it (default `seed - seed % 2`: the two frames of a pair
    share their structure, as two consecutive scans do)."""
    rng = np.random.default_rng(seed)
    h = np.arange(H, dtype=np.float64)[:, None]
    w = np.arange(W, dtype=np.float64)[None, :]
    az = math.pi - (w + 0.5) * (2.0 * math.pi / W) + yaw
    el = np.deg2rad(FOV_UP_DEG - h * (FOV_UP_DEG - FOV_DOWN_DEG) / max(H - 1, 1))
    el = np.broadcast_to(el, (H, W))
    az = np.broadcast_to(az, (H, W))
    wall = 20.0 + 5.0 * np.sin(3.0 * az)
    with np.errstate(divide="ignore", invalid="ignore"):
        ground = np.where(el < 0, 1.73 / np.sin(-el), np.inf)
    r = np.minimum(wall, ground) + rng.normal(0.0, noise, size=(H, W))
    xyz = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], -1)
    xyz = xyz - np.asarray(shift, dtype=np.float64)
    if profile == "dense":
        holes = rng.random((H, W)) < hole_rate
    elif profile == "kitti":
        srng = np.random.default_rng(1000003 + (seed - seed % 2 if scene_seed is None else scene_seed))
        far_wall = (np.cos(az - srng.uniform(-math.pi, math.pi)) > 0.8) & (el > np.deg2rad(-4.0))   # a segment whose returns lie beyond the crop
        r = np.where(far_wall, 48.0 + rng.normal(0.0, noise, size=(H, W)), r)
        xyz = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], -1) - np.asarray(shift, dtype=np.float64)
        p = 0.25 + 0.9 * np.minimum(r / crop, 1.0) ** 2
        p = np.where(el > 0, 0.85, p)                                  # above the horizon: sky
        holes = rng.random((H, W)) < p
        dead = srng.choice(np.arange(H // 8, H), size=min(2, max(H // 8, 1)), replace=False)    # ring dropouts
        holes[dead, :] = True
        w0 = int(srng.integers(0, W))
        cols = (w0 + np.arange(max(int(round(W * 7.0 / 360.0)), 1))) % W                        # the sector wraps with the cylinder
        holes[:, cols] = True
    else:
        raise ValueError("profile is 'dense' or 'kitti' (got %r)" % (profile,))
    far = np.hypot(xyz[..., 0], xyz[..., 1]) > crop
    xyz[holes | far] = 0.0
    return xyz.astype(dtype)


def frame_pair(B=1, H=64, W=1800, seed=0, starved=None, **kw):
    """(B,H,W,3) x2: frame 2 is the same scene seen after a small ego-motion
    (yaw 0.01 rad, 0.8 m forward), with its own noise/holes (seed+1).
    `starved` (default: True for profile "kitti" with B >= 2): the LAST batch element keeps only a 2 x 3 patch of valid pixels
    in each frame -- fewer valid points than any operator's K at every level, none at the coarse ones: the all-masked /
    empty-element paths of a batch (model_util.py:319-343 softmax_valid over no valid point, pointnet_util.py:92-98)."""
    f1 = np.stack([range_image(H, W, seed=seed + 2 * b, **kw) for b in range(B)])
    f2 = np.stack([range_image(H, W, seed=seed + 2 * b + 1, yaw=0.01, shift=(0.8, 0.0, 0.0), **kw)
                   for b in range(B)])
    if starved is None:
        starved = kw.get("profile") == "kitti" and B >= 2
    if starved:
        h0, w0 = (H * 5) // 8, W // 3
        for f in (f1, f2):
            keep = f[-1, h0:h0 + 2, w0:w0 + 3].copy()
            if not keep.any():                                       # (the patch fell into a hole: put one point there)
                keep[0, 0] = (9.0, 3.0, -1.5)
            f[-1] = 0.0
            f[-1, h0:h0 + 2, w0:w0 + 3] = keep
    return f1, f2


def hw_index(B, H, W):
    """All (h, w) pairs, row-major: the numpy twin of get_hw_idx (pointnet_util.py:23-30)."""
    hh, ww = np.meshgrid(np.arange(H, dtype=np.int32), np.arange(W, dtype=np.int32), indexing="ij")
    idx = np.stack([hh, ww], -1).reshape(1, H * W, 2)
    return np.ascontiguousarray(np.broadcast_to(idx, (B, H * W, 2)))


def strided_index(B, out_h, out_w, stride_h, stride_w):
    """(h*stride_h, w*stride_w) pairs: columns 1: of get_selected_idx (model_util.py:296-316)."""
    hh, ww = np.meshgrid(np.arange(out_h, dtype=np.int32) * stride_h,
                         np.arange(out_w, dtype=np.int32) * stride_w, indexing="ij")
    idx = np.stack([hh, ww], -1).reshape(1, out_h * out_w, 2)
    return np.ascontiguousarray(np.broadcast_to(idx, (B, out_h * out_w, 2)))
