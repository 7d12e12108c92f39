"""TEST INFRASTRUCTURE: differentiable restatements of the feature kernels of efficientlo-net_amd/_ops.py in plain
PyTorch ops (index_select gathers, index_add, scatter_reduce ...), dtype-generic so that they run in float64.
tests/test_backward_gpu.py differentiates them with torch.autograd in double precision and checks the hand-written
backward kernels (csrc/elo_backward.hip) against those analytic gradients.  (Round 1 used these functions as the
training path itself; the product no longer contains them: training runs the HIP kernels forward and backward.)
Neighbour indices and masks carry no gradient (utils/pointnet_util.py:54-55 wraps the mask in stop_gradient, indices
are integers).

Gradient semantics follow the reference's TF graph: gather_nd -> scatter-add of the incoming
gradient, reduce_max -> arg-max routing, and the re-projection passes gradients to the scattered
VALUES (model_util.py:264-273), never to the cell indices.
"""
import math

import torch


def _cell(idx, H, W):
    return (idx[..., 0].long() * H + idx[..., 1].long()) * W + idx[..., 2].long()


_SPREAD = 8192                                      # dummy rows that absorb the adds of rows without a contribution
_spread_cache = {}


def _spread_rows(n, base, device):
    """base + (0..n-1) % _SPREAD: distinct-enough targets for rows whose contribution is zero."""
    key = (n, str(device))
    if key not in _spread_cache:
        if len(_spread_cache) > 64:
            _spread_cache.clear()
        _spread_cache[key] = torch.arange(n, device=device) % _SPREAD
    return _spread_cache[key] + base


def _index_add_live(rows, C, target, values, live):
    """zeros((rows, C)).index_add(0, target, values) for the rows where `live`; the others (their value is zero by
    construction) are sent to dummy rows instead of their nominal target.  The nominal target of a masked slot is
    cell (0,0,0) and of an invalid point the one "zero cell": as atomic adds they all queue on
    ONE row -- same-address atomics serialise at ~170 ns each on this GPU (projection) -- which was most
    of the 41 % of a training step spent in index_add_."""
    tgt = torch.where(live, target, _spread_rows(target.numel(), rows, target.device))
    out = torch.zeros((rows + _SPREAD, C), dtype=values.dtype, device=values.device).index_add_(0, tgt, values)
    return out[:rows]


class _GatherMasked(torch.autograd.Function):
    """flat (R,C), cell (n,) long, mask (n,) of 0/1 -> flat[cell] * mask[:, None]; backward: the scatter-add of TF's
    gather_nd gradient, restricted to the unmasked slots."""

    @staticmethod
    def forward(ctx, flat, cell, mask):
        ctx.save_for_backward(cell, mask)
        ctx.rows = flat.shape[0]
        return flat.index_select(0, cell) * mask.unsqueeze(-1)

    @staticmethod
    def backward(ctx, grad):
        cell, mask = ctx.saved_tensors
        g = grad * mask.unsqueeze(-1)
        return _index_add_live(ctx.rows, grad.shape[-1], cell, g, mask != 0), None, None


def _gather(grid, idx, mask):
    """tf.gather_nd(grid (B,H,W,C), idx (...,3)) * mask[..., None] -> (..., C)   (utils/pointnet_util.py:54-55).
    index_select, not advanced indexing: its backward is one atomic index_add_ (the scatter-add of TF's gather_nd
    gradient), where the backward of `flat[cell]` is a sort-based index_put that took 80 % of a training step
    (5.3 ms per call, 27 calls: tools/train_profile.py)."""
    B, H, W, C = grid.shape
    cell = _cell(idx, H, W)
    out = _GatherMasked.apply(grid.reshape(B * H * W, C), cell.reshape(-1), mask.reshape(-1).to(grid.dtype))
    return out.reshape(*cell.shape, C)


def group_concat(centre_xyz, src_xyz, src_feat, idx, mask):
    return torch.cat([_gather(src_xyz, idx, mask) - centre_xyz.unsqueeze(2), _gather(src_feat, idx, mask)], -1)


def masked_maxpool(x, mask):
    return (x * mask.unsqueeze(-1)).max(dim=2).values


def _geometry(p, g):
    diff = g - p
    euc = torch.sqrt((diff * diff).sum(-1, keepdim=True) + 1e-20)
    return torch.cat([p.expand_as(g), g, diff, euc], -1)


def cv_encode1(xyz1, feat1, xyz2_proj, feat2_proj, idx, mask):
    K = idx.shape[2]
    q = _gather(xyz2_proj, idx, mask)
    f2 = _gather(feat2_proj, idx, mask)
    return torch.cat([_geometry(xyz1.unsqueeze(2), q), feat1.unsqueeze(2).expand(-1, -1, K, -1), f2], -1)


def cv_encode2(xyz1_proj, feat1_proj, cost_proj, idx, mask):
    B, H, W, C = feat1_proj.shape
    K = idx.shape[2]
    p = xyz1_proj.reshape(B, H * W, 1, 3)
    g = _gather(xyz1_proj, idx, mask)
    rest = torch.cat([feat1_proj.reshape(B, H * W, 1, C).expand(-1, -1, K, -1), _gather(cost_proj, idx, mask)], -1)
    return _geometry(p, g), rest


def masked_softmax_pool(logits, values, mask):
    l = torch.where(mask.unsqueeze(-1) == 1.0, logits, torch.full_like(logits, -1e10))
    return (torch.softmax(l, dim=2) * values).sum(2)


def softmax_valid(feature_bnc, weight_bnc, xyz_bn3):
    valid = (xyz_bn3 != 0).any(-1, keepdim=True)
    l = weight_bnc.masked_fill(~valid, float("-inf"))
    w = torch.nan_to_num(torch.softmax(l, dim=1), nan=0.0)          # a batch element without valid points -> 0
    return (feature_bnc * w).sum(1, keepdim=True)


def projection_constants(H_input, W_input):
    d2r = math.pi / 180
    az = (360.0 / W_input) * d2r
    down, up = -24.8 * d2r, 2.0 * d2r
    vres = (up - down) / (H_input - 1)
    return az, vres, -down / vres


def _hamilton(a, b):
    a0, a1, a2, a3 = a.unbind(-1)
    b0, b1, b2, b3 = b.unbind(-1)
    return torch.stack([a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3, a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2,
                        a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1, a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0], -1)


def warp_project(xyz, feat, q, t, H, W):
    """Optional quaternion warp + ProjectPC2SphericalRing; same returns as _ops.warp_project."""
    B, N, _ = xyz.shape
    warped = None
    pts = xyz
    if q is not None:
        q = q.reshape(B, 1, 4)
        t = t.reshape(B, 1, 3)
        keep = (xyz != 0).any(-1, keepdim=True).to(xyz.dtype)
        q_inv = torch.cat([q[..., :1], -q[..., 1:]], -1) / ((q * q).sum(-1, keepdim=True) + 1e-10)
        v = _hamilton(q, torch.cat([torch.zeros_like(xyz[..., :1]), xyz], -1))
        pts = warped = (_hamilton(v, q_inv)[..., 1:] + t) * keep
    az, vres, voff = projection_constants(H, W)
    with torch.no_grad():                                    # cell indices carry no gradient
        x, y, z = pts.detach().unbind(-1)
        r = torch.sqrt(x * x + y * y + z * z)
        col = torch.nan_to_num((math.pi - torch.atan2(y, x)) / az, nan=0.0).trunc().long().clamp(0, W - 1)
        row = (H - torch.nan_to_num(torch.asin(z / r) / vres + voff, nan=0.0).trunc().long()).clamp(0, H - 1)
        cell = (torch.arange(B, device=xyz.device).view(B, 1) * H + row) * W + col
        flat_cell, flat_r = cell.reshape(-1), r.reshape(-1)
        cells = B * H * W
        # the invalid points (r = 0) all sit in one cell per image: kept out of the atomic min (they would queue on
        # one address), their cells are set to 0 by a plain indexed store afterwards
        zero = flat_r == 0
        spread = _spread_rows(flat_cell.numel(), cells, xyz.device)
        min_r = torch.full((cells + _SPREAD,), float("inf"), device=xyz.device, dtype=flat_r.dtype).scatter_reduce(
            0, torch.where(zero, spread, flat_cell), flat_r, "amin")
        min_r.index_fill_(0, torch.where(zero, flat_cell, spread), 0.0)          # (no boolean indexing: no host sync)
        min_r = min_r[:cells]
        win = flat_r == min_r[flat_cell]
        same = win.to(xyz.dtype).unsqueeze(-1)
    # losers add nothing; winning zero points add zeros to xyz (skipped) and their features to feat (kept)
    out_xyz = _index_add_live(cells, 3, flat_cell, pts.reshape(-1, 3) * same, win & ~zero).reshape(B, H, W, 3)
    out_feat = None
    if feat is not None:
        C = feat.shape[-1]
        out_feat = _index_add_live(cells, C, flat_cell, feat.reshape(-1, C) * same, win).reshape(B, H, W, C)
    return warped, out_xyz, out_feat
