"""model_util.py operators with the reference's names and argument lists:
mul_q_point (:17-36), mul_point_q (:39-58), inv_q (:61-69), quatt2T (:72-110),
euler2quat (:112-127), mat2euler (:130-142), ProjectPC2SphericalRing (:181-292),
get_selected_idx (:296-316), softmax_valid (:319-343), PreProcess (:346-445).

The per-point quaternion algebra, the re-projection and the masked
softmax-pool run as hand-written HIP kernels (csrc/elo_features.hip); the
small per-batch pose algebra (a handful of scalars per frame pair) is plain
torch.  `warp_and_project` is the fused form of the warp idiom of
pwclo_model.py:213-232 (quaternion warp -> mask -> ProjectPC2SphericalRing).
"""
import math

import torch

from . import _ops


_HAMILTON = {}       # device -> (index of b per product term (16), sign per term (16))


def _hamilton(a, b):
    """Hamilton product over the last axis: term (k, i) of component k is sign[k][i] * a_i * b_(i xor k) -- the sixteen
    products of model_util.py:17-36 as ONE broadcast multiply and a sum over i, instead of 16 multiplies and 12
    adds on (B,1) slices (a training step launched ~1000 kernels of eight elements for its pose algebra)."""
    key = (a.device.type, a.device.index)
    if key not in _HAMILTON:
        i, k = torch.arange(16, device=a.device) % 4, torch.arange(16, device=a.device) // 4
        sign = torch.tensor([1, -1, -1, -1, 1, 1, 1, -1, 1, -1, 1, 1, 1, 1, -1, 1], dtype=torch.float32).to(a.device)
        _HAMILTON[key] = (i, i ^ k, sign)
    ia, ib, sign = _HAMILTON[key]
    terms = torch.index_select(a, -1, ia) * (torch.index_select(b, -1, ib) * sign.to(b.dtype))
    return terms.reshape(terms.shape[:-1] + (4, 4)).sum(-1)


def mul_q_point(q_a, q_b, batch_size):
    """model_util.py:17-36: q_a (B,1,4) (x) q_b (B,N,4) -> (B,N,4)."""
    return _hamilton(q_a.reshape(batch_size, 1, 4), q_b)


def mul_point_q(q_a, q_b, batch_size):
    """model_util.py:39-58: q_a (B,N,4) (x) q_b (B,1,4) -> (B,N,4)."""
    return _hamilton(q_a, q_b.reshape(batch_size, 1, 4))


def inv_q(q, batch_size):
    """model_util.py:61-69: (B,1,4) -> (B,4), conj(q) / (|q|^2 + 1e-10)."""
    q = q.squeeze(1)
    q_2 = (q * q).sum(-1, keepdim=True) + 1e-10
    return torch.cat([q[:, :1], -q[:, 1:]], -1) / q_2


def quatt2T(q, t):
    """model_util.py:72-110: 4x4 transform of quaternion q (4,) and translation t (3,)."""
    w, x, y, z = q[0], q[1], q[2], q[3]
    Nq = w * w + x * x + y * y + z * z
    s = 2.0 / Nq
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    one = torch.ones_like(w)
    T = torch.stack([torch.stack([one - (yY + zZ), xY - wZ, xZ + wY, t[0]]),
                     torch.stack([xY + wZ, one - (xX + zZ), yZ - wX, t[1]]),
                     torch.stack([xZ - wY, yZ + wX, one - (xX + yY), t[2]]),
                     torch.stack([0 * one, 0 * one, 0 * one, one])])
    return T


def euler2quat(z, y, x):
    """model_util.py:112-127."""
    z, y, x = z / 2.0, y / 2.0, x / 2.0
    cz, sz, cy, sy, cx, sx = torch.cos(z), torch.sin(z), torch.cos(y), torch.sin(y), torch.cos(x), torch.sin(x)
    return torch.stack([cx * cy * cz - sx * sy * sz, cx * sy * sz + cy * cz * sx,
                        cx * cz * sy - sx * cy * sz, cx * cy * sz + sx * cz * sy], -1)


def mat2euler(M, seq='zyx'):
    """model_util.py:130-142 on (...,3,3)."""
    cy = torch.sqrt(M[..., 2, 2] * M[..., 2, 2] + M[..., 1, 2] * M[..., 1, 2])
    z = torch.atan2(-M[..., 0, 1], M[..., 0, 0])
    y = torch.atan2(M[..., 0, 2], cy)
    x = torch.atan2(-M[..., 1, 2], M[..., 2, 2])
    return z, y, x


def ProjectPC2SphericalRing(PC, Feature, H_input, W_input):
    """model_util.py:181-292: (B,N,3[+]) points [+ (B,N,C) features] -> (B,H,W,3) [, (B,H,W,C)].
    One HIP launch pair for the whole batch (the reference loops over the batch in Python, :213)."""
    _, xyz_proj, feat_proj = _ops.warp_project(PC[..., :3], Feature, None, None, H_input, W_input)
    return (xyz_proj, feat_proj) if Feature is not None else (xyz_proj, xyz_proj)


def warp_and_project(xyz, feature, q_coarse, t_coarse, H_input, W_input, buffers=None):
    """pwclo_model.py:213-232 fused: p' = (q (x) [0,p] (x) q^-1)[1:] + t, zeroed where p == 0, then
    ProjectPC2SphericalRing(p', feature).  Returns (warped (B,N,3), xyz_proj, feat_proj).
    `buffers` (inference only): _ops.ProjectionBuffers already cleared by the pose head that produced q, t."""
    if buffers is not None:
        return _ops.warp_project(xyz, feature, q_coarse, t_coarse, H_input, W_input, buffers)
    return _ops.warp_project(xyz, feature, q_coarse, t_coarse, H_input, W_input)


_sel_cache = {}


def get_selected_idx(array, stride_h, stride_w, out_h, out_w):
    """model_util.py:296-316 -> (B,out_h,out_w,3) int32 (b, i*stride_h, j*stride_w)."""
    batch = array.shape[0]
    key = (batch, stride_h, stride_w, out_h, out_w, str(array.device))
    capturing = array.is_cuda and torch.cuda.is_current_stream_capturing()
    if key not in _sel_cache:
        dev = array.device
        hh = (torch.arange(out_h, dtype=torch.int32, device=dev) * stride_h).view(1, -1, 1, 1).expand(batch, out_h, out_w, 1)
        ww = (torch.arange(out_w, dtype=torch.int32, device=dev) * stride_w).view(1, 1, -1, 1).expand(batch, out_h, out_w, 1)
        bb = torch.arange(batch, dtype=torch.int32, device=dev).view(-1, 1, 1, 1).expand(batch, out_h, out_w, 1)
        grid = torch.cat([bb, hh, ww], -1).contiguous()
        if capturing:
            return grid                     # a tensor born inside a graph's private pool is never cached
        if len(_sel_cache) >= 256:
            _sel_cache.clear()
        _sel_cache[key] = grid
    return _sel_cache[key]


def softmax_valid(feature_bnc, weight_bnc, mask_valid):
    """model_util.py:319-343 -> (B,1,C).  `mask_valid` is either the (B,N) bool mask of the reference or --
    cheaper, no intermediate mask tensor -- the (B,N,3) xyz it is derived from (valid = any(xyz != 0))."""
    if mask_valid.dim() == 2:           # bool mask: encode as an xyz whose first component is the mask
        xyz = torch.zeros(mask_valid.shape + (3,), dtype=torch.float32, device=mask_valid.device)
        xyz[..., 0] = mask_valid.to(torch.float32)
        mask_valid = xyz
    return _ops.softmax_valid(feature_bnc, weight_bnc, mask_valid)


def PreProcess(PC_f1, PC_f2, T_gt, T_trans, T_trans_inv, aug_frame):
    """model_util.py:346-445: 35 m crop, optional augmentation of frame 1 or 2, validity re-mask,
    q_gt / t_gt of the (augmented) ground-truth transform.  Batched torch (the reference loops over B)."""
    B, N, _ = PC_f1.shape
    dev = PC_f1.device
    valid1 = (PC_f1 != 0).any(-1, keepdim=True).to(PC_f1.dtype)                                   # :357-363
    valid2 = (PC_f2 != 0).any(-1, keepdim=True).to(PC_f2.dtype)
    ones = torch.ones((B, N, 1), dtype=PC_f1.dtype, device=dev)
    p1 = torch.cat([PC_f1, ones], -1)
    p2 = torch.cat([PC_f2, ones], -1)
    p1 = torch.where(torch.linalg.norm(p1[..., :2], dim=-1, keepdim=True) > 35, torch.zeros_like(p1), p1)  # :380-383
    p2 = torch.where(torch.linalg.norm(p2[..., :2], dim=-1, keepdim=True) > 35, torch.zeros_like(p2), p2)
    aug = torch.as_tensor(aug_frame, device=dev).view(B, 1, 1)
    p1_aug = torch.matmul(p1, T_trans.transpose(1, 2))                                            # :408-410
    p2_aug = torch.matmul(p2, T_trans.transpose(1, 2))                                            # :392-394
    p1 = torch.where(aug == 1, p1_aug, p1)
    p2 = torch.where(aug == 2, p2_aug, p2)
    out1 = p1[..., :3] * valid1                                                                   # :421-422
    out2 = p2[..., :3] * valid2
    q_gt, t_gt = preprocess_gt(T_gt, T_trans, T_trans_inv, aug_frame)
    return out1, out2, q_gt, t_gt


def preprocess_gt(T_gt, T_trans, T_trans_inv, aug_frame):
    """The ground-truth half of PreProcess (model_util.py:403,:419,:427-445): the (augmented) T_gt as (q_gt (B,4),
    t_gt (B,3,1))."""
    B = T_gt.shape[0]
    aug = torch.as_tensor(aug_frame, device=T_gt.device).view(B, 1, 1)
    T = torch.where(aug == 2, torch.matmul(T_trans, T_gt),
                    torch.where(aug == 1, torch.matmul(T_gt, T_trans_inv), T_gt))                 # :403,:419
    z, y, x = mat2euler(T[:, :3, :3])
    q_gt = euler2quat(z, y, x)                                                                    # :427-428 -> (B,4)
    t_gt = T[:, :3, 3:]                                                                           # (B,3,1)
    return q_gt, t_gt


def input_stage(point_cloud, T_trans, aug_frame, H_input, W_input):
    """The point half of PreProcess + both input projections (pwclo_model.py:54-67) as ONE C-ABI call
    (`elo_input_stage`: three launches).  point_cloud (B, 2N, >=3) -> (points (2B,N,3), xyz_proj (2B,H,W,3)), frame 1
    of every batch element first.  Inference only (the reference wraps this stage in stop_gradient, :66-67)."""
    return _ops.input_stage(point_cloud, T_trans, aug_frame, H_input, W_input)
