"""oracle/ops_np.py -- numpy (fp32) restatement of the reference's Python operators.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Every function follows the cited lines of /root/reference one statement at a
time, with the stock TensorFlow ops replaced by their documented numpy
meaning (tf.gather_nd -> fancy indexing, tf.scatter_nd -> np.add.at (adds
duplicates), tf.nn.softmax -> exp(x-max)/sum, tf.contrib.layers.batch_norm in
inference mode -> (x-mean)/sqrt(var+1e-3)*gamma+beta).  The two custom ops come
from oracle/grouping.py (the C restatement pinned against the reference build).

PARITY UNPINNED for this file: TensorFlow 1.12 is not installable here and the
reference ships neither tests nor a weights blob, so these functions are pinned
only by (a) the two known-answer demos of the reference (KAT-2,
model_util.py:449-481) and (b) line-by-line review.  The grouping ops they call
ARE pinned (oracle/_ref, tests/golden).

Parameters live in a plain dict keyed by the TF variable names of the shipped
checkpoint index: '<scope>/weights' (Cin,Cout) [the
leading 1x1 / 1 kernel dims are dropped], '<scope>/biases', and for BN scopes
'<scope>/bn/{beta,gamma,moving_mean,moving_variance}'.
Permutations ("tf.random_shuffle(tf.range(KT))") are supplied by the caller
through `shuffle(scope, tag, KT) -> int32[KT]`.
"""
import math

import numpy as np

from . import grouping as G

F = np.float32
BN_EPS = F(1e-3)     # tf.contrib.layers.batch_norm default epsilon (utils/tf_util.py:526-531)

# Feature STORAGE type (BASELINE configs[2]: fp16 features in HBM, fp32 arithmetic).  With fp16 every feature tensor an
# operator of the product hands to the next one through HBM is rounded to fp16 -- the outputs of down_conv,
# flow_predictor, the projection, and both halves of cost_volume and up_conv (the product runs each half as one kernel).
# Geometry (xyz) and everything inside an operator stay fp32.  The reference itself has no fp16 mode.
_STORE = [None]


class feature_storage:
    """`with feature_storage(np.float16):` -- round the operator outputs listed above to fp16."""

    def __init__(self, dtype):
        self.dtype = None if dtype in (None, np.float32) else dtype

    def __enter__(self):
        _STORE.append(self.dtype)
        return self

    def __exit__(self, *exc):
        _STORE.pop()


def _st(x):
    if _STORE[-1] is None:
        return x.astype(F)
    stored = x.astype(_STORE[-1])
    # the ROUNDING of a stored feature is a discrete decision too (one fp16 ulp is 5e-4 relative: a value that lands on the
    # other side of a rounding boundary moves everything downstream by more than the parity tolerance): the trace keeps the
    # stored bit patterns, so the parity statistic can count them
    _rec("store", stored.view(np.uint16) if stored.dtype == np.float16 else stored)
    return stored.astype(F)


# DISCRETE decisions of a forward, recorded for the parity statistic (tests/test_parity_flips_gpu.py): every neighbour
# index / mask tensor the two grouping ops return, every point -> cell assignment (and per-cell winner set) of a
# projection and -- with fp16 feature storage -- the bit pattern of every stored feature, in call order, with the level
# marker get_model_from_projection sets.  Two oracle runs whose traces are equal
# took the same discrete decisions everywhere; what is left between them is continuous in the inputs.
_TRACE = [None]


class discrete_trace:
    """`with discrete_trace() as tr:` -- tr.events = [(level, kind, array), ...] in call order."""

    def __init__(self):
        self.events, self.level = [], None

    def __enter__(self):
        _TRACE.append(self)
        return self

    def __exit__(self, *exc):
        _TRACE.pop()


def _rec(kind, *arrays):
    tr = _TRACE[-1]
    if tr is not None:
        for a in arrays:
            tr.events.append((tr.level, kind, np.array(a, copy=True)))


def _grouped(out):
    """Pass a grouping op's 4-tuple through, recording indices and mask."""
    _rec("group", out[0], out[3])
    return out


# --------------------------------------------------------------------------- layers
def conv(params, scope, x, bn=True, relu=True):
    """1x1 conv + bias [+ BN(inference)] [+ ReLU]: utils/tf_util.py:120-185 (conv2d), :52-115 (conv1d)."""
    W = params[scope + "/weights"]
    y = x.astype(F) @ W + params[scope + "/biases"]
    if bn:
        g, b = params[scope + "/bn/gamma"], params[scope + "/bn/beta"]
        mu, var = params[scope + "/bn/moving_mean"], params[scope + "/bn/moving_variance"]
        y = (y - mu) / np.sqrt(var + BN_EPS) * g + b
    if relu:
        y = np.maximum(y, F(0))
    return y.astype(F)


def conv_train(params, scope, x, bn_decay=0.9, relu=True):
    """The TRAINING form of a batch-normalised layer: 1x1 conv + bias, batch norm with BATCH statistics over every axis
    but the last, ReLU (utils/tf_util.py:120-185 conv2d with bn=True; :512-531 batch_norm_template =
    tf.contrib.layers.batch_norm(center, scale, is_training, decay=bn_decay, updates_collections=None): the fused kernel it
    dispatches to for these NHWC maps normalises with the biased batch variance and feeds the UNBIASED one to the moving
    average, moving <- decay * moving + (1 - decay) * batch).  Moments are accumulated in float64.
    Returns (y fp32, new moving_mean, new moving_variance)."""
    W = params[scope + "/weights"]
    z = (x.astype(F) @ W + params[scope + "/biases"]).astype(F)
    flat = z.reshape(-1, z.shape[-1]).astype(np.float64)
    n = flat.shape[0]
    mean, var = flat.mean(0), flat.var(0)
    y = (z - mean.astype(F)) * (1.0 / np.sqrt(var + np.float64(BN_EPS))).astype(F) * params[scope + "/bn/gamma"] + params[scope + "/bn/beta"]
    if relu:
        y = np.maximum(y, F(0))
    d = np.float64(bn_decay)
    mm = d * params[scope + "/bn/moving_mean"].astype(np.float64) + (1 - d) * mean
    mv = d * params[scope + "/bn/moving_variance"].astype(np.float64) + (1 - d) * var * (n / max(n - 1, 1))
    return y.astype(F), mm.astype(F), mv.astype(F)


def gather_nd(grid, idx):
    """tf.gather_nd with (...,3) (b,h,w) indices into a (B,H,W,C) tensor."""
    return grid[idx[..., 0], idx[..., 1], idx[..., 2]]


def softmax(x, axis):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return (e / e.sum(axis=axis, keepdims=True)).astype(F)


# --------------------------------------------------------------------------- index helpers
def get_hw_idx(B, H, W):
    """utils/pointnet_util.py:23-30."""
    hh, ww = np.meshgrid(np.arange(H, dtype=np.int32), np.arange(W, dtype=np.int32), indexing="ij")
    idx = np.stack([hh, ww], -1).reshape(1, H * W, 2)
    return np.ascontiguousarray(np.broadcast_to(idx, (B, H * W, 2)))


def get_selected_idx(B, stride_h, stride_w, out_h, out_w):
    """model_util.py:296-316 -> (B,out_h,out_w,3) (b, i*stride_h, j*stride_w)."""
    hh, ww = np.meshgrid(np.arange(out_h, dtype=np.int32) * stride_h,
                         np.arange(out_w, dtype=np.int32) * stride_w, indexing="ij")
    bb = np.broadcast_to(np.arange(B, dtype=np.int32)[:, None, None], (B, out_h, out_w))
    return np.ascontiguousarray(np.stack([bb, np.broadcast_to(hh, bb.shape), np.broadcast_to(ww, bb.shape)], -1))


# --------------------------------------------------------------------------- set-conv
def down_conv(params, shuffle, xyz_proj, points_proj, selected_idx, K_sample, kernel_size, distance, mlp, scope):
    """utils/pointnet_util.py:179-250 (mlp2=None, pooling='max', NHWC)."""
    B, H, W, _ = xyz_proj.shape
    idx_n2 = selected_idx.reshape(B, -1, 3)
    n = idx_n2.shape[1]
    perm = shuffle(scope, "random_HW", kernel_size[0] * kernel_size[1])
    sel, _, _, mask = _grouped(G.fused_conv_random_k(xyz_proj, xyz_proj, idx_n2[:, :, 1:], perm, H, W, n, kernel_size[0],
                                                     kernel_size[1], K_sample, 0, distance, 1, 1))          # :197-199
    new_xyz_group = gather_nd(xyz_proj, sel) * mask                                               # :203
    new_points_group = gather_nd(points_proj, sel) * mask                                         # :204
    new_xyz_proj = gather_nd(xyz_proj, selected_idx)                                              # :206
    new_xyz = new_xyz_proj.reshape(B, -1, 3)
    xyz_diff = new_xyz_group - new_xyz[:, :, None, :]                                             # :211
    x = np.concatenate([xyz_diff, new_points_group], -1)                                          # :213
    for i, _c in enumerate(mlp):
        x = conv(params, "%s/conv%d" % (scope, i), x)                                             # :217-222
    x = x * mask                                                                                  # :224
    return _st(x.max(axis=2)), new_xyz_proj                                                       # :230,:248


# --------------------------------------------------------------------------- cost volume
def cost_volume(params, shuffle, warped_xyz1_proj, xyz2_proj, points1_proj, points2_proj, kernel_size1, kernel_size2,
                nsample, nsample_q, distance, mlp1, mlp2, scope, stage1=None, taps=None):
    """utils/pointnet_util.py:33-149.  Test hooks (not reference arguments): `taps` (a dict) receives the stage-1 tensor
    (B,H,W,64) under "stage1"; `stage1` replaces it for stage 2 -- with fp16 feature storage the product keeps that tensor
    in HBM as fp16, and an element-wise comparison of the FINAL tensor is made with the product's own stored intermediate
    (one fp16 ulp of difference there is allowed and moves the final value by more than 1e-4)."""
    B, H, W, _ = warped_xyz1_proj.shape
    warped_xyz1 = warped_xyz1_proj.reshape(B, H * W, -1)
    points1 = points1_proj.reshape(B, H * W, -1)
    perm_q = shuffle(scope, "random_HW_q", kernel_size2[0] * kernel_size2[1])
    idx_hw = get_hw_idx(B, H, W)
    qi_idx, _, _, valid_mask = _grouped(G.fused_conv_select_k(warped_xyz1_proj, xyz2_proj, idx_hw, perm_q, H, W, H * W,
                                                              kernel_size2[0], kernel_size2[1], nsample_q, 0, 1000, 1, 1))  # :49-51
    qi_xyz_grouped = gather_nd(xyz2_proj, qi_idx) * valid_mask                                    # :54
    qi_points_grouped = gather_nd(points2_proj, qi_idx) * valid_mask                              # :55
    pi_xyz = np.broadcast_to(warped_xyz1[:, :, None, :], qi_xyz_grouped.shape)                    # :57
    pi_points = np.broadcast_to(points1[:, :, None, :], qi_points_grouped.shape[:3] + (points1.shape[-1],))  # :58
    diff = qi_xyz_grouped - pi_xyz                                                                # :60
    euc = np.sqrt((diff * diff).sum(-1, keepdims=True) + F(1e-20))                                # :61
    xyz_cat = np.concatenate([pi_xyz, qi_xyz_grouped, diff, euc], -1).astype(F)                   # :62
    feat_cat = np.concatenate([xyz_cat, pi_points, qi_points_grouped], -1).astype(F)              # :65-66
    x = feat_cat
    for j, _c in enumerate(mlp1):
        x = conv(params, "%s/CV_%d" % (scope, j), x)                                              # :72-76
    enc = conv(params, scope + "/CV_xyz", xyz_cat)                                                # :79-82
    cat = np.concatenate([enc, x], -1)                                                            # :84
    for j, _c in enumerate(mlp2):
        cat = conv(params, "%s/sum_CV_%d" % (scope, j), cat)                                      # :86-90
    logits = np.where(valid_mask == 1.0, cat, F(-1e10)).astype(F)                                 # :92-94
    WQ = softmax(logits, 2)                                                                       # :96
    pi_feat1_new = _st((WQ * x).sum(2)).reshape(B, H, W, -1)                                      # :97-100
    if taps is not None:
        taps["stage1"] = pi_feat1_new
    if stage1 is not None:
        pi_feat1_new = np.asarray(stage1, F).reshape(pi_feat1_new.shape)

    perm_p = shuffle(scope, "random_HW_p", kernel_size1[0] * kernel_size1[1])
    pc_idx, _, _, valid_mask2 = _grouped(G.fused_conv_random_k(warped_xyz1_proj, warped_xyz1_proj, idx_hw, perm_p, H, W, H * W,
                                                               kernel_size1[0], kernel_size1[1], nsample, 0, distance, 1, 1))  # :106-108
    pc_points_grouped = gather_nd(pi_feat1_new, pc_idx) * valid_mask2                             # :110
    pc_xyz_grouped = gather_nd(warped_xyz1_proj, pc_idx) * valid_mask2                            # :111
    pc_xyz_new = np.broadcast_to(warped_xyz1[:, :, None, :], pc_xyz_grouped.shape)                # :114
    pc_points_new = np.broadcast_to(points1[:, :, None, :], pc_xyz_grouped.shape[:3] + (points1.shape[-1],))
    pc_diff = pc_xyz_grouped - pc_xyz_new                                                         # :118
    pc_euc = np.sqrt((pc_diff * pc_diff).sum(-1, keepdims=True) + F(1e-20))
    pc_xyz_cat = np.concatenate([pc_xyz_new, pc_xyz_grouped, pc_diff, pc_euc], -1).astype(F)      # :120
    pc_enc = conv(params, scope + "/sum_xyz_encoding", pc_xyz_cat)                                # :123-126
    pc_cat = np.concatenate([pc_enc, pc_points_new, pc_points_grouped], -1).astype(F)             # :129
    for j, _c in enumerate(mlp2):
        pc_cat = conv(params, "%s/sum_cost_volume_%d" % (scope, j), pc_cat)                       # :131-135
    logits2 = np.where(valid_mask2 == 1.0, pc_cat, F(-1e10)).astype(F)                            # :137-140
    WP = softmax(logits2, 2)                                                                      # :142
    return _st((WP * pc_points_grouped).sum(2))                                                   # :144-146


# --------------------------------------------------------------------------- flow predictor / up-conv
def flow_predictor(params, points_f1, upsampled_feat, cost_vol, mlp, scope):
    """utils/pointnet_util.py:153-175."""
    parts = [points_f1] + ([upsampled_feat] if upsampled_feat is not None else []) + \
            ([cost_vol] if cost_vol is not None else [])
    x = np.concatenate(parts, -1)[:, :, None, :]
    for i, _c in enumerate(mlp):
        x = conv(params, "%s/conv_predictor%d" % (scope, i), x)
    return _st(x[:, :, 0, :])


def up_conv(params, shuffle, xyz1_proj, xyz2_proj, feat1_proj, feat2_proj, kernel_size, stride_h, stride_w, nsample,
            distance, mlp, mlp2, scope, pooled=None, taps=None):
    """utils/pointnet_util.py:254-316.  Test hooks as in cost_volume: `taps["pooled"]` receives the max-pooled stage-1
    tensor (B,N,mlp[-1]), `pooled` replaces it for stage 2."""
    B, H, W, _ = xyz1_proj.shape
    xyz1 = xyz1_proj.reshape(B, H * W, -1)
    points1 = feat1_proj.reshape(B, H * W, -1)
    idx_hw = get_hw_idx(B, H, W)
    perm = shuffle(scope, "random_HW", kernel_size[0] * kernel_size[1])
    sel, _, _, mask = _grouped(G.fused_conv_random_k(xyz1_proj, xyz2_proj, idx_hw, perm, H, W, H * W, kernel_size[0],
                                                     kernel_size[1], nsample, 0, distance, stride_h, stride_w))   # :272-274
    up_grouped = gather_nd(xyz2_proj, sel) * mask                                                 # :277
    up_points_grouped = gather_nd(feat2_proj, sel) * mask                                         # :278
    diff = up_grouped - xyz1[:, :, None, :]                                                       # :283
    x = np.concatenate([diff, up_points_grouped], -1).astype(F)                                   # :284
    for j, _c in enumerate(mlp):
        x = conv(params, "%s/up_1_%d" % (scope, j), x)                                            # :289-293
    x = x * mask                                                                                  # :295
    up_feat = _st(x.max(axis=2))                                                                  # :298
    if taps is not None:
        taps["pooled"] = up_feat
    if pooled is not None:
        up_feat = np.asarray(pooled, F).reshape(up_feat.shape)
    y = np.concatenate([up_feat, points1], -1)[:, :, None, :]                                     # :303-305
    for i, _c in enumerate(mlp2):
        y = conv(params, "%s/up_2_%d" % (scope, i), y)                                            # :307-311
    return _st(y[:, :, 0, :])


# --------------------------------------------------------------------------- quaternions
def _hamilton(a, b):
    """Component formulas shared by mul_q_point / mul_point_q (model_util.py:21-34, :43-56)."""
    r0 = a[..., 0] * b[..., 0] - a[..., 1] * b[..., 1] - a[..., 2] * b[..., 2] - a[..., 3] * b[..., 3]
    r1 = a[..., 0] * b[..., 1] + a[..., 1] * b[..., 0] + a[..., 2] * b[..., 3] - a[..., 3] * b[..., 2]
    r2 = a[..., 0] * b[..., 2] - a[..., 1] * b[..., 3] + a[..., 2] * b[..., 0] + a[..., 3] * b[..., 1]
    r3 = a[..., 0] * b[..., 3] + a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1] + a[..., 3] * b[..., 0]
    return np.stack([r0, r1, r2, r3], -1).astype(F)


def mul_q_point(q_a, q_b, batch_size):
    """model_util.py:17-36: q_a (B,1,4) broadcast against q_b (B,N,4)."""
    return _hamilton(q_a.reshape(batch_size, 1, 4), q_b)


def mul_point_q(q_a, q_b, batch_size):
    """model_util.py:39-58: q_a (B,N,4) against q_b (B,1,4)."""
    return _hamilton(q_a, q_b.reshape(batch_size, 1, 4))


def inv_q(q, batch_size):
    """model_util.py:61-69: (B,1,4) -> (B,4)."""
    q = q[:, 0, :]
    q2 = (q * q).sum(-1, keepdims=True) + F(1e-10)
    return (np.concatenate([q[:, :1], -q[:, 1:]], -1) / q2).astype(F)


def warp(xyz, q_coarse, t_coarse):
    """pwclo_model.py:213-227: xyz (B,N,3), q (B,1,4), t (B,1,3) -> warped (B,N,3)."""
    B = xyz.shape[0]
    q_inv = inv_q(q_coarse, B)
    mask = (~np.all(xyz == 0, axis=-1)).astype(F)[..., None]
    xyz_q = np.concatenate([np.zeros(xyz.shape[:2] + (1,), F), xyz], -1)
    flow = mul_q_point(q_coarse, xyz_q, B)
    return ((mul_point_q(flow, q_inv, B)[:, :, 1:] + t_coarse) * mask).astype(F)


def compose(q_det, t_det, q_coarse, t_coarse):
    """pwclo_model.py:275-280: q = q_det (x) q_coarse ; t = (q_det (x) [0,t_coarse] (x) q_det^-1)[1:] + t_det.
    All of shape (B,1,*); returns q (B,4), t (B,3)."""
    B = q_det.shape[0]
    tq = np.concatenate([np.zeros((B, 1, 1), F), t_coarse], -1)
    tq = mul_q_point(q_det, tq, B)
    tq = mul_point_q(tq, inv_q(q_det, B), B)[:, :, 1:]
    q = mul_point_q(q_det, q_coarse, B)[:, 0, :]
    t = (tq + t_det)[:, 0, :]
    return q.astype(F), t.astype(F)


# --------------------------------------------------------------------------- spherical projection
def projection_constants(H_input, W_input):
    """model_util.py:189-200: python doubles, then tf.constant -> float32."""
    d2r = math.pi / 180
    az = (360.0 / W_input) * d2r
    down, up = -24.8 * d2r, 2.0 * d2r
    vres = (up - down) / (H_input - 1)
    voff = -down / vres
    return F(az), F(vres), F(voff)


def projection_coordinates64(P, H_input, W_input):
    """The CONTINUOUS coordinates behind ProjectPC2SphericalRing's two truncations, in float64 from the given points (N,3):
    (col, tmp) with iCol = int(col), iRow = H - int(tmp) (model_util.py:234-242), using the float32 constants both
    implementations use.  For the parity statistic: a point whose float32 cell differs between two correct implementations
    must sit within a few float32 ulps (of the index) of an integer value of one of them."""
    az, vres, voff = projection_constants(H_input, W_input)
    P = np.asarray(P, np.float64)
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    r = np.sqrt(x * x + y * y + z * z)
    with np.errstate(all="ignore"):
        col = (np.float64(F(np.pi)) - np.arctan2(y, x)) / np.float64(az)
        tmp = np.arcsin(z / r) / np.float64(vres) + np.float64(voff)
    return col, tmp


def scatter_min_range(cell, r, values, ncell, out_shape):
    """model_util.py:255-273: per-cell minimum range (tf.unique + unsorted_segment_min + gather),
    mask_same = (r == min_r), then tf.scatter_nd of the masked rows -- duplicates are ADDED.
    Returns (min_r gathered back per point, scattered grid)."""
    min_r = np.full(ncell, np.inf, F)
    np.minimum.at(min_r, cell, r)
    min_r_pt = min_r[cell]
    same = (r == min_r_pt).astype(F)[:, None]
    out = np.zeros((ncell, values.shape[-1]), F)
    np.add.at(out, cell, values * same)
    return min_r_pt, out.reshape(out_shape)


def ProjectPC2SphericalRing(PC, Feature, H_input, W_input):
    """model_util.py:181-292.  NaN -> int uses the GPU convention (NaN -> 0), SURVEY.md a-10."""
    B, N, _ = PC.shape
    az, vres, voff = projection_constants(H_input, W_input)
    PI = F(np.pi)
    out_xyz = np.zeros((B, H_input, W_input, 3), F)
    out_feat = None if Feature is None else np.zeros((B, H_input, W_input, Feature.shape[-1]), F)
    for b in range(B):
        cur = PC[b, :, :3].astype(F)
        x, y, z = cur[:, 0], cur[:, 1], cur[:, 2]
        r = np.sqrt((cur * cur).sum(1)).astype(F)                                                 # :229
        with np.errstate(all="ignore"):
            iCol = ((PI - np.arctan2(y, x).astype(F)) / az).astype(F)                             # :234
            beta = np.arcsin((z / r).astype(F)).astype(F)                                         # :237
            tmp = (beta / vres + voff).astype(F)                                                  # :239
        to_i32 = lambda v: np.nan_to_num(np.trunc(v), nan=0.0, posinf=2**31 - 1, neginf=-2**31).astype(np.int64)
        iCol = to_i32(iCol)
        iRow = H_input - to_i32(tmp)                                                              # :242
        iRow = np.clip(iRow, 0, H_input - 1)
        iCol = np.clip(iCol, 0, W_input - 1)
        cell = iRow * W_input + iCol
        min_r_pt, out_xyz[b] = scatter_min_range(cell, r, cur, H_input * W_input, (H_input, W_input, 3))
        _rec("cell", np.stack([cell, (r == min_r_pt).astype(np.int64), (r > 0).astype(np.int64)]))
        _rec("warped", np.concatenate([cur, r[:, None]], 1))       # (continuous: the points and ranges the decisions above were taken on)
        if Feature is not None:
            _, out_feat[b] = scatter_min_range(cell, r, Feature[b].astype(F), H_input * W_input,
                                               (H_input, W_input, Feature.shape[-1]))
    _rec("grid", out_xyz)                                    # (continuous: compared with the product's grid by the parity statistic, not counted as decisions)
    return out_xyz, (_st(out_feat) if Feature is not None else out_xyz)


def softmax_valid(feature_bnc, weight_bnc, mask_valid):
    """model_util.py:319-343 -> (B,1,C)."""
    out = []
    for b in range(feature_bnc.shape[0]):
        f, w = feature_bnc[b][mask_valid[b]], weight_bnc[b][mask_valid[b]]
        if f.shape[0] == 0:
            out.append(np.zeros((1, 1, feature_bnc.shape[-1]), F))
            continue
        out.append((f * softmax(w, 0)).sum(0, keepdims=True)[None])
    return np.concatenate(out, 0).astype(F)


# --------------------------------------------------------------------------- pre-process, pose utilities, loss
def mat2euler(M):
    """model_util.py:130-142."""
    cy = math.sqrt(M[2, 2] * M[2, 2] + M[1, 2] * M[1, 2])
    return math.atan2(-M[0, 1], M[0, 0]), math.atan2(M[0, 2], cy), math.atan2(-M[1, 2], M[2, 2])


def euler2quat(z, y, x):
    """model_util.py:112-127."""
    z, y, x = z / 2.0, y / 2.0, x / 2.0
    cz, sz, cy, sy, cx, sx = math.cos(z), math.sin(z), math.cos(y), math.sin(y), math.cos(x), math.sin(x)
    return np.array([cx * cy * cz - sx * sy * sz, cx * sy * sz + cy * cz * sx,
                     cx * cz * sy - sx * cy * sz, cx * cy * sz + sx * cz * sy], F)


def PreProcess(PC_f1, PC_f2, T_gt, T_trans, T_trans_inv, aug_frame):
    """model_util.py:346-445."""
    B = PC_f1.shape[0]
    outs1, outs2, qs, ts = [], [], [], []
    for i in range(B):
        m1 = (~np.all(PC_f1[i] == 0, -1)).astype(F)[:, None]
        m2 = (~np.all(PC_f2[i] == 0, -1)).astype(F)[:, None]
        p1 = np.concatenate([PC_f1[i], np.ones((PC_f1.shape[1], 1), F)], -1).astype(F)
        p2 = np.concatenate([PC_f2[i], np.ones((PC_f2.shape[1], 1), F)], -1).astype(F)
        p1[np.sqrt((p1[:, :2] ** 2).sum(1)) > 35] = 0                                             # :380-383
        p2[np.sqrt((p2[:, :2] ** 2).sum(1)) > 35] = 0
        Tg = T_gt[i].astype(F)
        if aug_frame[i] == 2:                                                                     # :390-403
            p2 = (T_trans[i].astype(F) @ p2.T).T
            Tg = T_trans[i].astype(F) @ Tg
        elif aug_frame[i] == 1:                                                                   # :406-419
            p1 = (T_trans[i].astype(F) @ p1.T).T
            Tg = Tg @ T_trans_inv[i].astype(F)
        outs1.append(p1[:, :3] * m1)
        outs2.append(p2[:, :3] * m2)
        qs.append(euler2quat(*mat2euler(Tg[:3, :3])))
        ts.append(Tg[:3, 3:])
    return (np.stack(outs1).astype(F), np.stack(outs2).astype(F), np.stack(qs).astype(F), np.stack(ts).astype(F))


def normalise_q(q):
    """pwclo_model.py:203 and :427-430."""
    return (q / (np.sqrt((q * q).sum(-1, keepdims=True) + F(1e-10)) + F(1e-10))).astype(F)


def get_loss(l0_q, l0_t, l1_q, l1_t, l2_q, l2_t, l3_q, l3_t, q_gt, t_gt, w_x, w_q):
    """pwclo_model.py:437-481."""
    t_gt = t_gt[..., 0]
    def level(q, t):
        qn = normalise_q(q)
        lq = np.sqrt(((q_gt - qn) ** 2).sum(-1, keepdims=True) + 1e-10).mean()
        lx = np.sqrt((t - t_gt) ** 2 + 1e-10).mean()
        return lx * math.exp(-w_x) + w_x + lq * math.exp(-w_q) + w_q
    return 1.6 * level(l3_q, l3_t) + 0.8 * level(l2_q, l2_t) + 0.4 * level(l1_q, l1_t) + 0.2 * level(l0_q, l0_t)


# --------------------------------------------------------------------------- the pyramid schedule
def pose_head(params, feat_b1c, level, coarse):
    """pwclo_model.py:197-208 (and :264-273 ...): conv1d 64->256 (no activation, no BN), q/t heads."""
    big = conv(params, "l%d_big" % level, feat_b1c, bn=False, relu=False)
    qn, tn = ("l%d_q_coarse", "l%d_t_coarse") if coarse else ("l%d_q_det", "l%d_t_det")
    q = normalise_q(conv(params, qn % level, big, bn=False, relu=False))
    t = conv(params, tn % level, big, bn=False, relu=False)
    return q, t


def get_model_from_projection(params, shuffle, xyz_f1_proj, xyz_f2_proj, coarse_pose=None):
    """pwclo_model.py:69-433 (inference: dropout off, BN moving stats), starting from the two
    (B,H,W,3) range images that PreProcess + ProjectPC2SphericalRing produce (:61-67).

    `coarse_pose` (a test device, not in the reference): {level: (q (B,4), t (B,3))} -- the pose the NEXT finer level
    warps by and composes with is taken from this table instead of from the oracle's own level `level` ("teacher
    forcing" with the checked implementation's poses).  Each level's own output is still what the oracle computes, so
    every level is compared on the same coarse pose and a discrete decision (a point changing its projection cell, a
    neighbour entering a window) taken differently at a coarse level does not compound into the finer ones."""
    B, H_input, W_input, _ = xyz_f1_proj.shape
    Down_conv_dis, Up_conv_dis, Cost_volume_dis = [0.5, 3.0, 6.0, 12.0], [3.0, 6.0, 9.0], [1.0, 2.0, 4.0]   # :38-40
    sh, sw = [1, 1, 4, 2, 2, 1], [1, 1, 8, 2, 2, 2]                                                        # :42-43
    oh, ow = [math.ceil(H_input / sh[0])], [math.ceil(W_input / sw[0])]
    for i in range(1, 6):
        oh.append(math.ceil(oh[i - 1] / sh[i])); ow.append(math.ceil(ow[i - 1] / sw[i]))                   # :45-50
    pts1 = np.zeros((B, H_input, W_input, 3), F); pts2 = pts1                                              # :69-70
    pre2 = get_selected_idx(B, sh[1], sw[1], oh[1], ow[1])
    pre2_f1, pre2_f2 = gather_nd(xyz_f1_proj, pre2), gather_nd(xyz_f2_proj, pre2)                          # :88-90
    l0_sel = get_selected_idx(B, sh[2], sw[2], oh[2], ow[2])
    l0_x1, l0_x2 = gather_nd(pre2_f1, l0_sel), gather_nd(pre2_f2, l0_sel)                                  # :95-97
    l1_sel = get_selected_idx(B, sh[3], sw[3], oh[3], ow[3])
    l1_x1, l1_x2 = gather_nd(l0_x1, l1_sel), gather_nd(l0_x2, l1_sel)                                      # :102-104
    l2_sel = get_selected_idx(B, sh[4], sw[4], oh[4], ow[4])
    l2_x1 = gather_nd(l1_x1, l2_sel)                                                                       # :108-110
    l3_sel = get_selected_idx(B, sh[5], sw[5], oh[5], ow[5])                                               # :114

    def pyramid(xyz_in, pts_in):                                                                           # :126-139 / :151-164
        l0_p, l0_x = down_conv(params, shuffle, xyz_in, pts_in, l0_sel, 32, [9, 15], Down_conv_dis[0], [8, 8, 16], "sa1/layer0")
        l0_pp = l0_p.reshape(B, oh[2], ow[2], -1)
        l1_p, l1_x = down_conv(params, shuffle, l0_x, l0_pp, l1_sel, 32, [7, 11], Down_conv_dis[1], [16, 16, 32], "sa1/layer1")
        l1_pp = l1_p.reshape(B, oh[3], ow[3], -1)
        l2_p, l2_x = down_conv(params, shuffle, l1_x, l1_pp, l2_sel, 16, [5, 9], Down_conv_dis[2], [32, 32, 64], "sa1/layer2")
        l2_pp = l2_p.reshape(B, oh[4], ow[4], -1)
        l3_p, l3_x = down_conv(params, shuffle, l2_x, l2_pp, l3_sel, 16, [5, 9], Down_conv_dis[3], [64, 64, 128], "sa1/layer3")
        return dict(p=[l0_p, l1_p, l2_p, l3_p], pp=[l0_pp, l1_pp, l2_pp, l3_p.reshape(B, oh[5], ow[5], -1)],
                    x=[l0_x, l1_x, l2_x, l3_x])
    f1, f2 = pyramid(xyz_f1_proj, pts1), pyramid(xyz_f2_proj, pts2)

    l2_new = cost_volume(params, shuffle, f1["x"][2], f2["x"][2], f1["pp"][2], f2["pp"][2], [3, 5], [5, 35], 4, 32,
                         Cost_volume_dis[2], [128, 64, 64], [128, 64], "flow_embedding_l2_origin")        # :170
    l2_new_proj = l2_new.reshape(B, oh[4], ow[4], -1)
    l3_cv, _ = down_conv(params, shuffle, f1["x"][2], l2_new_proj, l3_sel, 16, [5, 9], Down_conv_dis[3], [128, 64, 64], "new_layer3")  # :177
    l3_pred = l3_cv
    l3_pred_proj = l3_pred.reshape(B, oh[5], ow[5], -1)
    l3_w = flow_predictor(params, f1["p"][3], None, l3_pred, [128, 64], "l3_costvolume_predict_ww")       # :187
    l3_w_proj = l3_w.reshape(B, oh[5], ow[5], -1)
    l3_xyz = f1["x"][3].reshape(B, -1, 3)
    l3_valid = ~np.all(l3_xyz == 0, -1)
    l3_feat = softmax_valid(l3_pred, l3_w, l3_valid)                                                       # :194
    q, t = pose_head(params, l3_feat, 3, coarse=True)                                                      # :197-208
    l3_q, l3_t = q[:, 0, :], t[:, 0, :]
    outs = {3: (l3_q, l3_t)}
    if coarse_pose and 3 in coarse_pose:
        l3_q, l3_t = (np.asarray(v, F) for v in coarse_pose[3])

    prev_w_proj, prev_pred_proj, prev_xyz_proj = l3_w_proj, l3_pred_proj, f1["x"][3]
    cv_kernels = {2: [5, 15], 1: [7, 25], 0: [11, 41]}
    q_prev, t_prev = l3_q, l3_t
    for lvl, gi in ((2, 4), (1, 3), (0, 2)):
        if _TRACE[-1] is not None:
            _TRACE[-1].level = lvl
        q_coarse, t_coarse = q_prev.reshape(B, 1, -1), t_prev.reshape(B, 1, -1)                            # :211-212
        xyz = f1["x"][lvl].reshape(B, -1, 3)
        warped = warp(xyz, q_coarse, t_coarse)                                                             # :217-227
        w_xyz_proj, w_pts_proj = ProjectPC2SphericalRing(warped, f1["p"][lvl], oh[gi], ow[gi])             # :232 (sizes generalised)
        w_xyz = w_xyz_proj.reshape(B, -1, 3)
        w_pts = w_pts_proj.reshape(B, oh[gi] * ow[gi], -1)
        valid_warp = ~np.all(w_xyz == 0, -1)
        cv = cost_volume(params, shuffle, w_xyz_proj, f2["x"][lvl], w_pts_proj, f2["pp"][lvl], [3, 5], cv_kernels[lvl],
                         4, 6, Cost_volume_dis[lvl], [128, 64, 64], [128, 64], "flow_embedding_l%d" % lvl)    # :242
        w_up = up_conv(params, shuffle, w_xyz_proj, prev_xyz_proj, w_pts_proj, prev_w_proj, [7, 15], sh[gi + 1], sw[gi + 1], 8,
                       Up_conv_dis[lvl], [128, 64], [128, 64], "up_sa_layer_layer_l%dw" % lvl)                # :247
        cv_up = up_conv(params, shuffle, w_xyz_proj, prev_xyz_proj, w_pts_proj, prev_pred_proj, [7, 15], sh[gi + 1], sw[gi + 1], 8,
                        Up_conv_dis[lvl], [128, 64], [128, 64], "up_sa_layer_layer_l%dcostvolume" % lvl)      # :250
        pred = flow_predictor(params, w_pts, cv_up, cv, [128, 64], "l%d_costvolume_predict" % lvl)         # :253
        wgt = flow_predictor(params, w_pts, w_up, cv, [128, 64], "l%d_w_predict" % lvl)                    # :254
        feat = softmax_valid(pred, wgt, valid_warp)                                                        # :262
        q_det, t_det = pose_head(params, feat, lvl, coarse=False)                                          # :264-273
        q_prev, t_prev = compose(q_det, t_det, q_coarse, t_coarse)                                         # :275-280
        outs[lvl] = (q_prev, t_prev)
        if coarse_pose and lvl in coarse_pose:
            q_prev, t_prev = (np.asarray(v, F) for v in coarse_pose[lvl])
        prev_w_proj = wgt.reshape(B, oh[gi], ow[gi], -1)
        prev_pred_proj = pred.reshape(B, oh[gi], ow[gi], -1)
        prev_xyz_proj = w_xyz_proj
    l0_xyz_f1 = f1["x"][0].reshape(B, -1, 3)
    return (normalise_q(outs[0][0]), outs[0][1], normalise_q(outs[1][0]), outs[1][1],
            normalise_q(outs[2][0]), outs[2][1], normalise_q(outs[3][0]), outs[3][1], l0_xyz_f1)
