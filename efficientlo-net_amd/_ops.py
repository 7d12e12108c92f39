"""Thin torch-tensor front-ends of the feature kernels (include/elo.h, csrc/elo_features.hip) and of their hand-written
backward passes (csrc/elo_backward.hip).  Every public operator here is ONE implementation: the HIP kernel.  When an
input requires grad (training) the call goes through a torch.autograd.Function whose forward is that same kernel and
whose backward is the matching `elo_*_backward` kernel -- there is no second (torch) implementation to fall onto and
nothing here looks at the global autograd mode.  Shapes are validated in C as well.  No CPU fallback: CPU tensors raise."""
import math

import numpy as np

import ctypes

import torch

from . import _lib as L
from . import tuning


def _wants_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


_ptr = lambda x: x.data_ptr() if x is not None else None


def _feature_dtype(*ts):
    """Feature tensors are stored as fp32 or fp16 (fp32 arithmetic either way): all feature tensors of a call share one
    of the two; returns (contiguous tensors, torch dtype, ELO_F32 / ELO_F16)."""
    dt = ts[0].dtype
    if dt not in (torch.float32, torch.float16) or any(t.dtype != dt for t in ts):
        raise TypeError("cost-volume feature tensors are all float32 or all float16 (got %s)" % [str(t.dtype) for t in ts])
    return [t.contiguous() for t in ts], dt, L.ELO_F16 if dt == torch.float16 else L.ELO_F32


def _f32(*ts):
    out = []
    for t in ts:
        if t.dtype != torch.float32:
            raise TypeError("feature-path tensors are float32 (got %s)" % t.dtype)
        out.append(t.contiguous())
    return out


class _GroupConcat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, centre_xyz, src_xyz, src_feat, idx, mask):
        ctx.save_for_backward(idx, mask)
        ctx.shapes = (centre_xyz.shape, src_xyz.shape, src_feat.shape)
        return _group_concat(centre_xyz, src_xyz, src_feat, idx, mask)

    @staticmethod
    def backward(ctx, grad):
        idx, mask = ctx.saved_tensors
        (cs, xs, fs), dev = ctx.shapes, grad.device
        (grad,) = _f32(grad)
        need = ctx.needs_input_grad
        g_c = torch.empty(cs, dtype=torch.float32, device=dev) if need[0] else None
        g_x = torch.zeros(xs, dtype=torch.float32, device=dev) if need[1] else None
        g_f = torch.zeros(fs, dtype=torch.float32, device=dev) if need[2] else None
        B, N, K, _ = idx.shape
        a = L.GroupConcatBwdArgs(B, N, K, fs[1], fs[2], fs[3], grad.data_ptr(), idx.data_ptr(), mask.data_ptr(),
                                 _ptr(g_c), _ptr(g_x), _ptr(g_f))
        L.call("elo_group_concat_backward", a, grad)
        return g_c, g_x, g_f, None, None


def group_concat(centre_xyz, src_xyz, src_feat, idx, mask):
    """[src_xyz[idx]*m - centre, src_feat[idx]*m] -> (B,N,K,3+C).  pointnet_util.py:203-213, :277-284."""
    if _wants_grad(centre_xyz, src_xyz, src_feat):
        return _GroupConcat.apply(centre_xyz, src_xyz, src_feat, idx.contiguous(), _f32(mask)[0])
    return _group_concat(centre_xyz, src_xyz, src_feat, idx, mask)


def _group_concat(centre_xyz, src_xyz, src_feat, idx, mask):
    L.require_gpu(centre_xyz, src_xyz, src_feat, idx, mask)
    centre_xyz, src_xyz, src_feat, mask = _f32(centre_xyz, src_xyz, src_feat, mask)
    idx = idx.contiguous()
    B, N, K, _ = idx.shape
    _, H2, W2, C = src_feat.shape
    out = torch.empty((B, N, K, 3 + C), dtype=torch.float32, device=idx.device)
    a = L.GroupConcatArgs(B, N, K, H2, W2, C, centre_xyz.data_ptr(), src_xyz.data_ptr(), src_feat.data_ptr(),
                          idx.data_ptr(), mask.data_ptr(), out.data_ptr())
    L.call("elo_group_concat", a, out)
    return out


class _MaskedMaxpool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(x, mask)
        return _masked_maxpool(x, mask)

    @staticmethod
    def backward(ctx, grad):
        x, mask = ctx.saved_tensors
        (grad,) = _f32(grad)
        B, N, K, C = x.shape
        g_x = torch.empty_like(x)
        L.call("elo_masked_maxpool_backward",
               L.MaskedMaxpoolBwdArgs(B, N, K, C, x.data_ptr(), mask.data_ptr(), grad.data_ptr(), g_x.data_ptr()), grad)
        return g_x, None


def masked_maxpool(x, mask):
    """max_k x*mask -> (B,N,C).  pointnet_util.py:224-230, :295-298."""
    if _wants_grad(x):
        return _MaskedMaxpool.apply(*_f32(x, mask))
    return _masked_maxpool(x, mask)


def _masked_maxpool(x, mask):
    L.require_gpu(x, mask)
    x, mask = _f32(x, mask)
    B, N, K, C = x.shape
    out = torch.empty((B, N, C), dtype=torch.float32, device=x.device)
    L.call("elo_masked_maxpool", L.MaskedMaxpoolArgs(B, N, K, C, x.data_ptr(), mask.data_ptr(), out.data_ptr()), out)
    return out


class _CvEncode1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, feat1, xyz2_proj, feat2_proj, idx, mask):
        ctx.save_for_backward(xyz1, xyz2_proj, idx, mask)
        ctx.C = feat1.shape[-1]
        return _cv_encode1(xyz1, feat1, xyz2_proj, feat2_proj, idx, mask)

    @staticmethod
    def backward(ctx, grad):
        xyz1, xyz2, idx, mask = ctx.saved_tensors
        (grad,) = _f32(grad)
        B, N, K, _ = idx.shape
        _, H2, W2, _ = xyz2.shape
        C, dev, need = ctx.C, grad.device, ctx.needs_input_grad
        new = lambda shape, zero: (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=dev)
        g_x1 = new((B, N, 3), False) if need[0] else None
        g_f1 = new((B, N, C), False) if need[1] else None
        g_x2 = new((B, H2, W2, 3), True) if need[2] else None
        g_f2 = new((B, H2, W2, C), True) if need[3] else None
        a = L.CvEncode1BwdArgs(B, N, K, H2, W2, C, xyz1.data_ptr(), xyz2.data_ptr(), idx.data_ptr(), mask.data_ptr(),
                               grad.data_ptr(), _ptr(g_x1), _ptr(g_f1), _ptr(g_x2), _ptr(g_f2))
        L.call("elo_cv_encode1_backward", a, grad)
        return g_x1, g_f1, g_x2, g_f2, None, None


def cv_encode1(xyz1, feat1, xyz2_proj, feat2_proj, idx, mask):
    """(B,N,K,10+2C) = [p, q, q-p, |q-p|, feat1, feat2[idx]*m].  pointnet_util.py:54-66."""
    if _wants_grad(xyz1, feat1, xyz2_proj, feat2_proj):
        return _CvEncode1.apply(*_f32(xyz1, feat1, xyz2_proj, feat2_proj), idx.contiguous(), _f32(mask)[0])
    return _cv_encode1(xyz1, feat1, xyz2_proj, feat2_proj, idx, mask)


def _cv_encode1(xyz1, feat1, xyz2_proj, feat2_proj, idx, mask):
    L.require_gpu(xyz1, feat1, xyz2_proj, feat2_proj, idx, mask)
    xyz1, xyz2_proj, mask = _f32(xyz1, xyz2_proj, mask)
    (feat1, feat2_proj), dt, code = _feature_dtype(feat1, feat2_proj)
    idx = idx.contiguous()
    B, N, K, _ = idx.shape
    _, H2, W2, C = feat2_proj.shape
    out = torch.empty((B, N, K, 10 + 2 * C), dtype=dt, device=idx.device)
    a = L.CvEncode1Args(B, N, K, H2, W2, C, xyz1.data_ptr(), feat1.data_ptr(), xyz2_proj.data_ptr(),
                        feat2_proj.data_ptr(), idx.data_ptr(), mask.data_ptr(), out.data_ptr(), code)
    L.call("elo_cv_encode1", a, out)
    return out


class _CvEncode2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1_proj, feat1_proj, cost_proj, idx, mask):
        ctx.save_for_backward(xyz1_proj, idx, mask)
        ctx.widths = (feat1_proj.shape[-1], cost_proj.shape[-1])
        return _cv_encode2(xyz1_proj, feat1_proj, cost_proj, idx, mask)

    @staticmethod
    def backward(ctx, g_cat, g_rest):
        xyz1, idx, mask = ctx.saved_tensors
        B, H, W, _ = xyz1.shape
        K, (C, Cc), dev, need = idx.shape[2], ctx.widths, xyz1.device, ctx.needs_input_grad
        zeros = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        g_cat = zeros(B, H * W, K, 10) if g_cat is None else _f32(g_cat)[0]
        g_rest = zeros(B, H * W, K, C + Cc) if g_rest is None else _f32(g_rest)[0]
        g_x = zeros(B, H, W, 3) if need[0] else None
        g_f = torch.empty((B, H, W, C), dtype=torch.float32, device=dev) if need[1] else None
        g_c = zeros(B, H, W, Cc) if need[2] else None
        a = L.CvEncode2BwdArgs(B, H * W, K, H, W, C, Cc, xyz1.data_ptr(), idx.data_ptr(), mask.data_ptr(), g_cat.data_ptr(),
                               g_rest.data_ptr(), _ptr(g_x), _ptr(g_f), _ptr(g_c))
        L.call("elo_cv_encode2_backward", a, xyz1)
        return g_x, g_f, g_c, None, None


def cv_encode2(xyz1_proj, feat1_proj, cost_proj, idx, mask):
    """xyz_cat (B,N,K,10) and rest (B,N,K,C+Cc) = [feat1, cost[idx]*m].  pointnet_util.py:110-129."""
    if _wants_grad(xyz1_proj, feat1_proj, cost_proj):
        return _CvEncode2.apply(*_f32(xyz1_proj, feat1_proj, cost_proj), idx.contiguous(), _f32(mask)[0])
    return _cv_encode2(xyz1_proj, feat1_proj, cost_proj, idx, mask)


def _cv_encode2(xyz1_proj, feat1_proj, cost_proj, idx, mask):
    L.require_gpu(xyz1_proj, feat1_proj, cost_proj, idx, mask)
    xyz1_proj, mask = _f32(xyz1_proj, mask)
    (feat1_proj, cost_proj), dt, code = _feature_dtype(feat1_proj, cost_proj)
    idx = idx.contiguous()
    B, N, K, _ = idx.shape
    _, H, W, C = feat1_proj.shape
    Cc = cost_proj.shape[-1]
    xyz_cat = torch.empty((B, N, K, 10), dtype=dt, device=idx.device)
    rest = torch.empty((B, N, K, C + Cc), dtype=dt, device=idx.device)
    a = L.CvEncode2Args(B, N, K, H, W, C, Cc, xyz1_proj.data_ptr(), feat1_proj.data_ptr(), cost_proj.data_ptr(),
                        idx.data_ptr(), mask.data_ptr(), xyz_cat.data_ptr(), rest.data_ptr(), code)
    L.call("elo_cv_encode2", a, rest)
    return xyz_cat, rest


class _MaskedSoftmaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, values, mask):
        ctx.save_for_backward(logits, values, mask)
        return _masked_softmax_pool(logits, values, mask)

    @staticmethod
    def backward(ctx, grad):
        logits, values, mask = ctx.saved_tensors
        (grad,) = _f32(grad)
        B, N, K, C = logits.shape
        if values.stride(-1) != 1 or values.stride(1) != K * values.stride(2) or values.stride(0) != N * values.stride(1):
            values = values.contiguous()
        g_l, g_v = torch.empty_like(logits), torch.empty((B, N, K, C), dtype=torch.float32, device=grad.device)
        a = L.SoftmaxPoolBwdArgs(B, N, K, C, logits.data_ptr(), values.data_ptr(), values.stride(2), mask.data_ptr(),
                                 grad.data_ptr(), g_l.data_ptr(), g_v.data_ptr())
        L.call("elo_masked_softmax_pool_backward", a, grad)
        return g_l, g_v, None


def masked_softmax_pool(logits, values, mask):
    """sum_k softmax_k(where(mask==1, logits, -1e10)) * values -> (B,N,C).  pointnet_util.py:92-98, :137-146.
    `values` may be a last-dim slice of a wider contiguous tensor (no copy)."""
    if _wants_grad(logits, values):
        if logits.dtype != torch.float32 or values.dtype != torch.float32:
            raise TypeError("training stores its features in float32")
        return _MaskedSoftmaxPool.apply(logits.contiguous(), values, _f32(mask)[0])
    return _masked_softmax_pool(logits, values, mask)


def _masked_softmax_pool(logits, values, mask):
    L.require_gpu(logits, values, mask)
    (mask,) = _f32(mask)
    if logits.dtype not in (torch.float32, torch.float16):
        raise TypeError("logits are float32 or float16 (got %s)" % logits.dtype)
    logits = logits.contiguous()
    dt, code = logits.dtype, L.ELO_F16 if logits.dtype == torch.float16 else L.ELO_F32
    B, N, K, C = logits.shape
    if values.dtype != dt or values.shape != logits.shape:
        raise ValueError("values must have the dtype and shape of logits")
    if values.stride(-1) != 1 or values.stride(1) != K * values.stride(2) or values.stride(0) != N * values.stride(1):
        values = values.contiguous()
    out = torch.empty((B, N, C), dtype=dt, device=logits.device)
    a = L.SoftmaxPoolArgs(B, N, K, C, logits.data_ptr(), values.data_ptr(), values.stride(2), mask.data_ptr(),
                          out.data_ptr(), code)
    L.call("elo_masked_softmax_pool", a, out)
    return out


class _SoftmaxValid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feature_bnc, weight_bnc, xyz_bn3):
        B, N, C = feature_bnc.shape
        stats = torch.empty((B, 2, C), dtype=torch.float32, device=feature_bnc.device)
        out = _softmax_valid(feature_bnc, weight_bnc, xyz_bn3, stats)
        ctx.save_for_backward(feature_bnc, weight_bnc, xyz_bn3, out, stats)
        return out

    @staticmethod
    def backward(ctx, grad):
        f, w, xyz, out, stats = ctx.saved_tensors
        (grad,) = _f32(grad)
        B, N, C = f.shape
        g_f, g_w = torch.empty_like(f), torch.empty_like(w)
        a = L.SoftmaxValidBwdArgs(B, N, C, f.data_ptr(), w.data_ptr(), xyz.data_ptr(), grad.data_ptr(), g_f.data_ptr(),
                                  g_w.data_ptr(), out.data_ptr(), stats.data_ptr())
        L.call("elo_softmax_valid_backward", a, grad)
        return g_f, g_w, None


def softmax_valid(feature_bnc, weight_bnc, xyz_bn3):
    """model_util.py:319-343 with mask_valid = any(xyz != 0) -> (B,1,C)."""
    if _wants_grad(feature_bnc, weight_bnc):
        return _SoftmaxValid.apply(*_f32(feature_bnc, weight_bnc, xyz_bn3.detach()))
    return _softmax_valid(feature_bnc, weight_bnc, xyz_bn3)


def _softmax_valid(feature_bnc, weight_bnc, xyz_bn3, stats=None):
    L.require_gpu(feature_bnc, weight_bnc, xyz_bn3)
    feature_bnc, weight_bnc, xyz_bn3 = _f32(feature_bnc, weight_bnc, xyz_bn3)
    B, N, C = feature_bnc.shape
    out = torch.empty((B, 1, C), dtype=torch.float32, device=feature_bnc.device)
    scratch = torch.empty((3 * B * L.SV_MAX_PARTS * C,), dtype=torch.float32, device=feature_bnc.device)
    a = L.SoftmaxValidArgs(B, N, C, feature_bnc.data_ptr(), weight_bnc.data_ptr(), xyz_bn3.data_ptr(), out.data_ptr(),
                           scratch.data_ptr(), stats.data_ptr() if stats is not None else None)
    L.call("elo_softmax_valid", a, out)
    return out


class _PoseCompose(torch.autograd.Function):
    """normalise -> compose with the coarse pose -> normalise (pwclo_model.py:197-208, :262-280) as ONE launch forward and ONE
    backward (elo_pose_compose); torch ran the same chain as ~60 + ~120 kernels of eight elements per level."""

    @staticmethod
    def forward(ctx, q_raw, t_det, q_coarse, t_coarse):
        B = q_raw.shape[0]
        q, t, q_norm = torch.empty_like(q_raw), torch.empty_like(t_det), torch.empty_like(q_raw)
        ptr = lambda x: x.data_ptr() if x is not None else None
        a = L.PoseComposeArgs(B, q_raw.data_ptr(), t_det.data_ptr(), ptr(q_coarse), ptr(t_coarse), q.data_ptr(), t.data_ptr(),
                              q_norm.data_ptr(), None, None, None, None, None, None, None)
        L.call("elo_pose_compose", a, q_raw)
        ctx.save_for_backward(q_raw, t_det, q_coarse, t_coarse)
        return q, t, q_norm

    @staticmethod
    def backward(ctx, gq, gt, gqn):
        q_raw, t_det, q_coarse, t_coarse = ctx.saved_tensors
        B = q_raw.shape[0]
        z = lambda g, like: torch.zeros_like(like) if g is None else g.contiguous()
        gq, gt, gqn = z(gq, q_raw), z(gt, t_det), z(gqn, q_raw)
        g_qr, g_td = torch.empty_like(q_raw), torch.empty_like(t_det)
        g_qc = torch.empty_like(q_coarse) if q_coarse is not None else None
        g_tc = torch.empty_like(t_coarse) if t_coarse is not None else None
        ptr = lambda x: x.data_ptr() if x is not None else None
        a = L.PoseComposeArgs(B, q_raw.data_ptr(), t_det.data_ptr(), ptr(q_coarse), ptr(t_coarse), None, None, None,
                              gq.data_ptr(), gt.data_ptr(), gqn.data_ptr(), g_qr.data_ptr(), g_td.data_ptr(), ptr(g_qc), ptr(g_tc))
        L.call("elo_pose_compose", a, q_raw)
        return g_qr, g_td, g_qc, g_tc


def pose_compose(q_raw, t_det, q_coarse=None, t_coarse=None):
    """(q (B,4), t (B,3), q_norm (B,4)) of a level from the head's raw quaternion and translation and the coarse pose
    (None at the coarsest level); differentiable (the training path's pose algebra in one launch each way)."""
    L.require_gpu(q_raw, t_det, q_coarse, t_coarse)
    B = q_raw.shape[0]
    f = lambda x, n: None if x is None else _f32(x.reshape(B, n))[0]
    return _PoseCompose.apply(f(q_raw, 4), f(t_det, 3), f(q_coarse, 4), f(t_coarse, 3))


class _PoseLoss(torch.autograd.Function):
    """get_loss (pwclo_model.py:437-481) over the four levels: one launch forward, one backward (elo_pose_loss)."""

    @staticmethod
    def forward(ctx, w_x, w_q, q_gt, t_gt, *poses):                # poses: l0_q, l0_t, l1_q, l1_t, l2_q, l2_t, l3_q, l3_t
        B = q_gt.shape[0]
        loss = torch.empty((), dtype=torch.float32, device=q_gt.device)
        P4 = ctypes.c_void_p * 4
        a = L.PoseLossArgs(B, P4(*[poses[2 * i].data_ptr() for i in range(4)]), P4(*[poses[2 * i + 1].data_ptr() for i in range(4)]),
                           q_gt.data_ptr(), t_gt.data_ptr(), w_x.data_ptr(), w_q.data_ptr(), loss.data_ptr(), None, P4(), P4(), None, None)
        L.call("elo_pose_loss", a, q_gt)
        ctx.save_for_backward(w_x, w_q, q_gt, t_gt, *poses)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        w_x, w_q, q_gt, t_gt, *poses = ctx.saved_tensors
        B = q_gt.shape[0]
        grads = [torch.empty_like(x) for x in poses]
        g_wx, g_wq = torch.empty_like(w_x), torch.empty_like(w_q)
        go = grad_out.contiguous().float()
        P4 = ctypes.c_void_p * 4
        a = L.PoseLossArgs(B, P4(*[poses[2 * i].data_ptr() for i in range(4)]), P4(*[poses[2 * i + 1].data_ptr() for i in range(4)]),
                           q_gt.data_ptr(), t_gt.data_ptr(), w_x.data_ptr(), w_q.data_ptr(), None, go.data_ptr(),
                           P4(*[grads[2 * i].data_ptr() for i in range(4)]), P4(*[grads[2 * i + 1].data_ptr() for i in range(4)]),
                           g_wx.data_ptr(), g_wq.data_ptr())
        L.call("elo_pose_loss", a, q_gt)
        return (g_wx, g_wq, None, None) + tuple(grads)


def pose_loss(l0_q, l0_t, l1_q, l1_t, l2_q, l2_t, l3_q, l3_t, q_gt, t_gt, w_x, w_q):
    """get_loss's arithmetic on the device in one launch (arguments in get_loss's order)."""
    L.require_gpu(l0_q, q_gt, w_x)
    B = q_gt.shape[0]
    poses = [_f32(x.reshape(B, n))[0] for x, n in ((l0_q, 4), (l0_t, 3), (l1_q, 4), (l1_t, 3), (l2_q, 4), (l2_t, 3), (l3_q, 4), (l3_t, 3))]
    return _PoseLoss.apply(w_x, w_q, _f32(q_gt.reshape(B, 4))[0], _f32(t_gt.reshape(B, 3))[0], *poses)


class ProjectionBuffers:
    """Outputs + scratch of one warp_project call, allocated ahead of it so that the pose head that produces its
    (q, t) can clear them on the side (`pose_head(clear=...)`): the warp call then skips its init launch."""

    def __init__(self, B, N, H, W, C, device, dtype=torch.float32):
        self.shape = (B, N, H, W, C)
        self.out_xyz = torch.empty((B, H, W, 3), dtype=torch.float32, device=device)
        self.out_feat = torch.empty((B, H, W, C), dtype=dtype, device=device) if C else None
        self.scratch = torch.empty((B * H * W + 4 * B + 2 * B * N,), dtype=torch.int32, device=device)   # include/elo.h
        self.cleared = False
        self.result = None          # (warped, out_xyz, out_feat) once a pose head has run the projection itself


class SvPartials:
    """softmax_valid's partial sums as a hand-over: the launch that produces the pose head's inputs (fused.mlp2_pair / fused.mlp
    with sv=...) reduces each of its row tiles to a (maximum, denominator, weighted sum) triple per channel and leaves them in
    `scratch`; pose_head(partials=...) merges them instead of running its partial-sums launch (include/elo.h elo_mlp_args.sv_*,
    elo_pose_head_args.ready_parts).  `parts` = slices per batch element, 0 until (unless) a launch has taken the ride.
    xyz: (B,N,3) the cloud whose validity masks the softmax; feature: (B,N,64) for a SINGLE launch (whose output are the logits)."""

    def __init__(self, xyz_bn3, feature_bnc=None):
        (self.xyz,) = _f32(xyz_bn3)
        self.feature = feature_bnc
        self.scratch = torch.empty((3 * xyz_bn3.shape[0] * L.SV_MAX_PARTS * 64,), dtype=torch.float32, device=xyz_bn3.device)
        self.parts = 0


class PoseRing:
    """(slots, B, 7) rows [q_norm | t] + a per-batch-element cursor: the pose output of a launch that is REPLAYED from a
    captured graph.  Replay r writes slot cursor % slots and advances the cursor, so a stream of frame pairs needs no
    copy-out per pair (elo_pose_head_args.pose7_slots / pose7_cursor); drain with rows() every `slots` replays."""

    def __init__(self, slots, batch, device):
        if slots < 2:
            raise ValueError("a pose ring has at least 2 slots")
        self.rows = torch.zeros((slots, batch, 7), dtype=torch.float32, device=device)
        self.cursor = torch.zeros((batch,), dtype=torch.int32, device=device)
        self.slots = slots

    def reset(self):
        """Cursor back to slot 0 (on the current stream)."""
        self.cursor.zero_()


def pose_head(feature_bnc, weight_bnc, xyz_bn3, W_big, b_big, W_q, b_q, W_t, b_t, q_coarse=None, t_coarse=None, pose7=None,
              clear=None, warp=None, next_orders=None, partials=None):
    """softmax_valid -> conv1d(256) -> q,t heads -> normalise -> compose with the coarse pose, two launches.
    pwclo_model.py:194-208 / :262-280.  Returns (q (B,4), t (B,3), q_norm (B,4)); `pose7` (B,7), if given, also receives [q_norm | t].
    `next_orders`: an elo_perm_refresh_args (perm.PermSource.refresh_args): the next pooled set of visiting orders is
    loaded by this launch once the pose is written (the last launch of a captured forward).
    `partials`: an SvPartials a preceding launch has filled (partials.parts > 0): no partial-sums launch here; `clear` must
    then already have been cleared by that launch (fused.mlp(clear=...))).
    `clear`: ProjectionBuffers of the projection that will consume this pose (cleared on the side).
    `warp` = (xyz (B,N,3), feat (B,N,C) or None) with `clear`: that projection itself -- warp by this pose, spherical
    re-projection -- is run by this call (elo_pose_head_warp: 3 launches instead of 2 + 2); its result is left in
    `clear.result` for the warp_project call that follows."""
    L.require_gpu(feature_bnc, weight_bnc, xyz_bn3, W_big, W_q, W_t, q_coarse, t_coarse)
    xyz_bn3, W_big, b_big, W_q, b_q, W_t, b_t = _f32(xyz_bn3, W_big, b_big, W_q, b_q, W_t, b_t)
    (feature_bnc, weight_bnc), fdt, fcode = _feature_dtype(feature_bnc, weight_bnc)      # fp32 or fp16 storage
    B, N, C = feature_bnc.shape
    hidden = W_big.shape[1]
    dev = feature_bnc.device
    if q_coarse is not None:
        q_coarse, t_coarse = _f32(q_coarse.reshape(B, 4), t_coarse.reshape(B, 3))
    q = torch.empty((B, 4), dtype=torch.float32, device=dev)
    t = torch.empty((B, 3), dtype=torch.float32, device=dev)
    q_norm = torch.empty((B, 4), dtype=torch.float32, device=dev)
    ready = partials.parts if partials is not None else 0
    if ready and C != 64:
        raise ValueError("partials= goes with the 64-channel head")
    scratch = partials.scratch if ready else torch.empty((3 * B * L.SV_MAX_PARTS * C,), dtype=torch.float32, device=dev)
    ptr = lambda x: x.data_ptr() if x is not None else None
    ring = pose7 if isinstance(pose7, PoseRing) else None
    if ring is not None:
        if ring.rows.shape[1] != B:
            raise ValueError("pose ring of batch %d used with batch %d" % (ring.rows.shape[1], B))
        pose7 = ring.rows
    a = L.PoseHeadArgs(B, N, C, hidden, feature_bnc.data_ptr(), weight_bnc.data_ptr(), xyz_bn3.data_ptr(),
                       W_big.data_ptr(), b_big.data_ptr(), W_q.data_ptr(), b_q.data_ptr(), W_t.data_ptr(), b_t.data_ptr(),
                       ptr(q_coarse), ptr(t_coarse), q.data_ptr(), t.data_ptr(), q_norm.data_ptr(), scratch.data_ptr(),
                       ptr(pose7), *((clear.scratch.data_ptr(), clear.out_xyz.data_ptr(), ptr(clear.out_feat),
                                      clear.shape[0] * clear.shape[2] * clear.shape[3], clear.shape[4])
                                     if clear is not None else (None, None, None, 0, 0)), fcode,
                       ring.slots if ring is not None else 0, ring.cursor.data_ptr() if ring is not None else None,
                       next_orders if next_orders is not None else L.PermRefreshArgs(), ready)
    if ready and clear is not None and not clear.cleared:
        raise ValueError("partials=: the ProjectionBuffers must have been cleared by an earlier launch (fused.mlp(clear=...))")
    if clear is not None and clear.out_feat is not None and clear.out_feat.dtype != fdt:
        raise TypeError("the projection buffers and the pose head's features must share one storage dtype")
    if warp is not None:
        if clear is None:
            raise ValueError("warp= needs clear= (the ProjectionBuffers the projection writes)")
        xyz_w, feat_w = warp
        (xyz_w,) = _f32(xyz_w)
        Bw, Nw, Hw, Ww, Cw = clear.shape
        if xyz_w.shape != (Bw, Nw, 3) or (Cw and (feat_w is None or feat_w.shape != (Bw, Nw, Cw))):
            raise ValueError("warp inputs do not match the ProjectionBuffers %s" % (clear.shape,))
        if feat_w is not None:
            if feat_w.dtype != fdt:
                raise TypeError("the warped features and the pose head's features must share one storage dtype")
            feat_w = feat_w.contiguous()
        warped = torch.empty((Bw, Nw, 3), dtype=torch.float32, device=dev)
        az, vres, voff = projection_constants(Hw, Ww)
        w = L.WarpProjectArgs(Bw, Nw, Cw, Hw, Ww, az, vres, voff, xyz_w.data_ptr(), ptr(feat_w), None, None,
                              warped.data_ptr(), clear.out_xyz.data_ptr(), ptr(clear.out_feat), clear.scratch.data_ptr(), 1, fcode)
        L.call2("elo_pose_head_warp", a, w, q)
        clear.result = (warped, clear.out_xyz, clear.out_feat)
        return q, t, q_norm
    L.call("elo_pose_head", a, q)
    if clear is not None:
        clear.cleared = True
    return q, t, q_norm


def projection_constants(H_input, W_input):
    """model_util.py:189-200: python doubles (cast to float32 by the ctypes struct)."""
    d2r = math.pi / 180
    az = (360.0 / W_input) * d2r
    down, up = -24.8 * d2r, 2.0 * d2r
    vres = (up - down) / (H_input - 1)
    return az, vres, -down / vres


def input_stage(cloud, T_trans, aug_frame, H, W, crop_xy=35.0):
    """elo_input_stage: cloud (B, 2N, S>=3) fp32, T_trans (B,4,4) or None, aug_frame (B) of 1/2 (array-like) ->
    (points (2B,N,3), xyz_proj (2B,H,W,3))."""
    L.require_gpu(cloud, T_trans)
    (cloud,) = _f32(cloud)
    B, N2, S = cloud.shape
    if N2 % 2 or S < 3:
        raise ValueError("point_cloud must be (B, 2*N, >=3)")
    N, dev = N2 // 2, cloud.device
    if T_trans is not None:
        (T_trans,) = _f32(T_trans.reshape(B, 4, 4))
        aug = torch.as_tensor(np.asarray(aug_frame).reshape(B), dtype=torch.int32).to(dev)
    points = torch.empty((2 * B, N, 3), dtype=torch.float32, device=dev)
    out_xyz = torch.empty((2 * B, H, W, 3), dtype=torch.float32, device=dev)
    scratch = torch.empty((2 * B * H * W + 4 * 2 * B + 2 * 2 * B * N,), dtype=torch.int32, device=dev)
    az, vres, voff = projection_constants(H, W)
    a = L.InputStageArgs(B, N, S, H, W, az, vres, voff, float(crop_xy), cloud.data_ptr(),
                         T_trans.data_ptr() if T_trans is not None else None,
                         aug.data_ptr() if T_trans is not None else None, points.data_ptr(), out_xyz.data_ptr(),
                         scratch.data_ptr())
    L.call("elo_input_stage", a, out_xyz)
    return points, out_xyz


class _WarpProject(torch.autograd.Function):
    """elo_warp_project / elo_warp_project_backward.  The forward's scratch (who won each cell) is kept for the backward."""

    @staticmethod
    def forward(ctx, xyz, feat, q, t, H, W):
        B, N, _ = xyz.shape
        C = 0 if feat is None else feat.shape[-1]
        buffers = ProjectionBuffers(B, N, H, W, C, xyz.device)
        warped, out_xyz, out_feat = _warp_project(xyz, feat, q, t, H, W, buffers)
        ctx.save_for_backward(xyz, q, t, buffers.scratch)
        ctx.dims = (B, N, C, H, W)
        return warped, out_xyz, out_feat

    @staticmethod
    def backward(ctx, g_warped, g_xyz_proj, g_feat_proj):
        xyz, q, t, scratch = ctx.saved_tensors
        B, N, C, H, W = ctx.dims
        dev, need = xyz.device, ctx.needs_input_grad
        cont = lambda g: None if g is None else _f32(g)[0]
        g_warped, g_xyz_proj, g_feat_proj = cont(g_warped), cont(g_xyz_proj), cont(g_feat_proj)
        g_x = torch.empty((B, N, 3), dtype=torch.float32, device=dev) if need[0] else None
        g_f = torch.empty((B, N, C), dtype=torch.float32, device=dev) if (need[1] and C) else None
        if g_f is not None and g_feat_proj is None:
            g_feat_proj = torch.zeros((B, H, W, C), dtype=torch.float32, device=dev)
        g_q = torch.zeros((B, 4), dtype=torch.float32, device=dev) if q is not None else None
        g_t = torch.zeros((B, 3), dtype=torch.float32, device=dev) if q is not None else None
        az, vres, voff = projection_constants(H, W)
        a = L.WarpProjectBwdArgs(B, N, C, H, W, az, vres, voff, xyz.data_ptr(), _ptr(q), _ptr(t), scratch.data_ptr(),
                                 _ptr(g_xyz_proj), _ptr(g_feat_proj), _ptr(g_warped), _ptr(g_x), _ptr(g_f), _ptr(g_q), _ptr(g_t))
        L.call("elo_warp_project_backward", a, xyz)
        return g_x, g_f, g_q, g_t, None, None


def warp_project(xyz, feat, q, t, H, W, buffers=None):
    """Optional quaternion warp (q,t: (B,4),(B,3) or None) + ProjectPC2SphericalRing.
    Returns (warped (B,N,3) or None, xyz_proj (B,H,W,3), feat_proj (B,H,W,C) or None).
    `buffers`: a ProjectionBuffers of this call's shape, used (and, if a pose head cleared it, not re-initialised)."""
    if buffers is None and _wants_grad(xyz, feat, q, t):
        B = xyz.shape[0]
        if feat is not None and feat.dtype != torch.float32:
            raise TypeError("training stores its features in float32")
        return _WarpProject.apply(_f32(xyz)[0], None if feat is None else feat.contiguous(),
                                  None if q is None else _f32(q.reshape(B, 4))[0],
                                  None if q is None else _f32(t.reshape(B, 3))[0], H, W)
    return _warp_project(xyz, feat, q, t, H, W, buffers)


def _warp_project(xyz, feat, q, t, H, W, buffers=None):
    if buffers is not None and buffers.result is not None:   # the pose head that produced (q, t) already did it
        result, buffers.result = buffers.result, None
        return result
    L.require_gpu(xyz, feat, q, t)
    (xyz,) = _f32(xyz)
    B, N, _ = xyz.shape
    C = 0 if feat is None else feat.shape[-1]
    fdt, fcode = torch.float32, L.ELO_F32
    if feat is not None:
        (feat,), fdt, fcode = _feature_dtype(feat)
    if q is not None:
        q, t = _f32(q.reshape(B, 4), t.reshape(B, 3))
    dev = xyz.device
    warped = torch.empty((B, N, 3), dtype=torch.float32, device=dev) if q is not None else None
    if buffers is not None and buffers.shape != (B, N, H, W, C):
        raise ValueError("ProjectionBuffers of shape %s given to a %s projection" % (buffers.shape, (B, N, H, W, C)))
    if buffers is None:
        buffers = ProjectionBuffers(B, N, H, W, C, dev, fdt)
    if buffers.out_feat is not None and buffers.out_feat.dtype != fdt:
        raise TypeError("ProjectionBuffers of dtype %s given to a projection of %s features" % (buffers.out_feat.dtype, fdt))
    out_xyz, out_feat, scratch = buffers.out_xyz, buffers.out_feat, buffers.scratch
    az, vres, voff = projection_constants(H, W)
    ptr = lambda x: x.data_ptr() if x is not None else None
    a = L.WarpProjectArgs(B, N, C, H, W, az, vres, voff, xyz.data_ptr(), ptr(feat), ptr(q), ptr(t), ptr(warped),
                          out_xyz.data_ptr(), ptr(out_feat), scratch.data_ptr(), 1 if buffers.cleared else 0, fcode)
    buffers.cleared = False                               # single use: the outputs now hold this call's result
    L.call("elo_warp_project", a, out_xyz)
    return warped, out_xyz, out_feat


# ---- training layer: conv2d -> batch norm (batch statistics) -> ReLU (utils/tf_util.py:120-185, :512-563) ----------
def dense_bn_supported(x2, cout):
    """The row-reduction kernels of csrc/elo_train.hip take fp32 (rows, C) matrices on the GPU with C a power of two in 4..256."""
    return x2.is_cuda and x2.dtype == torch.float32 and 4 <= cout <= 256 and cout & (cout - 1) == 0 and x2.shape[0] > 0


def _aligned16(*ts):
    return all(t.data_ptr() % 16 == 0 for t in ts)


def dense_rows_supported(x2, W):
    """elo_dense_rows takes fp32 (rows, Cin) x (Cin, Cout) on the GPU with Cout <= 192 and W within 160 KB of LDS."""
    return (x2.is_cuda and x2.dtype == torch.float32 and W.dtype == torch.float32 and x2.shape[0] > 0
            and bool(L.lib().elo_dense_rows_supported(x2.shape[0], W.shape[0], W.shape[1]))
            and bool(L.lib().elo_dense_rows_supported(x2.shape[0], W.shape[1], W.shape[0])))


def dense_rows(x2, W, bias=None, transposed=False, moments=None, bn_backward=None, groups=1):
    """x2 @ W + bias ((rows, Cin) x (Cin, Cout)), or x2 @ W.t() with transposed=True ((rows, Cout) x (Cin, Cout)^T), on
    csrc/elo_train_dense.hip.  moments = (eps, momentum, mean, invstd, running_mean, running_var): also the batch-norm moments of
    the result, written into mean / invstd (and the moving averages updated) -- what elo_bn_stats does in a second pass.
    bn_backward = (z, mean, invstd, gamma, beta, sums, relu, dz_out): x2 holds dy of a batch-normalised layer and the operand is that
    layer's dz, formed on the load from dy, z and the two sums of elo_bn_backward(dz=None) and written into dz_out on the way.
    groups > 1: the rows are that many equal blocks with their own batch statistics (mean, invstd (groups, C); sums (groups, 2C))."""
    L.require_gpu(x2, W)
    x2, Wc = x2.contiguous(), W.detach().contiguous()
    cin, cout = (Wc.shape[1], Wc.shape[0]) if transposed else (Wc.shape[0], Wc.shape[1])
    if x2.shape[1] != cin:
        raise ValueError("dense_rows: x is %s, W %s%s" % (tuple(x2.shape), tuple(W.shape), " (transposed)" if transposed else ""))
    out = torch.empty((x2.shape[0], cout), dtype=torch.float32, device=x2.device)
    bias_c = bias.detach().contiguous() if bias is not None else None
    head = (x2.shape[0], cin, cout, x2.data_ptr(), Wc.data_ptr(), 1 if transposed else 0, _ptr(bias_c), out.data_ptr())
    if moments is None:
        bn = (None,) * 6 + (0, None)
        if bn_backward is not None:
            z, mean, invstd, gamma, beta, sums, relu, dz_out = bn_backward
            bn = (z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), sums.data_ptr(), 1 if relu else 0, dz_out.data_ptr())
        a = L.DenseRowsArgs(*head, None, 0.0, 0.0, None, None, None, None, *bn, groups)
        L.call("elo_dense_rows", a, x2)
        return out
    eps, momentum, mean, invstd, running_mean, running_var = moments
    scratch = torch.empty((L.lib().elo_dense_rows_scratch_floats(cout, groups),), dtype=torch.float32, device=x2.device)
    a = L.DenseRowsArgs(*head, scratch.data_ptr(), eps, momentum, mean.data_ptr(), invstd.data_ptr(), _ptr(running_mean), _ptr(running_var),
                        None, None, None, None, None, None, 0, None, groups)
    L.call("elo_dense_rows", a, x2)
    return out


class _DenseBN(torch.autograd.Function):
    """y = act(batch_norm(x @ W + b)) with batch statistics.  The two dense products (forward, dx) run on elo_dense_rows where the layer
    has enough rows (tuning.train_dense*; the library GEMM below that); every pass OVER THE ROWS -- the batch moments (from the forward
    product's accumulators on the own kernel), the normalisation, the two sums of batch norm's backward, dz (on the dx kernel's operand
    load where that runs), the weight gradient x^T dz and the bias gradient -- is a hand-written kernel.  Saved for backward: x, W, the
    pre-normalisation z and the moments; the ReLU mask is recomputed from z.  The moving averages are updated in place exactly as
    F.batch_norm(training=True) does.  groups = G > 1: the rows are G equal blocks normalised with their OWN batch statistics (the
    reference calls the layer once per frame with shared variables, pwclo_model.py:117-143; one call on the 2B batch does the same
    arithmetic in half the launches): moments (G, C), moving averages updated G times in order."""

    @staticmethod
    def forward(ctx, x2, W, b, gamma, beta, running_mean, running_var, momentum, eps, relu, groups):
        x2 = x2.contiguous()
        M, C, G = x2.shape[0], W.shape[1], int(groups)
        dev = x2.device
        mean = torch.empty((G * C,), dtype=torch.float32, device=dev)
        invstd = torch.empty((G * C,), dtype=torch.float32, device=dev)
        g, bt = gamma.detach().contiguous(), beta.detach().contiguous()
        ctx.own_dense = bool(tuning.get("train_dense")) and dense_rows_supported(x2, W)
        wide_and_short = W.shape[0] > 128 and M < tuning.get("train_dense_dx_rows")
        if ctx.own_dense and M >= tuning.get("train_dense_rows") and not wide_and_short:
            # the product AND the batch moments in one pass over the rows
            z = dense_rows(x2, W, b, moments=(float(eps), float(momentum), mean, invstd, running_mean, running_var), groups=G)
        else:
            z = torch.addmm(b, x2, W)
            scratch = torch.empty((L.lib().elo_bn_scratch_floats(C, G),), dtype=torch.float32, device=dev)
            L.call("elo_bn_stats", L.BnStatsArgs(M, C, z.data_ptr(), scratch.data_ptr(), float(eps), float(momentum), mean.data_ptr(),
                                                 invstd.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), G), z)
        y = torch.empty_like(z)
        L.call("elo_bn_apply", L.BnApplyArgs(M, C, z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), g.data_ptr(), bt.data_ptr(),
                                             1 if relu else 0, y.data_ptr(), G), z)
        ctx.save_for_backward(x2, W, z, mean, invstd, g, bt)
        ctx.relu = bool(relu)
        ctx.groups = G
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, W, z, mean, invstd, g, bt = ctx.saved_tensors
        (dy,) = _f32(dy)
        M, C = z.shape
        G = ctx.groups
        dev = z.device
        scratch = torch.empty((L.lib().elo_bn_scratch_floats(C, G),), dtype=torch.float32, device=dev)
        sums = torch.empty((G * 2 * C,), dtype=torch.float32, device=dev)
        dz = torch.empty_like(z)
        own_dx = ctx.needs_input_grad[0] and ctx.own_dense and M >= tuning.get("train_dense_dx_rows")
        fused = own_dx and tuning.get("train_dense_fused_dz") and _aligned16(mean, invstd, g, bt)
        # (fused: the reduction's two launches only -- dz is formed by the dx kernel on its operand load and written from there)
        L.call("elo_bn_backward", L.BnBackwardArgs(M, C, dy.data_ptr(), z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), g.data_ptr(),
                                                   bt.data_ptr(), 1 if ctx.relu else 0, scratch.data_ptr(), sums.data_ptr(),
                                                   None if fused else dz.data_ptr(), G), z)
        total = sums if G == 1 else sums.view(G, 2 * C).sum(0)
        dbeta, dgamma = total[:C], total[C:]
        dx = None
        if fused:
            dx = dense_rows(dy, W, None, transposed=True, bn_backward=(z, mean, invstd, g, bt, sums, ctx.relu, dz), groups=G)
        elif ctx.needs_input_grad[0]:
            dx = dense_rows(dz, W, None, transposed=True) if own_dx else dz @ W.t()
        cin = W.shape[0]
        dW = torch.empty_like(W, memory_format=torch.contiguous_format)
        db = torch.empty((C,), dtype=torch.float32, device=dev)
        slices = L.lib().elo_weight_grad_slices(M, cin, C)
        wscratch = torch.empty((slices * (cin * C + C),), dtype=torch.float32, device=dev)
        L.call("elo_dense_weight_grad", L.WeightGradArgs(M, cin, C, x2.data_ptr(), dz.data_ptr(), dW.data_ptr(), db.data_ptr(),
                                                         wscratch.data_ptr()), z)
        return dx, dW, db, dgamma, dbeta, None, None, None, None, None, None


def dense_bn(x2, W, b, gamma, beta, running_mean, running_var, momentum, eps, relu, groups=1):
    """(rows, Cin) -> (rows, Cout): the training layer on the kernels above (dense_bn_supported(x2, Cout) must hold).  groups: see _DenseBN."""
    L.require_gpu(x2, W, gamma, running_mean)
    if groups < 1 or x2.shape[0] % groups:
        raise ValueError("dense_bn: %d rows do not split into %d groups" % (x2.shape[0], groups))
    return _DenseBN.apply(x2, W, b, gamma, beta, running_mean, running_var, momentum, eps, relu, groups)
