"""Operator library with the reference's names and argument lists
(utils/pointnet_util.py): warping_layers (:18-20), get_hw_idx (:23-30),
cost_volume (:33-149), flow_predictor (:153-175), down_conv (:179-250),
up_conv (:254-316).

Same call signatures, so the pyramid schedule (pwclo_model.py) calls them the
way the reference does.  Inference (is_training False): everything between two
poolings is ONE fused HIP kernel (csrc/elo_fused.hip).  Training, and
inference with ELO_FUSED=0: every chain of stock TF ops between two 1x1
convolutions is ONE hand-written HIP kernel (csrc/elo_features.hip; backward:
csrc/elo_backward.hip through _ops' autograd Functions), the two custom ops
are the HIP grouping kernels (csrc/elo_grouping.hip), and the 1x1 convolutions
are hipBLASLt GEMMs (tf_util.conv2d).  Concatenations feeding a
convolution are never materialised when they only join already-existing
tensors: conv(concat[a, b]) is computed as a @ W[:Ca] + b @ W[Ca:] (two
accumulating GEMMs on row-slices of the same weight variable).

`tf.random_shuffle(tf.range(KT))` becomes `random_shuffle(scope, tag, KT)` from
the active PermSource (see perm.py): the caller owns the randomness.
"""
import os

import torch

from . import _ops, fused, tf_util, tuning
from .fused_conv import fused_conv_random_k, fused_conv_select_k, register_dense_index
from .perm import random_shuffle


# Inference uses the fused "gather -> conv chain -> pool" kernels (csrc/elo_fused.hip); ELO_FUSED=0 or
# use_fused(False) selects the per-operator kernels + hipBLASLt GEMMs instead (same results to ~1e-6).
# The switch is tuning's host field "fused" (ELO_FUSED), read at the point of use: tuning.override(fused=...) and use_fused agree.
def use_fused(flag):
    tuning.set_host("fused", bool(flag))


def _fused_path(is_training):
    """The implementation is chosen by the CALLER's `is_training` (as in the reference, where it selects batch statistics
    and dropout), never by the global autograd mode: inference under torch.enable_grad() stays on the fused kernels."""
    return tuning.get("fused") and not is_training


# ---- fp16 feature STORAGE on the per-operator path (BASELINE configs[2] with ELO_FUSED=0) ---------------------------
# The storage contract (fused.storage, oracle/ops_np.feature_storage): every feature tensor an operator hands to the next
# one through HBM is fp16 -- the outputs of down_conv and flow_predictor and both halves of cost_volume and up_conv --,
# everything inside an operator is fp32.  The fused kernels read and write fp16 themselves; the per-operator kernels +
# hipBLASLt GEMMs keep their fp32 tensors INSIDE an operator, so here the boundary is explicit: fp16 feature arguments are
# widened on entry (exact), the result is rounded to fp16 on exit, and the stage-1 tensor of the two two-stage operators
# is rounded where the fused path stores it.  Same numbers as the oracle's storage mode, two elementwise launches per
# operator more than the fp32 run: this path is the HBM-roofline / training path, not the throughput one.
def _store_round(x):
    """x rounded to the storage dtype and widened again (the value the next stage would read back from HBM)."""
    return x.half().float() if fused.storage_dtype() == torch.float16 else x


def _storage_boundary(*feature_args, tuple_out=False):
    import functools
    import inspect

    def deco(fn):
        sig = inspect.signature(fn)

        @functools.wraps(fn)
        def wrapped(*args, **kw):
            b = sig.bind(*args, **kw)
            b.apply_defaults()
            if _fused_path(b.arguments["is_training"]):
                return fn(*args, **kw)
            halves = [n for n in feature_args if torch.is_tensor(b.arguments.get(n)) and b.arguments[n].dtype == torch.float16]
            if not halves:
                return fn(*args, **kw)
            if b.arguments["is_training"]:
                raise NotImplementedError("training stores its features in fp32")
            for n in halves:
                b.arguments[n] = b.arguments[n].float()
            with fused.storage(torch.float16):
                out = fn(*b.args, **b.kwargs)
            return ((out[0].half(),) + tuple(out[1:])) if tuple_out else out.half()
        return wrapped
    return deco


def warping_layers(xyz1, upsampled_flow):
    """utils/pointnet_util.py:18-20."""
    return xyz1 + upsampled_flow


_hw_cache = {}
_centre_hw_cache = {}


def _centre_hw(selected_idx):
    """(B,n,2) contiguous (h,w) columns of a (B,H',W',3) (b,h,w) index grid; cached for the cached grids
    model_util.get_selected_idx hands out (keyed by storage address + shape)."""
    key = (selected_idx.data_ptr(), tuple(selected_idx.shape), str(selected_idx.device))
    hit = _centre_hw_cache.get(key)
    if hit is None or hit[0] is not selected_idx:
        B = selected_idx.shape[0]
        hit = (selected_idx, selected_idx.reshape(B, -1, 3)[:, :, 1:].contiguous())
        if len(_centre_hw_cache) > 64:
            _centre_hw_cache.clear()
        _centre_hw_cache[key] = hit
    return hit[1]


def get_hw_idx(B, H, W, device="cuda"):
    """utils/pointnet_util.py:23-30 -> (B, H*W, 2) int32, row-major (h, w) of every pixel."""
    key = (B, H, W, str(device))
    if key not in _hw_cache:
        hh = torch.arange(H, dtype=torch.int32, device=device).view(1, H, 1, 1).expand(B, H, W, 1)
        ww = torch.arange(W, dtype=torch.int32, device=device).view(1, 1, W, 1).expand(B, H, W, 1)
        grid = torch.cat([hh, ww], -1).reshape(B, H * W, 2).contiguous()
        if grid.is_cuda and torch.cuda.is_current_stream_capturing():
            return grid                     # a tensor born inside a graph's private pool is never cached
        if len(_hw_cache) >= 256:
            _hw_cache.clear()
        _hw_cache[key] = register_dense_index(grid)      # random-k calls with it take the LDS-tiled kernel
    return _hw_cache[key]


def _split_conv(parts, num_output_channels, scope, is_training, bn_decay, bn=True):
    """tf_util.conv2d(tf.concat(parts, -1), ...) without building the concat.

    In inference the folded weight rows are sliced per part and the GEMMs accumulate into one output;
    in training it is the literal concat + conv (batch statistics, autograd)."""
    if is_training:
        return tf_util.conv2d(torch.cat(parts, -1), num_output_channels, [1, 1], padding='VALID', stride=[1, 1],
                              bn=bn, is_training=is_training, scope=scope, bn_decay=bn_decay)
    cin = sum(p.shape[-1] for p in parts)
    lead = parts[0].shape[:-1]
    W, b = tf_util.folded_variables(scope, cin, num_output_channels, (1, 1), bn)
    y, row = None, 0
    for i, p in enumerate(parts):
        c = p.shape[-1]
        x2 = p.reshape(-1, c)
        if y is None:
            y = torch.addmm(b, x2, W[row:row + c])
        elif i + 1 < len(parts):
            y = y.addmm_(x2, W[row:row + c])
        else:
            y = torch._addmm_activation(y, x2, W[row:row + c])      # relu(y + x @ W)
        row += c
    return y.reshape(lead + (num_output_channels,))


@_storage_boundary("points1_proj", "points2_proj")
def cost_volume(warped_xyz1_proj, xyz2_proj, points1_proj, points2_proj, kernel_size1, kernel_size2, nsample,
                nsample_q, distance, mlp1, mlp2, is_training, bn_decay, scope, bn=True, pooling='max', knn=True,
                corr_func='elementwise_product', side_jobs=None, side_chain=False):
    """Attentive cost volume, utils/pointnet_util.py:33-149.  Returns (B, H*W, mlp2[-1]).
    `side_jobs` (fused inference path only): one or two set-conv jobs (fused.setconv keyword dicts) that only share
    inputs with this cost volume; they run inside stage 1's launch and the call returns (cost, [(out, new_xyz), ...]).
    `side_chain`: they ride only on the chain form of stage 1 (fused.cv_stage1); else the second element is None."""
    with tf_util.variable_scope(scope):
        B, H, W, _ = warped_xyz1_proj.shape
        N = H * W
        warped_xyz1 = warped_xyz1_proj.reshape(B, N, 3)
        points1 = points1_proj.reshape(B, N, -1)
        dev = warped_xyz1_proj.device

        C = points1.shape[-1]
        random_HW_q = random_shuffle(tf_util.scope_name(), "random_HW_q", kernel_size2[0] * kernel_size2[1], dev, kernel_size2)
        random_HW_p = random_shuffle(tf_util.scope_name(), "random_HW_p", kernel_size1[0] * kernel_size1[1], dev, kernel_size1)
        if _fused_path(is_training) and nsample_q <= 32 and nsample <= 32:
            # two launches for the whole operator: each fused kernel does its own grouping (select-k / random-k),
            # gather + encoding, the conv chain on the matrix cores and the masked softmax pooling
            if list(mlp1) != [128, 64, 64] or list(mlp2) != [128, 64]:
                raise NotImplementedError("the fused cost volume is built for mlp1=[128,64,64], mlp2=[128,64]")
            P = fused.packed_layer
            stage1 = fused.cv_stage1(
                warped_xyz1, points1, xyz2_proj, points2_proj, None, None,
                P('CV_0', 10 + 2 * C, 128, row_order=fused.cv0_row_order(C)), P('CV_1', 128, 64), P('CV_2', 64, 64),
                P('CV_xyz', 10, 64),
                P('sum_CV_0', 128, 128, row_order=list(range(64, 128)) + list(range(64))),      # kernel order [x | enc]
                P('sum_CV_1', 128, 64),
                group=fused.Grouping(random_HW_q, kernel_size2, 1000), K=nsample_q, side=side_jobs, side_chain=side_chain)   # :49-100
            pi_feat1_new, side_out = stage1 if side_jobs else (stage1, None)
            pi_feat1_new = pi_feat1_new.reshape(B, H, W, -1)
            order = list(range(64 + C, 128 + C)) + list(range(64)) + list(range(64, 64 + C))   # [grouped | enc | feat1]
            cost = fused.cv_stage2(warped_xyz1_proj, points1_proj, pi_feat1_new, None, None,
                                   P('sum_xyz_encoding', 10, 64), P('sum_cost_volume_0', 128 + C, 128, row_order=order),
                                   P('sum_cost_volume_1', 128, 64),
                                   group=fused.Grouping(random_HW_p, kernel_size1, distance), K=nsample)   # :104-146
            return (cost, side_out) if side_jobs else cost
        if side_jobs:
            raise NotImplementedError("side_jobs ride on the fused cost-volume launch only")

        # ---- stage 1: point -> patch in frame 2 (:47-100)
        idx_hw = get_hw_idx(B, H, W, dev)
        qi_idx, _, _, valid_mask = fused_conv_select_k(
            warped_xyz1_proj, xyz2_proj, idx_hw, random_HW_q, H, W, N, kernel_size2[0], kernel_size2[1], nsample_q,
            flag_copy=0, distance=1000, stride_h=1, stride_w=1, want_valid=False)            # :49-51 (1000 is literal)
        mask = valid_mask.reshape(B, N, nsample_q)
        pi_feat1_new = _store_round(_cost_volume_stage1(warped_xyz1, points1, xyz2_proj, points2_proj, qi_idx, mask, mlp1, mlp2,
                                                        is_training, bn_decay)).reshape(B, H, W, -1)

        # ---- stage 2: patch -> patch inside frame 1 (:104-146)
        pc_idx, _, _, valid_mask2 = fused_conv_random_k(
            warped_xyz1_proj, warped_xyz1_proj, idx_hw, random_HW_p, H, W, N, kernel_size1[0], kernel_size1[1],
            nsample, flag_copy=0, distance=distance, stride_h=1, stride_w=1, want_valid=False)    # :106-108
        mask2 = valid_mask2.reshape(B, N, nsample)
        return _cost_volume_stage2(warped_xyz1_proj, points1_proj, pi_feat1_new, pc_idx, mask2, mlp1, mlp2, is_training,
                                   bn_decay)


def _cost_volume_stage1(warped_xyz1, points1, xyz2_proj, points2_proj, qi_idx, mask, mlp1, mlp2, is_training, bn_decay):
    """utils/pointnet_util.py:54-100 as per-operator kernels + GEMMs (caller holds the variable scope)."""
    B, N, nsample_q = mask.shape
    feat_cat = _ops.cv_encode1(warped_xyz1, points1, xyz2_proj, points2_proj, qi_idx, mask)       # :54-66
    xyz_cat = feat_cat.reshape(-1, feat_cat.shape[-1])[:, :10]      # :62 -- 2-D strided view, no copy (lda = 10+2C)
    x = feat_cat
    for j, num_out_channel in enumerate(mlp1):
        x = tf_util.conv2d(x, num_out_channel, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                           is_training=is_training, scope='CV_%d' % j, bn_decay=bn_decay)         # :72-76
    enc = tf_util.conv2d(xyz_cat, mlp1[-1], [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='CV_xyz', bn_decay=bn_decay)              # :79-82
    enc = enc.reshape(B, N, nsample_q, -1)
    cat = None
    for j, num_out_channel in enumerate(mlp2):                                                    # :84-90
        if j == 0:
            cat = _split_conv([enc, x], num_out_channel, 'sum_CV_0', is_training, bn_decay)
        else:
            cat = tf_util.conv2d(cat, num_out_channel, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                                 is_training=is_training, scope='sum_CV_%d' % j, bn_decay=bn_decay)
    return _ops.masked_softmax_pool(cat, x, mask)                                                 # :92-98


def _cost_volume_stage2(warped_xyz1_proj, points1_proj, pi_feat1_new, pc_idx, mask2, mlp1, mlp2, is_training, bn_decay):
    """utils/pointnet_util.py:110-146 as per-operator kernels + GEMMs (caller holds the variable scope)."""
    C = points1_proj.shape[-1]
    pc_xyz_cat, rest = _ops.cv_encode2(warped_xyz1_proj, points1_proj, pi_feat1_new, pc_idx, mask2)  # :110-120
    pc_enc = tf_util.conv2d(pc_xyz_cat, mlp1[-1], [1, 1], padding='VALID', stride=[1, 1], bn=True,
                            is_training=is_training, scope='sum_xyz_encoding', bn_decay=bn_decay)    # :123-126
    pc_cat = None
    for j, num_out_channel in enumerate(mlp2):                                                    # :129-135
        if j == 0:
            pc_cat = _split_conv([pc_enc, rest], num_out_channel, 'sum_cost_volume_0', is_training, bn_decay)
        else:
            pc_cat = tf_util.conv2d(pc_cat, num_out_channel, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                                    is_training=is_training, scope='sum_cost_volume_%d' % j, bn_decay=bn_decay)
    pc_points_grouped = rest[..., C:]                              # cost[idx]*mask, a channel slice of `rest`
    return _ops.masked_softmax_pool(pc_cat, pc_points_grouped, mask2)                             # :137-146


@_storage_boundary("points_f1", "upsampled_feat", "cost_volume")
def flow_predictor(points_f1, upsampled_feat, cost_volume, mlp, is_training, bn_decay, scope, bn=True, clear=None, sv=None):
    """utils/pointnet_util.py:153-175: MLP over concat[points_f1, upsampled_feat?, cost_volume?] -> (B,N,mlp[-1]).
    `sv` (fused inference only): an _ops.SvPartials -- the output are softmax_valid's logits and the launch also leaves its
    partial sums there (fused.mlp)."""
    with tf_util.variable_scope(scope):
        parts = [points_f1]
        if upsampled_feat is not None:
            parts.append(upsampled_feat)
        if cost_volume is not None:
            parts.append(cost_volume)
        if _fused_path(is_training) and len(mlp) <= 3:
            widths = [sum(p.shape[-1] for p in parts)] + list(mlp)
            layers = [fused.packed_layer('conv_predictor%d' % i, widths[i], widths[i + 1], bn=bn)
                      for i in range(len(mlp))]
            return fused.mlp(parts, layers, clear=clear, sv=sv)                                   # one launch (clear, sv: side jobs)
        parts = [p.unsqueeze(2) for p in parts]                                                   # :166
        x = None
        for i, num_out_channel in enumerate(mlp):
            if i == 0:
                x = _split_conv(parts, num_out_channel, 'conv_predictor0', is_training, bn_decay, bn=bn)
            else:
                x = tf_util.conv2d(x, num_out_channel, [1, 1], padding='VALID', stride=[1, 1], bn=bn,
                                   is_training=is_training, scope='conv_predictor%d' % i, bn_decay=bn_decay)
        return x.squeeze(2)


def fused_pairs_available(is_training):
    """True when the paired-launch forms below can be used (fused inference path)."""
    return _fused_path(is_training)


# Branches that only share inputs ride on ONE heterogeneous launch (fused.cv_stage1(side=...)) while the GPU is
# underfilled: measured at 64x1800 (8 lanes) batch 1 8770 -> 9290 pairs/s, one lane 2700 -> 3020; at batch 8 the merged
# grid is slower (15 800 -> 15 200: the set-conv tiles run at the cost-volume kernel's register / LDS footprint), so
# the merge is taken only in forwards of up to `merge_points` centre points at the finest level (batch x H x W of l0; end of
# round 2, pairs/s at 8 lanes, none / levels under the threshold / all: batch 2 13 330 / 13 940 / 14 090, batch 4 16 650 /
# 15 990 / 16 530, batch 8 18 170 / 18 150 / 17 790) -- and, since round 5, there only at the LEVELS of up to
# `merge_level_points` centre points: with the register-resident kernels of rounds 3-4 a level whose merged grid would not
# fit the GPU at once (l0 of a 64 x 1800 pair: 2520 workgroups on 1280 slots) is faster as a cost-volume chain launch + a
# set-conv chain launch, once four forwards are in flight (batch 1, 8 lanes: 11.14 -> 11.74 k pairs/s with l0 unmerged and on
# the chain kernels; one lane alone 3.55 -> 3.46 k: profiles/r05_batch1_regimes.txt).


def merge_branches(is_training, finest_points, level_points=0):
    return (_fused_path(is_training) and finest_points <= tuning.get("merge_points") and
            level_points <= tuning.get("merge_level_points"))


def flow_predictor_pair(call_a, call_b):
    """Two flow_predictor calls with identical shapes (dicts of flow_predictor keyword arguments) in ONE launch
    (pwclo_model.py:253-254: the embedding and the embedding-mask predictors of a refinement level)."""
    specs = []
    for c in (call_a, call_b):
        with tf_util.variable_scope(c["scope"]):
            parts = [p for p in (c["points_f1"], c["upsampled_feat"], c["cost_volume"]) if p is not None]
            widths = [sum(p.shape[-1] for p in parts)] + list(c["mlp"])
            layers = [fused.packed_layer('conv_predictor%d' % i, widths[i], widths[i + 1], bn=c.get("bn", True))
                      for i in range(len(c["mlp"]))]
        specs.append((parts, layers))
    return fused.mlp_pair(specs[0][0], specs[0][1], specs[1][0], specs[1][1])


def up_conv_pair(call_a, call_b):
    """Two up_conv calls that differ only in scope and feat2_proj (pwclo_model.py:247,250) in TWO launches
    instead of four: paired set-conv stage (grouping + gather + MLP + max-pool) and paired stage-2 MLP."""
    jobs, stage2 = [], []
    for c in (call_a, call_b):
        with tf_util.variable_scope(c["scope"]):
            xyz1_proj, feat1_proj, feat2_proj = c["xyz1_proj"], c["feat1_proj"], c["feat2_proj"]
            B, H, W, _ = xyz1_proj.shape
            ks, mlp, mlp2 = c["kernel_size"], c["mlp"], c["mlp2"]
            random_HW = random_shuffle(tf_util.scope_name(), "random_HW", ks[0] * ks[1], xyz1_proj.device, ks)
            P = fused.packed_layer
            w1 = [3 + feat2_proj.shape[-1]] + list(mlp)
            jobs.append(dict(src_xyz=c["xyz2_proj"], src_feat=feat2_proj, idx=None, mask=None,
                             layers=[P('up_1_%d' % j, w1[j], w1[j + 1],
                                       row_order=fused.setconv_row_order(w1[0] - 3) if j == 0 else None)
                                     for j in range(len(mlp))],
                             xyz1_grid=xyz1_proj, K=c["nsample"],
                             group=fused.Grouping(random_HW, ks, c["distance"], c["stride_h"], c["stride_w"])))
            points1 = feat1_proj.reshape(B, H * W, -1)
            w2 = [mlp[-1] + points1.shape[-1]] + list(mlp2)
            stage2.append((points1, [P('up_2_%d' % i, w2[i], w2[i + 1]) for i in range(len(mlp2))]))
    (up_a, _), (up_b, _) = fused.setconv_pair(jobs[0], jobs[1])
    return fused.mlp_pair([up_a, stage2[0][0]], stage2[0][1], [up_b, stage2[1][0]], stage2[1][1])


def up_conv_predict_pair(up_a, up_b, fp_a, fp_b):
    """The two set-upconvs of a refinement level and the two flow predictors they feed (pwclo_model.py:247-254) in
    TWO launches: the paired set-conv stage, then ONE launch that runs each up_conv's stage-2 MLP and, on its
    output, the predictor's MLP (flow_predictor's concat [points_f1, upsampled_feat, cost_volume] with
    upsampled_feat coming straight from the tile).  up_x / fp_x: the keyword dicts of up_conv_pair /
    flow_predictor_pair; fp_x's upsampled_feat is up_x's output.  Returns (up_out_a, predictor_a, up_out_b, predictor_b)."""
    jobs = up_conv_stage1_jobs(up_a, up_b)
    (up_a_pooled, _), (up_b_pooled, _) = fused.setconv_pair(jobs[0], jobs[1])
    return up_conv_predict_finish(up_a, up_b, fp_a, fp_b, up_a_pooled, up_b_pooled)


def up_conv_stage1_jobs(up_a, up_b):
    """Stage 1 (grouping + gather + MLP + max-pool, pointnet_util.py:272-298) of the two set-upconvs of a level as two
    fused.setconv job dicts: for fused.setconv_pair, or as `side_jobs` of the level's cost volume (they only share
    inputs with it, so they can ride on its first launch)."""
    jobs = []
    for up in (up_a, up_b):
        P = fused.packed_layer
        with tf_util.variable_scope(up["scope"]):
            feat2_proj, ks, mlp = up["feat2_proj"], up["kernel_size"], up["mlp"]
            random_HW = random_shuffle(tf_util.scope_name(), "random_HW", ks[0] * ks[1], up["xyz1_proj"].device, ks)
            w1 = [3 + feat2_proj.shape[-1]] + list(mlp)
            jobs.append(dict(src_xyz=up["xyz2_proj"], src_feat=feat2_proj, idx=None, mask=None,
                             layers=[P('up_1_%d' % j, w1[j], w1[j + 1],
                                       row_order=fused.setconv_row_order(w1[0] - 3) if j == 0 else None)
                                     for j in range(len(mlp))],
                             xyz1_grid=up["xyz1_proj"], K=up["nsample"],
                             group=fused.Grouping(random_HW, ks, up["distance"], up["stride_h"], up["stride_w"])))
    return jobs


def up_conv_predict_finish(up_a, up_b, fp_a, fp_b, up_a_pooled, up_b_pooled, clear=None, sv=None):
    """Stage 2 of both set-upconvs and the two flow predictors they feed, in ONE launch (see up_conv_predict_pair).
    `sv`: an _ops.SvPartials: predictor a's output are softmax_valid's logits, predictor b's its features (fused.mlp2_pair)."""
    stage2 = []
    for up, fp, pooled in ((up_a, fp_a, up_a_pooled), (up_b, fp_b, up_b_pooled)):
        P = fused.packed_layer
        with tf_util.variable_scope(up["scope"]):
            feat1_proj, mlp, mlp2 = up["feat1_proj"], up["mlp"], up["mlp2"]
            B, H, W, _ = up["xyz1_proj"].shape
            points1 = feat1_proj.reshape(B, H * W, -1)
            w2 = [mlp[-1] + points1.shape[-1]] + list(mlp2)
            layers = [P('up_2_%d' % i, w2[i], w2[i + 1]) for i in range(len(mlp2))]
        with tf_util.variable_scope(fp["scope"]):
            before, after = fp["points_f1"], fp["cost_volume"]
            wp = [sum(p.shape[-1] for p in (before, after) if p is not None) + mlp2[-1]] + list(fp["mlp"])
            w_before = before.shape[-1] if before is not None else 0
            w_after = after.shape[-1] if after is not None else 0
            layers2 = [P('conv_predictor%d' % i, wp[i], wp[i + 1], bn=fp.get("bn", True),
                         row_order=fused.stage2_row_order(w_before, mlp2[-1], w_after) if i == 0 else None)
                       for i in range(len(fp["mlp"]))]
        stage2.append(dict(sources=[pooled, points1], layers=layers, before=before, after=after, layers2=layers2))
    (out_a, pred_a), (out_b, pred_b) = fused.mlp2_pair(stage2[0], stage2[1], clear=clear, sv=sv)
    return out_a, pred_a, out_b, pred_b


def down_conv_job(xyz_proj, points_proj, selected_idx, K_sample, kernel_size, distance, mlp, scope, bn=True):
    """down_conv (fused inference form) as a DEFERRED job: returns (job, finish) where `job` is a fused.setconv keyword dict
    -- to run on its own or as a `side_jobs` entry of a cost volume that only shares inputs with it -- and
    finish(out, new_xyz) gives down_conv's return value."""
    with tf_util.variable_scope(scope):
        random_HW = random_shuffle(tf_util.scope_name(), "random_HW", kernel_size[0] * kernel_size[1], xyz_proj.device, kernel_size)
        widths = [3 + points_proj.shape[-1]] + list(mlp)
        layers = [fused.packed_layer('conv%d' % i, widths[i], widths[i + 1], bn=bn,
                                     row_order=fused.setconv_row_order(widths[0] - 3) if i == 0 else None)
                  for i in range(len(mlp))]
    job = dict(src_xyz=xyz_proj, src_feat=points_proj, idx=None, mask=None, layers=layers, xyz1_grid=xyz_proj,
               centre_hw=_centre_hw(selected_idx), K=K_sample, group=fused.Grouping(random_HW, kernel_size, distance))
    return job, lambda out, new_xyz: (out, new_xyz.reshape(selected_idx.shape[:-1] + (3,)))


@_storage_boundary("points_proj", tuple_out=True)
def down_conv(xyz_proj, points_proj, selected_idx, K_sample, kernel_size, distance, mlp, mlp2, flag_add, is_training,
              bn_decay, scope, bn=True, pooling='max', knn=False, use_xyz=True, use_nchw=False):
    """Set-conv, utils/pointnet_util.py:179-250.  Returns ((B, n, mlp[-1]), new_xyz_proj (B,H',W',3))."""
    if use_nchw or pooling != 'max' or mlp2 is not None:
        raise NotImplementedError("the model uses NHWC, max pooling and mlp2=None (pwclo_model.py:126-177)")
    with tf_util.variable_scope(scope):
        B, H, W, _ = xyz_proj.shape
        idx_n2 = selected_idx.reshape(B, -1, 3)
        n_sampled = idx_n2.shape[1]
        dev = xyz_proj.device
        random_HW = random_shuffle(tf_util.scope_name(), "random_HW", kernel_size[0] * kernel_size[1], dev, kernel_size)
        centre_hw = _centre_hw(selected_idx)
        if _fused_path(is_training) and len(mlp) <= 3 and K_sample <= 32:
            # ONE launch: random-k grouping, gather, centre-subtract, MLP on the matrix cores, masked max-pool.
            # The centre is xyz_proj[b, h, w] (selected_idx's batch column is the batch index, as
            # get_selected_idx builds it)                                                           :197-230
            widths = [3 + points_proj.shape[-1]] + list(mlp)
            layers = [fused.packed_layer('conv%d' % i, widths[i], widths[i + 1], bn=bn,
                                         row_order=fused.setconv_row_order(widths[0] - 3) if i == 0 else None)
                      for i in range(len(mlp))]
            out, new_xyz = fused.setconv(xyz_proj, points_proj, None, None, layers, xyz1_grid=xyz_proj,
                                         centre_hw=centre_hw, K=K_sample,
                                         group=fused.Grouping(random_HW, kernel_size, distance))
            return out, new_xyz.reshape(selected_idx.shape[:-1] + (3,))
        sel, _, _, valid_mask = fused_conv_random_k(
            xyz_proj, xyz_proj, centre_hw, random_HW, H, W, n_sampled, kernel_size[0],
            kernel_size[1], K_sample, flag_copy=0, distance=distance, stride_h=1, stride_w=1, want_valid=False)  # :197-199
        mask = valid_mask.reshape(B, n_sampled, K_sample)
        li = selected_idx.reshape(-1, 3).long()
        new_xyz_proj = xyz_proj[li[:, 0], li[:, 1], li[:, 2]].reshape(selected_idx.shape[:-1] + (3,))  # :206
        new_xyz = new_xyz_proj.reshape(B, -1, 3)
        x = _ops.group_concat(new_xyz, xyz_proj, points_proj, sel, mask)                          # :203-213
        for i, num_out_channel in enumerate(mlp):
            x = tf_util.conv2d(x, num_out_channel, [1, 1], padding='VALID', stride=[1, 1], bn=bn,
                               is_training=is_training, scope='conv%d' % i, bn_decay=bn_decay)    # :217-222
        return _ops.masked_maxpool(x, mask), new_xyz_proj                                         # :224-230


@_storage_boundary("feat1_proj", "feat2_proj")
def up_conv(xyz1_proj, xyz2_proj, feat1_proj, feat2_proj, kernel_size, stride_h, stride_w, nsample, distance, mlp,
            mlp2, is_training, scope, bn_decay=None, bn=True, pooling='max', radius=None, knn=True):
    """Set-upconv (embedding and embedding-mask up-convolution), utils/pointnet_util.py:254-316."""
    with tf_util.variable_scope(scope):
        B, H, W, _ = xyz1_proj.shape
        N = H * W
        dev = xyz1_proj.device
        xyz1 = xyz1_proj.reshape(B, N, 3)
        points1 = feat1_proj.reshape(B, N, -1)
        random_HW = random_shuffle(tf_util.scope_name(), "random_HW", kernel_size[0] * kernel_size[1], dev, kernel_size)
        if _fused_path(is_training) and len(mlp) <= 3 and len(mlp2) <= 3 and nsample <= 32:
            P = fused.packed_layer
            w1 = [3 + feat2_proj.shape[-1]] + list(mlp)
            up_feat, _ = fused.setconv(xyz2_proj, feat2_proj, None, None,
                                       [P('up_1_%d' % j, w1[j], w1[j + 1],
                                          row_order=fused.setconv_row_order(w1[0] - 3) if j == 0 else None)
                                        for j in range(len(mlp))],
                                       xyz1_grid=xyz1_proj, K=nsample,
                                       group=fused.Grouping(random_HW, kernel_size, distance, stride_h, stride_w))  # :272-298
            w2 = [mlp[-1] + points1.shape[-1]] + list(mlp2)
            return fused.mlp([up_feat, points1], [P('up_2_%d' % i, w2[i], w2[i + 1]) for i in range(len(mlp2))])  # :303-311
        idx_hw = get_hw_idx(B, H, W, dev)
        sel, _, _, valid_mask = fused_conv_random_k(
            xyz1_proj, xyz2_proj, idx_hw, random_HW, H, W, N, kernel_size[0], kernel_size[1], nsample,
            flag_copy=0, distance=distance, stride_h=stride_h, stride_w=stride_w, want_valid=False)      # :272-274
        mask = valid_mask.reshape(B, N, nsample)
        x = _ops.group_concat(xyz1, xyz2_proj, feat2_proj, sel, mask)                             # :277-284
        for j, num_out_channel in enumerate(mlp):
            x = tf_util.conv2d(x, num_out_channel, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                               is_training=is_training, scope='up_1_%d' % j, bn_decay=bn_decay)   # :289-293
        up_feat = _store_round(_ops.masked_maxpool(x, mask))                                      # :295-298
        y = None
        for i, num_out_channel in enumerate(mlp2):                                                # :303-311
            if i == 0:
                y = _split_conv([up_feat.unsqueeze(2), points1.unsqueeze(2)], num_out_channel, 'up_2_0',
                                is_training, bn_decay)
            else:
                y = tf_util.conv2d(y, num_out_channel, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                                   is_training=is_training, scope='up_2_%d' % i, bn_decay=bn_decay)
        return y.squeeze(2)
