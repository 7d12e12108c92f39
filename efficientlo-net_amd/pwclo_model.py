"""The pyramid schedule, get_model / get_loss of pwclo_model.py (:30-433, :437-481),
calling the operator library with the reference's own call shapes.

Differences from the reference, all documented in DESIGN.md:
  * TF graph-building -> eager PyTorch under a VariableStore (tf_util.py) and a
    PermSource (perm.py); capture with model.PWCLONet for HIP-graph replay.
  * the re-projection sizes (4,57), (8,113), (16,225) that the reference
    hard-codes (:232,:306,:380) come from out_h_list/out_w_list so other
    resolutions (128x2048) work.
  * `get_model_from_projection` starts at the two (B,H,W,3) range images;
    `get_model` keeps the reference signature and does PreProcess + input
    projection first (:54-67).
"""
import math
import os

import numpy as np
import torch

from . import _ops, fused, tf_util, tuning
from .model_util import (PreProcess, ProjectPC2SphericalRing, input_stage, preprocess_gt, get_selected_idx, inv_q, mul_point_q, mul_q_point,
                         softmax_valid, warp_and_project)
from .pointnet_util import (cost_volume, down_conv, down_conv_job, flow_predictor, fused_pairs_available, merge_branches,
                            up_conv, up_conv_predict_finish, up_conv_stage1_jobs)

Down_conv_dis = [0.5, 3.0, 6.0, 12.0]       # pwclo_model.py:38
Up_conv_dis = [3.0, 6.0, 9.0]               # :39
Cost_volume_dis = [1.0, 2.0, 4.0]           # :40
stride_h_list = [1, 1, 4, 2, 2, 1]          # :42
stride_w_list = [1, 1, 8, 2, 2, 2]          # :43


def pyramid_sizes(H_input, W_input):
    """pwclo_model.py:45-50."""
    out_h = [math.ceil(H_input / stride_h_list[0])]
    out_w = [math.ceil(W_input / stride_w_list[0])]
    for i in range(1, 6):
        out_h.append(math.ceil(out_h[i - 1] / stride_h_list[i]))
        out_w.append(math.ceil(out_w[i - 1] / stride_w_list[i]))
    return out_h, out_w


def variable_specs():
    """Every conv layer of the model as (scope, Cin, Cout, tf_kernel_dims, has_bn), in creation order.
    Matches the shipped checkpoint index name for name (tests/golden/ckpt_index_shapes.json,
    SURVEY.md Appendix B); lets a VariableStore be populated without running a forward pass."""
    specs = []

    def chain(prefix, names, cin, couts, bn=True, dims=(1, 1)):
        for n, cout in zip(names, couts):
            specs.append((prefix + n, cin, cout, dims, bn))
            cin = cout

    conv3 = ['conv0', 'conv1', 'conv2']
    chain('sa1/layer0/', conv3, 3 + 3, [8, 8, 16])                  # pwclo_model.py:126
    chain('sa1/layer1/', conv3, 3 + 16, [16, 16, 32])               # :130
    chain('sa1/layer2/', conv3, 3 + 32, [32, 32, 64])               # :134
    chain('sa1/layer3/', conv3, 3 + 64, [64, 64, 128])              # :138
    for scope, C in (('flow_embedding_l2_origin', 64), ('flow_embedding_l2', 64), ('flow_embedding_l1', 32),
                     ('flow_embedding_l0', 16)):                    # :170,:242,:316,:390
        chain(scope + '/', ['CV_0', 'CV_1', 'CV_2'], 10 + 2 * C, [128, 64, 64])
        chain(scope + '/', ['CV_xyz'], 10, [64])
        chain(scope + '/', ['sum_CV_0', 'sum_CV_1'], 128, [128, 64])
        chain(scope + '/', ['sum_xyz_encoding'], 10, [64])
        chain(scope + '/', ['sum_cost_volume_0', 'sum_cost_volume_1'], 64 + C + 64, [128, 64])
    chain('new_layer3/', conv3, 3 + 64, [128, 64, 64])              # :177
    pred = ['conv_predictor0', 'conv_predictor1']
    chain('l3_costvolume_predict_ww/', pred, 128 + 64, [128, 64])   # :187
    for level, C in ((2, 64), (1, 32), (0, 16)):
        for kind in ('w', 'costvolume'):                            # :247,:250
            chain('up_sa_layer_layer_l%d%s/' % (level, kind), ['up_1_0', 'up_1_1'], 3 + 64, [128, 64])
            chain('up_sa_layer_layer_l%d%s/' % (level, kind), ['up_2_0', 'up_2_1'], 64 + C, [128, 64])
        chain('l%d_costvolume_predict/' % level, pred, C + 64 + 64, [128, 64])      # :253
        chain('l%d_w_predict/' % level, pred, C + 64 + 64, [128, 64])               # :254
    for level in (3, 2, 1, 0):                                      # :197-208, :264-273
        heads = ('q_coarse', 't_coarse') if level == 3 else ('q_det', 't_det')
        specs.append(('l%d_big' % level, 64, 256, (1,), False))
        specs.append(('l%d_%s' % (level, heads[0]), 256, 4, (1,), False))
        specs.append(('l%d_%s' % (level, heads[1]), 256, 3, (1,), False))
    return specs


def create_variables(store=None):
    """Populate the active (or given) VariableStore with every variable of the model."""
    store = store if store is not None else tf_util.get_store()
    with tf_util.default_store(store):
        for scope, cin, cout, dims, bn in variable_specs():
            tf_util.dense_variables(scope, cin, cout, dims, bn)
    return store


def placeholder_inputs(batch_size, NUM_POINTS, device="cuda"):
    """pwclo_model.py:19-27: zero tensors of the feed shapes."""
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=device)
    return z(batch_size, NUM_POINTS * 2, 6), z(batch_size, 4, 4), z(batch_size, 4, 4), z(batch_size, 4, 4)


def _parallel(branches, is_training=False):
    """Independent closures of a refinement level, run in order on the current stream.  (Forking them onto side streams was built and
    measured in rounds 1 and 5 -- inference: 8 % slower at one lane, 35 % at eight; the captured training step 15.2 -> 15.9 ms --
    cross-stream edges in a hipGraph cost more than the overlap of these small kernels gives: DESIGN.md.)"""
    return [b() for b in branches]


def _adjacent_frames(a, b):
    """If b starts exactly where a ends in the same allocation (PWCLONet keeps both range images in one
    (2B,H,W,3) buffer), return that (2B,H,W,3) tensor as a view -- no copy; else None."""
    if (a.shape == b.shape and a.dtype == b.dtype and a.device == b.device and a.is_contiguous() and b.is_contiguous()
            and b.data_ptr() == a.data_ptr() + a.numel() * a.element_size()
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()):
        return torch.as_strided(a, (2 * a.shape[0],) + tuple(a.shape[1:]), a.stride())
    return None


# Test hook: a list here receives (level, scratch, B, N, H, W, projected xyz grid) of every warp-refinement projection of an
# eager inference forward -- `scratch` is elo_warp_project_args.scratch, whose words [B*H*W + 4B, B*H*W + 4B + B*N) are the
# projection cell of every warped point: the product's own DISCRETE decisions (which cell, and through the grid which point
# won it), for tests/test_parity_flips_gpu.py to compare with the oracle's; the warped points themselves come last.
PROJECTION_TAP = None

_ZEROS = {}


def _zero_features(like):
    """The all-zero input features (pwclo_model.py:69-70).  Read-only, so one tensor per shape is shared
    instead of a fill launch per forward."""
    dtype = fused.storage_dtype()                   # fp16 feature storage starts here: every kernel keeps its inputs' dtype
    key = (tuple(like.shape), like.device, dtype)
    if key not in _ZEROS:
        if like.is_cuda and torch.cuda.is_current_stream_capturing():
            return torch.zeros_like(like, dtype=dtype)   # never cache a tensor that lives in a graph's private pool
        _ZEROS[key] = torch.zeros_like(like, dtype=dtype)
    return _ZEROS[key]


def _normalise_q(q):
    return q / (torch.sqrt((q * q).sum(-1, keepdim=True) + 1e-10) + 1e-10)


def _pose_head(feat_b1c, level, coarse, is_training, raw=False):
    """pwclo_model.py:197-208 (l3) and :264-273 / :340-349 / :408-417 (refinement levels)."""
    # (the reference leaves is_training at its default for these BN-free layers; here it also selects the autograd path)
    big = tf_util.conv1d(feat_b1c, 256, 1, padding='VALID', activation_fn=None, scope='l%d_big' % level, is_training=is_training)
    if is_training:
        big = torch.nn.functional.dropout(big, p=0.5, training=True)                                # :199
    qn, tn = ('l%d_q_coarse', 'l%d_t_coarse') if coarse else ('l%d_q_det', 'l%d_t_det')
    q = tf_util.conv1d(big, 4, 1, padding='VALID', activation_fn=None, scope=qn % level, is_training=is_training)
    t = tf_util.conv1d(big, 3, 1, padding='VALID', activation_fn=None, scope=tn % level, is_training=is_training)
    return (q if raw else _normalise_q(q)), t


def _estimate_pose(predict, weight, xyz, level, q_coarse, t_coarse, is_training, pose7=None, clear=None, warp=None, partials=None):
    """softmax_valid + pose head + composition with the coarse pose (q_coarse None at l3).
    Returns (q (B,4), t (B,3), q_norm (B,4)): the level's pose and its final normalisation (:427-430).
    Inference: two fused HIP launches (_ops.pose_head).  Training: the literal operator chain."""
    batch_size = predict.shape[0]
    coarse = q_coarse is None
    if not is_training:
        qn, tn = ('l%d_q_coarse', 'l%d_t_coarse') if coarse else ('l%d_q_det', 'l%d_t_det')
        W_big, b_big = tf_util.folded_variables('l%d_big' % level, predict.shape[-1], 256, (1,), bn=False)
        W_q, b_q = tf_util.folded_variables(qn % level, 256, 4, (1,), bn=False)
        W_t, b_t = tf_util.folded_variables(tn % level, 256, 3, (1,), bn=False)
        # the l0 head is the last launch of a forward: in a graph captured with fresh_orders it also loads the NEXT replay's orders
        from . import perm
        nxt = perm.tail_refresh_args() if level == 0 else None
        return _ops.pose_head(predict, weight, xyz, W_big, b_big, W_q, b_q, W_t, b_t, q_coarse, t_coarse, pose7, clear, warp, nxt, partials)
    summed = softmax_valid(feature_bnc=predict, weight_bnc=weight, mask_valid=xyz)                  # :194 / :262
    if predict.is_cuda and pose7 is None:
        # the pose algebra of :206-208 / :271-280 in ONE launch forward and one backward (_ops.pose_compose; the literal chain
        # below is ~60 + ~120 eight-element torch kernels per level)
        q_raw, t_det = _pose_head(summed, level, coarse, is_training, raw=True)                     # :197-205 / :264-270
        return _ops.pose_compose(q_raw, t_det, q_coarse, t_coarse)
    q_det, t_det = _pose_head(summed, level, coarse, is_training)                                   # :197-208 / :264-273
    if coarse:
        q = q_det.squeeze(1)
        return q, t_det.squeeze(1), _normalise_q(q)
    dev = predict.device
    t_coarse_trans = torch.cat([torch.zeros((batch_size, 1, 1), dtype=torch.float32, device=dev), t_coarse], -1)
    t_coarse_trans = mul_q_point(q_det, t_coarse_trans, batch_size)
    t_coarse_trans = mul_point_q(t_coarse_trans, inv_q(q_det, batch_size), batch_size)[:, :, 1:]    # :275-277
    q = mul_point_q(q_det, q_coarse, batch_size).squeeze(1)                                         # :279
    t = (t_coarse_trans + t_det).squeeze(1)                                                         # :280
    q_norm = _normalise_q(q)
    if pose7 is not None:
        pose7.copy_(torch.cat([q_norm, t], -1).detach())
    return q, t, q_norm


def get_model_from_projection(xyz_f1_input_proj, xyz_f2_input_proj, is_training, bn_decay=None, pose_out=None):
    """pwclo_model.py:69-433 from the projected inputs on.
    Returns (l0_q_norm, l0_t, l1_q_norm, l1_t, l2_q_norm, l2_t, l3_q_norm, l3_t, l0_xyz_f1).
    `pose_out` (B,7), if given, also receives [l0_q_norm | l0_t] (written by the l0 pose-head kernel: the row a
    caller logs per frame pair, main.py:557-572)."""
    batch_size, H_input, W_input, _ = xyz_f1_input_proj.shape
    dev = xyz_f1_input_proj.device
    out_h_list, out_w_list = pyramid_sizes(H_input, W_input)

    zero_features = _zero_features                                                                  # :69-70
    # (pointnet_util.merge_branches: a forward small enough to merge at all, then level by level)
    finest_points = batch_size * out_h_list[2] * out_w_list[2]

    # strided centre grids (:88-114).  Only the index tensors matter: down_conv re-gathers the xyz itself.
    pre2 = xyz_f1_input_proj
    l0_selected_idx = get_selected_idx(pre2, stride_h_list[2], stride_w_list[2], out_h_list[2], out_w_list[2])
    l1_selected_idx = get_selected_idx(pre2, stride_h_list[3], stride_w_list[3], out_h_list[3], out_w_list[3])
    l2_selected_idx = get_selected_idx(pre2, stride_h_list[4], stride_w_list[4], out_h_list[4], out_w_list[4])
    l3_selected_idx = get_selected_idx(pre2, stride_h_list[5], stride_w_list[5], out_h_list[5], out_w_list[5])

    def feature_pyramid(xyz_in, points_in, sel):                                                    # :126-139 / :151-164
        l0_selected_idx, l1_selected_idx, l2_selected_idx, l3_selected_idx = sel
        batch_size = xyz_in.shape[0]
        l0_points, l0_xyz_proj = down_conv(xyz_in, points_in, l0_selected_idx, K_sample=32, kernel_size=[9, 15],
                                           distance=Down_conv_dis[0], mlp=[8, 8, 16], mlp2=None, flag_add=False,
                                           is_training=is_training, bn_decay=bn_decay, scope='layer0')
        l0_points_proj = l0_points.reshape(batch_size, out_h_list[2], out_w_list[2], -1)
        l1_points, l1_xyz_proj = down_conv(l0_xyz_proj, l0_points_proj, l1_selected_idx, K_sample=32,
                                           kernel_size=[7, 11], distance=Down_conv_dis[1], mlp=[16, 16, 32], mlp2=None,
                                           flag_add=False, is_training=is_training, bn_decay=bn_decay, scope='layer1')
        l1_points_proj = l1_points.reshape(batch_size, out_h_list[3], out_w_list[3], -1)
        l2_points, l2_xyz_proj = down_conv(l1_xyz_proj, l1_points_proj, l2_selected_idx, K_sample=16,
                                           kernel_size=[5, 9], distance=Down_conv_dis[2], mlp=[32, 32, 64], mlp2=None,
                                           flag_add=False, is_training=is_training, bn_decay=bn_decay, scope='layer2')
        l2_points_proj = l2_points.reshape(batch_size, out_h_list[4], out_w_list[4], -1)
        if merge_branches(is_training, finest_points, batch_size * out_h_list[4] * out_w_list[4]):
            # inference: the layer-3 set-conv is only needed at the coarse pose (:187-194); it rides on the first launch of
            # the initial cost volume (:170), which only shares its inputs -- one launch and one serial stage less
            deferred.append(down_conv_job(l2_xyz_proj, l2_points_proj, l3_selected_idx, K_sample=16, kernel_size=[5, 9],
                                          distance=Down_conv_dis[3], mlp=[64, 64, 128], scope='layer3'))
            l3_points = l3_xyz_proj = None
        else:
            l3_points, l3_xyz_proj = down_conv(l2_xyz_proj, l2_points_proj, l3_selected_idx, K_sample=16,
                                               kernel_size=[5, 9], distance=Down_conv_dis[3], mlp=[64, 64, 128], mlp2=None,
                                               flag_add=False, is_training=is_training, bn_decay=bn_decay, scope='layer3')
        return ([l0_points, l1_points, l2_points, l3_points],
                [l0_points_proj, l1_points_proj, l2_points_proj, None],
                [l0_xyz_proj, l1_xyz_proj, l2_xyz_proj, l3_xyz_proj])

    deferred = []                      # (job, finish) of set-convs that wait for a cost-volume launch to ride on
    with tf_util.variable_scope('sa1') as scope:                                                    # :117
        both = _adjacent_frames(xyz_f1_input_proj, xyz_f2_input_proj) if not is_training else None
        siamese_train = (is_training and tuning.get("train_siamese_batch") and tuning.get("train_kernels")
                         and xyz_f1_input_proj.shape == xyz_f2_input_proj.shape)
        if siamese_train:
            # Training: the same ONE pass over the 2B batch; the batch-norm layers keep the frames' statistics apart (tf_util.bn_groups:
            # moments per frame, moving averages updated frame 1 then frame 2 -- the arithmetic of two calls with shared variables),
            # the gradients of the shared weights are the sums over both frames as before.  Half the encoder's launches.
            both_t = torch.cat([xyz_f1_input_proj, xyz_f2_input_proj], 0)
            sel2 = [get_selected_idx(both_t, stride_h_list[i], stride_w_list[i], out_h_list[i], out_w_list[i])
                    for i in (2, 3, 4, 5)]
            with tf_util.bn_groups(2):
                pts, pts_proj, xyz_proj = feature_pyramid(both_t, zero_features(both_t), sel2)
            B = batch_size
            two = lambda t: (None, None) if t is None else t.split(B)          # (split: ONE concatenation in backward, not two padded adds)
            pts_s = [two(p) for p in pts]
            pts_f1, pts_f2 = [a for a, _ in pts_s], [b for _, b in pts_s]
            shape_like = lambda t, ref: None if t is None or ref is None else t.reshape((B,) + tuple(ref.shape[1:]))
            pts_proj_f1 = [shape_like(a, ref) for a, ref in zip(pts_f1, pts_proj)]
            pts_proj_f2 = [shape_like(b, ref) for b, ref in zip(pts_f2, pts_proj)]
            xyz_s = [two(x) for x in xyz_proj]
            xyz_proj_f1, xyz_proj_f2 = [a for a, _ in xyz_s], [b for _, b in xyz_s]
        elif both is not None:
            # Siamese pyramid in ONE pass: the two frames share every weight (:143), and inference BN uses moving
            # statistics, so frame 2 is just batch elements B..2B-1 -- half the launches, identical numbers.
            sel2 = [get_selected_idx(both, stride_h_list[i], stride_w_list[i], out_h_list[i], out_w_list[i])
                    for i in (2, 3, 4, 5)]
            pts, pts_proj, xyz_proj = feature_pyramid(both, zero_features(both), sel2)
            B = batch_size
            half = lambda t, lo, hi: None if t is None else t[lo:hi]
            pts_f1, pts_f2 = [half(p, 0, B) for p in pts], [half(p, B, 2 * B) for p in pts]
            pts_proj_f1 = [half(p, 0, B) for p in pts_proj]
            pts_proj_f2 = [half(p, B, 2 * B) for p in pts_proj]
            xyz_proj_f1, xyz_proj_f2 = [half(x, 0, B) for x in xyz_proj], [half(x, B, 2 * B) for x in xyz_proj]
        else:
            sel1 = [l0_selected_idx, l1_selected_idx, l2_selected_idx, l3_selected_idx]
            points_input_proj = zero_features(xyz_f1_input_proj)
            pts_f1, pts_proj_f1, xyz_proj_f1 = feature_pyramid(xyz_f1_input_proj, points_input_proj, sel1)
            scope.reuse_variables()                                                                 # :143
            pts_f2, pts_proj_f2, xyz_proj_f2 = feature_pyramid(xyz_f2_input_proj, points_input_proj, sel1)

    # initial cost volume at l2 and the extra set-conv on it (:170-177)
    l2_points_f1_new = cost_volume(xyz_proj_f1[2], xyz_proj_f2[2], pts_proj_f1[2], pts_proj_f2[2],
                                   kernel_size1=[3, 5], kernel_size2=[5, 35], nsample=4, nsample_q=32,
                                   distance=Cost_volume_dis[2], mlp1=[128, 64, 64], mlp2=[128, 64],
                                   is_training=is_training, bn_decay=bn_decay, scope='flow_embedding_l2_origin',
                                   bn=True, pooling='max', knn=True, corr_func='concat',
                                   side_jobs=[job for job, _ in deferred] or None)
    if deferred:                                      # the layer-3 set-conv(s) that rode on the cost volume's first launch
        l2_points_f1_new, side = l2_points_f1_new
        for f, ((_job, finish), (out, new_xyz)) in enumerate(zip(deferred, side)):
            l3_points, l3_xyz_proj = finish(out, new_xyz)
            if len(deferred) == 1 and l3_points.shape[0] == 2 * batch_size:          # the 2B Siamese batch: frame 1 | frame 2
                pts_f1[3], pts_f2[3] = l3_points[:batch_size], l3_points[batch_size:]
                xyz_proj_f1[3], xyz_proj_f2[3] = l3_xyz_proj[:batch_size], l3_xyz_proj[batch_size:]
            else:
                (pts_f1 if f == 0 else pts_f2)[3] = l3_points
                (xyz_proj_f1 if f == 0 else xyz_proj_f2)[3] = l3_xyz_proj
    l2_points_new_proj_f1 = l2_points_f1_new.reshape(batch_size, out_h_list[4], out_w_list[4], -1)
    l3_points_f1_cost_volume, _ = down_conv(xyz_proj_f1[2], l2_points_new_proj_f1, l3_selected_idx, K_sample=16,
                                            kernel_size=[5, 9], distance=Down_conv_dis[3], mlp=[128, 64, 64], mlp2=None,
                                            flag_add=False, is_training=is_training, bn_decay=bn_decay,
                                            scope='new_layer3')

    # coarse pose at l3 (:183-208)
    l3_points_predict = l3_points_f1_cost_volume
    l3_points_predict_proj = l3_points_predict.reshape(batch_size, out_h_list[5], out_w_list[5], -1)
    l3_xyz_f1 = xyz_proj_f1[3].reshape(batch_size, -1, 3)
    # inference: the pose head of level L+1 also clears the projection buffers of level L's warp (one launch less)
    inference = not is_training

    def projection_buffers(level, g):
        if not inference:
            return None
        return _ops.ProjectionBuffers(batch_size, xyz_proj_f1[level].shape[1] * xyz_proj_f1[level].shape[2],
                                      out_h_list[g], out_w_list[g], pts_f1[level].shape[-1], dev, pts_f1[level].dtype)
    def sv_ride(xyz_bn3, feature=None):
        # softmax_valid's partial sums ride on the launch that produces the pose head's inputs (tuning sv_ride; one launch less
        # per level where that launch is a tile kernel: fused.mlp / fused.mlp2_pair decide)
        if inference and fused_pairs_available(is_training) and tuning.get("sv_ride"):
            return _ops.SvPartials(xyz_bn3, feature)
        return None
    def next_warp(level):          # the cloud + features the NEXT warp moves: run by the pose head's own launches
        return (xyz_proj_f1[level].reshape(batch_size, -1, 3), pts_f1[level]) if inference else None
    next_buffers = projection_buffers(2, 4)
    sv3 = sv_ride(l3_xyz_f1, l3_points_predict)
    l3_cost_volume_w = flow_predictor(pts_f1[3], None, l3_points_predict, mlp=[128, 64], is_training=is_training,
                                      bn_decay=bn_decay, scope='l3_costvolume_predict_ww',
                                      clear=next_buffers if sv3 is not None else None,
                                      **({"sv": sv3} if sv3 is not None else {}))
    l3_cost_volume_w_proj = l3_cost_volume_w.reshape(batch_size, out_h_list[5], out_w_list[5], -1)
    l3_q, l3_t, l3_q_norm = _estimate_pose(l3_points_predict, l3_cost_volume_w, l3_xyz_f1, 3, None, None,
                                           is_training, clear=next_buffers, warp=next_warp(2), partials=sv3)             # :194-208

    # three warp-refinement levels (:211-425); one loop instead of three pasted blocks
    cv_kernel2 = {2: [5, 15], 1: [7, 25], 0: [11, 41]}                                              # :243,:317,:391
    poses = {3: (l3_q_norm, l3_t)}
    q_prev, t_prev = l3_q, l3_t
    coarse_w_proj, coarse_predict_proj, coarse_xyz_proj = l3_cost_volume_w_proj, l3_points_predict_proj, xyz_proj_f1[3]
    for level, g in ((2, 4), (1, 3), (0, 2)):
        q_coarse = q_prev.reshape(batch_size, 1, -1)                                                # :211-212
        t_coarse = t_prev.reshape(batch_size, 1, -1)
        xyz_f1 = xyz_proj_f1[level].reshape(batch_size, -1, 3)
        # warp by the coarse pose, zero invalid points, re-project with the level's features (:217-236)
        _warped, xyz_warp_proj_f1, points_warp_proj_f1 = warp_and_project(
            xyz_f1, pts_f1[level], q_coarse, t_coarse, out_h_list[g], out_w_list[g], next_buffers)
        if PROJECTION_TAP is not None and next_buffers is not None:      # (parity tests: the cell every warped point landed in)
            PROJECTION_TAP.append((level, next_buffers.scratch, batch_size, xyz_f1.shape[1], out_h_list[g], out_w_list[g], xyz_warp_proj_f1, _warped))
        next_buffers = projection_buffers(level - 1, g - 1) if level > 0 else None
        xyz_warp_f1 = xyz_warp_proj_f1.reshape(batch_size, -1, 3)
        points_warp_f1 = points_warp_proj_f1.reshape(batch_size, out_h_list[g] * out_w_list[g], -1)

        def branch_cost(side_jobs=None, side_chain=False):
            return cost_volume(xyz_warp_proj_f1, xyz_proj_f2[level], points_warp_proj_f1, pts_proj_f2[level],
                               kernel_size1=[3, 5], kernel_size2=cv_kernel2[level], nsample=4, nsample_q=6,
                               distance=Cost_volume_dis[level], mlp1=[128, 64, 64], mlp2=[128, 64],
                               is_training=is_training, bn_decay=bn_decay, scope='flow_embedding_l%d' % level,
                               bn=True, pooling='max', knn=True, corr_func='concat', side_jobs=side_jobs,
                               **({"side_chain": True} if side_chain else {}))                             # :242

        def branch_up(kind, coarse_feat_proj):
            return up_conv(xyz_warp_proj_f1, coarse_xyz_proj, points_warp_proj_f1, coarse_feat_proj,
                           kernel_size=[7, 15], stride_h=stride_h_list[g + 1], stride_w=stride_w_list[g + 1],
                           nsample=8, distance=Up_conv_dis[level], mlp=[128, 64], mlp2=[128, 64],
                           scope='up_sa_layer_layer_l%d%s' % (level, kind), is_training=is_training,
                           bn_decay=bn_decay, knn=True)                                             # :247, :250

        if fused_pairs_available(is_training):
            # inference: the embedding / embedding-mask twins of a level run as PAIRED launches (same shapes,
            # different weights): 2 + 1 launches instead of 4 + 2
            up = dict(xyz1_proj=xyz_warp_proj_f1, xyz2_proj=coarse_xyz_proj, feat1_proj=points_warp_proj_f1,
                      kernel_size=[7, 15], stride_h=stride_h_list[g + 1], stride_w=stride_w_list[g + 1], nsample=8,
                      distance=Up_conv_dis[level], mlp=[128, 64], mlp2=[128, 64])
            up_w = dict(up, feat2_proj=coarse_w_proj, scope='up_sa_layer_layer_l%dw' % level)
            up_c = dict(up, feat2_proj=coarse_predict_proj, scope='up_sa_layer_layer_l%dcostvolume' % level)
            # the cost volume and stage 1 of the two set-upconvs only share inputs (:242-250): while the GPU is underfilled
            # ONE launch runs cost-volume stage 1 and both set-conv jobs (both branches in flight together, one launch
            # boundary less); a full GPU takes them as two launches
            jobs = up_conv_stage1_jobs(up_w, up_c)
            if merge_branches(is_training, finest_points, batch_size * out_h_list[g] * out_w_list[g]):
                cost, ((up_w_pooled, _), (up_c_pooled, _)) = branch_cost(side_jobs=jobs)
            else:
                # ... and where both are register-resident chain launches (a larger level below the throughput batch), the
                # chain kernels' heterogeneous launch (tuning chain_pair); the select-k pre-pass stays a launch of its own
                # (only in the forwards small enough to merge at all: from batch 4 on the GPU is full and the two launches are faster,
                #  batch 4 23.3 -> 21.9 k pairs/s, batch 8 fp16 28.0 -> 27.4 k: profiles/r05_batch1_regimes.txt)
                pair = tuning.get("chain_pair") and finest_points <= tuning.get("merge_points")
                cost, sides = branch_cost(side_jobs=jobs, side_chain=True) if pair else (branch_cost(), None)
                (up_w_pooled, _), (up_c_pooled, _) = sides if sides is not None else fused.setconv_pair(jobs[0], jobs[1])
            fp = dict(points_f1=points_warp_f1, cost_volume=cost, mlp=[128, 64])
            # set-upconv stage 2 and the predictor it feeds share a launch
            sv = sv_ride(xyz_warp_f1)
            w_up_sample, weight, cost_up_sample, predict = up_conv_predict_finish(
                up_w, up_c, dict(fp, scope='l%d_w_predict' % level), dict(fp, scope='l%d_costvolume_predict' % level),
                up_w_pooled, up_c_pooled, clear=next_buffers if sv is not None else None, sv=sv)
        else:
            # the cost volume and the two set-upconvs only share inputs: optional concurrent branches
            cost, w_up_sample, cost_up_sample = _parallel([branch_cost, lambda: branch_up('w', coarse_w_proj),
                                                           lambda: branch_up('costvolume', coarse_predict_proj)], is_training)
            sv = None
            predict, weight = _parallel([
                lambda: flow_predictor(points_warp_f1, cost_up_sample, cost, mlp=[128, 64], is_training=is_training,
                                       bn_decay=bn_decay, scope='l%d_costvolume_predict' % level),     # :253
                lambda: flow_predictor(points_warp_f1, w_up_sample, cost, mlp=[128, 64], is_training=is_training,
                                       bn_decay=bn_decay, scope='l%d_w_predict' % level)], is_training)  # :254
        q_prev, t_prev, q_norm = _estimate_pose(predict, weight, xyz_warp_f1, level, q_coarse, t_coarse,
                                                is_training, pose_out if level == 0 else None,
                                                clear=next_buffers,
                                                warp=next_warp(level - 1) if level > 0 else None, partials=sv)  # :262-280
        poses[level] = (q_norm, t_prev)

        coarse_w_proj = weight.reshape(batch_size, out_h_list[g], out_w_list[g], -1)                # :256-257
        coarse_predict_proj = predict.reshape(batch_size, out_h_list[g], out_w_list[g], -1)
        coarse_xyz_proj = xyz_warp_proj_f1

    l0_xyz_f1 = xyz_proj_f1[0].reshape(batch_size, -1, 3)
    return (poses[0][0], poses[0][1], poses[1][0], poses[1][1], poses[2][0], poses[2][1], poses[3][0], poses[3][1],
            l0_xyz_f1)                                                                              # :427-433


def get_model(point_cloud, H_input, W_input, T_gt, T_trans, T_trans_inv, is_training, bn_decay=None, aug_frame=None):
    """pwclo_model.py:30-433 with the reference's signature: point_cloud (B, 2*N, >=3), three (B,4,4).
    Returns the reference's 11-tuple."""
    batch_size = point_cloud.shape[0]
    if aug_frame is None:
        aug_frame = np.random.choice([1, 2], size=batch_size, replace=True)                         # :59
    with torch.no_grad():                                                                           # tf.stop_gradient, :66-67
        # PreProcess's crop + augmentation and both ProjectPC2SphericalRing calls in one C-ABI call (three launches)
        _points, both = input_stage(point_cloud, T_trans, aug_frame, H_input, W_input)
        xyz_f1_proj, xyz_f2_proj = both[:batch_size], both[batch_size:]         # adjacent: one 2B Siamese batch
        q_gt, t_gt = preprocess_gt(T_gt, T_trans, T_trans_inv, aug_frame)
    out = get_model_from_projection(xyz_f1_proj, xyz_f2_proj, is_training, bn_decay)
    return out + (q_gt, t_gt)


def get_loss(l0_q, l0_t, l1_q, l1_t, l2_q, l2_t, l3_q, l3_t, q_gt, t_gt, w_x, w_q):
    """pwclo_model.py:437-481."""
    if l0_q.is_cuda and w_x.is_cuda and w_x.dim() == 0:
        return _ops.pose_loss(l0_q, l0_t, l1_q, l1_t, l2_q, l2_t, l3_q, l3_t, q_gt, t_gt, w_x, w_q)   # one launch each way
    t_gt = t_gt.squeeze(-1)

    def level(q, t):
        q_norm = _normalise_q(q)
        loss_q = torch.sqrt(((q_gt - q_norm) * (q_gt - q_norm)).sum(-1, keepdim=True) + 1e-10).mean()
        loss_x = torch.sqrt((t - t_gt) * (t - t_gt) + 1e-10).mean()
        return loss_x * torch.exp(-w_x) + w_x + loss_q * torch.exp(-w_q) + w_q

    return 1.6 * level(l3_q, l3_t) + 0.8 * level(l2_q, l2_t) + 0.4 * level(l1_q, l1_t) + 0.2 * level(l0_q, l0_t)
