"""GPU parity of the operator library (set-conv, cost volume, set-upconv, flow predictor,
softmax_valid, quaternion warp + re-projection) against the numpy restatement in oracle/ops_np.py.
Tolerance (north_star): 1e-4 for fp32 features; neighbour indices are compared bit-exact elsewhere."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import load_pkg
from oracle import ops_np as O
from util_params import close, export, randomise, shuffle_fn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True, params=["fused", "per-operator"])
def kernel_path(request):
    """Every parity test runs on both inference paths: the fused MFMA kernels (csrc/elo_fused.hip) and the
    per-operator kernels + hipBLASLt GEMMs (csrc/elo_features.hip)."""
    pu = load_pkg("pointnet_util")
    pu.use_fused(request.param == "fused")
    yield request.param
    pu.use_fused(True)


def _ctx():
    tf_util, perm = load_pkg("tf_util"), load_pkg("perm")
    store = tf_util.VariableStore(DEV, seed=1)
    return tf_util, perm, store, perm.PermSource(fn=shuffle_fn)


def _run(fn, store, perms):
    """Run once to create variables, randomise them, run again for the checked result."""
    tf_util, perm = load_pkg("tf_util"), load_pkg("perm")
    with tf_util.default_store(store), perm.default_perm_source(perms), torch.no_grad():
        fn()
        randomise(store, seed=3)
        out = fn()
    torch.cuda.synchronize()
    return out


def _scene(B, H, W, seed, C=None):
    synth = load_pkg("synth")
    f1, f2 = synth.frame_pair(B, H, W, seed=seed)
    rng = np.random.default_rng(seed)
    feats = None
    if C:
        feats = [rng.normal(0, 1, (B, H, W, C)).astype(np.float32) for _ in range(2)]
    return f1, f2, feats


t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("B,H,W,C,K,ks,dist,mlp", [(1, 16, 225, 16, 32, [7, 11], 3.0, [16, 16, 32]),
                                                   (2, 8, 113, 32, 16, [5, 9], 6.0, [32, 32, 64]),
                                                   (1, 64, 450, 3, 32, [9, 15], 0.5, [8, 8, 16])])
def test_down_conv(B, H, W, C, K, ks, dist, mlp):
    pu, mu = load_pkg("pointnet_util"), load_pkg("model_util")
    _, _, store, perms = _ctx()
    f1, _, feats = _scene(B, H, W, 11, C)
    oh, ow = (H + 1) // 2, (W + 1) // 2
    sel_np = O.get_selected_idx(B, 2, 2, oh, ow)
    xyz, pts = t(f1), t(feats[0])
    sel = mu.get_selected_idx(xyz, 2, 2, oh, ow)
    assert np.array_equal(sel.cpu().numpy(), sel_np)
    got = _run(lambda: pu.down_conv(xyz, pts, sel, K_sample=K, kernel_size=ks, distance=dist, mlp=mlp, mlp2=None,
                                    flag_add=False, is_training=False, bn_decay=None, scope='layerX'), store, perms)
    want = O.down_conv(export(store), shuffle_fn, f1, feats[0], sel_np, K, ks, dist, mlp, "layerX")
    close(got[0], want[0])
    close(got[1], want[1], atol=0, rtol=0)


@pytest.mark.parametrize("B,H,W,C,Kq,ks2,dist", [(1, 4, 57, 64, 32, [5, 35], 4.0), (2, 8, 113, 32, 6, [7, 25], 2.0),
                                                 (1, 16, 225, 16, 6, [11, 41], 1.0)])
def test_cost_volume(B, H, W, C, Kq, ks2, dist):
    pu = load_pkg("pointnet_util")
    _, _, store, perms = _ctx()
    f1, f2, feats = _scene(B, H, W, 21, C)
    a = [t(x) for x in (f1, f2, feats[0], feats[1])]
    got = _run(lambda: pu.cost_volume(a[0], a[1], a[2], a[3], kernel_size1=[3, 5], kernel_size2=ks2, nsample=4,
                                      nsample_q=Kq, distance=dist, mlp1=[128, 64, 64], mlp2=[128, 64],
                                      is_training=False, bn_decay=None, scope='flow_embedding_t', bn=True,
                                      pooling='max', knn=True, corr_func='concat'), store, perms)
    want = O.cost_volume(export(store), shuffle_fn, f1, f2, feats[0], feats[1], [3, 5], ks2, 4, Kq, dist,
                         [128, 64, 64], [128, 64], "flow_embedding_t")
    close(got, want)


def test_cost_volume_all_masked_is_uniform_softmax():
    """Every neighbour masked (frame 2 empty): softmax over K of identical -1e10 logits is uniform 1/K
     -- must not produce NaN."""
    pu = load_pkg("pointnet_util")
    _, _, store, perms = _ctx()
    f1, _, feats = _scene(1, 4, 57, 5, 8)
    f2 = np.zeros_like(f1)
    a = [t(x) for x in (f1, f2, feats[0], feats[1])]
    got = _run(lambda: pu.cost_volume(a[0], a[1], a[2], a[3], kernel_size1=[3, 5], kernel_size2=[5, 15], nsample=4,
                                      nsample_q=6, distance=4.0, mlp1=[128, 64, 64], mlp2=[128, 64],
                                      is_training=False, bn_decay=None, scope='cv_empty'), store, perms)
    want = O.cost_volume(export(store), shuffle_fn, f1, f2, feats[0], feats[1], [3, 5], [5, 15], 4, 6, 4.0,
                         [128, 64, 64], [128, 64], "cv_empty")
    assert torch.isfinite(got).all()
    close(got, want)


@pytest.mark.parametrize("B,H,W,sh,sw,C1,dist", [(1, 4, 57, 1, 2, 64, 9.0), (2, 8, 113, 2, 2, 32, 6.0),
                                                 (1, 16, 225, 2, 2, 16, 3.0)])
def test_up_conv(B, H, W, sh, sw, C1, dist):
    pu = load_pkg("pointnet_util")
    _, _, store, perms = _ctx()
    f1, _, feats = _scene(B, H, W, 31, C1)
    H2, W2 = -(-H // sh), -(-W // sw)
    sparse_xyz = np.ascontiguousarray(f1[:, ::sh, ::sw][:, :H2, :W2])
    sparse_feat = np.random.default_rng(2).normal(0, 1, (B, H2, W2, 64)).astype(np.float32)
    a = [t(x) for x in (f1, sparse_xyz, feats[0], sparse_feat)]
    got = _run(lambda: pu.up_conv(a[0], a[1], a[2], a[3], kernel_size=[7, 15], stride_h=sh, stride_w=sw, nsample=8,
                                  distance=dist, mlp=[128, 64], mlp2=[128, 64], scope='up_t', is_training=False,
                                  bn_decay=None, knn=True), store, perms)
    want = O.up_conv(export(store), shuffle_fn, f1, sparse_xyz, feats[0], sparse_feat, [7, 15], sh, sw, 8, dist,
                     [128, 64], [128, 64], "up_t")
    close(got, want)


def test_flow_predictor_none_combinations():
    pu = load_pkg("pointnet_util")
    rng = np.random.default_rng(0)
    a, b, c = (rng.normal(0, 1, (2, 116, n)).astype(np.float32) for n in (128, 64, 64))
    for name, args in (("fp_ac", (a, None, c)), ("fp_ab", (a, b, None)), ("fp_abc", (a, b, c))):
        _, _, store, perms = _ctx()
        ta = [None if x is None else t(x) for x in args]
        got = _run(lambda: pu.flow_predictor(ta[0], ta[1], ta[2], mlp=[128, 64], is_training=False, bn_decay=None,
                                             scope=name), store, perms)
        want = O.flow_predictor(export(store), args[0], args[1], args[2], [128, 64], name)
        close(got, want)


def test_softmax_valid():
    mu = load_pkg("model_util")
    rng = np.random.default_rng(4)
    B, N, C = 3, 904, 64
    f = rng.normal(0, 1, (B, N, C)).astype(np.float32)
    w = rng.normal(0, 3, (B, N, C)).astype(np.float32)
    xyz = rng.normal(0, 5, (B, N, 3)).astype(np.float32)
    xyz[rng.random((B, N)) < 0.3] = 0
    xyz[2] = 0                                            # a batch element without any valid point -> zeros
    valid = ~np.all(xyz == 0, -1)
    want = O.softmax_valid(f, w, valid)
    close(mu.softmax_valid(t(f), t(w), t(xyz)), want, atol=1e-5)
    close(mu.softmax_valid(t(f), t(w), torch.from_numpy(valid).to(DEV)), want, atol=1e-5)


def test_quaternion_ops_and_pose_composition():
    mu = load_pkg("model_util")
    rng = np.random.default_rng(6)
    B, N = 2, 50
    q = rng.normal(0, 1, (B, 1, 4)).astype(np.float32)
    p = rng.normal(0, 1, (B, N, 4)).astype(np.float32)
    close(mu.mul_q_point(t(q), t(p), B), O.mul_q_point(q, p, B), atol=1e-6)
    close(mu.mul_point_q(t(p), t(q), B), O.mul_point_q(p, q, B), atol=1e-6)
    close(mu.inv_q(t(q), B), O.inv_q(q, B), atol=1e-6)


def _boundary_safe_points(rng, B, N, H, W):
    """Points whose (row, col) sit well inside their cell, so 1-ulp differences between libm
    implementations of atan2/asin cannot move them to a neighbouring cell."""
    az_res, vres, voff = (float(x) for x in O.projection_constants(H, W))
    col = rng.integers(0, W, (B, N)) + rng.uniform(0.2, 0.8, (B, N))
    rowf = rng.integers(1, H, (B, N)) + rng.uniform(0.2, 0.8, (B, N))       # tmp_int in [1, H): row = H - int(.)
    az = np.pi - col * az_res
    beta = (rowf - voff) * vres
    r = rng.uniform(3, 30, (B, N))
    return np.stack([r * np.cos(beta) * np.cos(az), r * np.cos(beta) * np.sin(az), r * np.sin(beta)], -1).astype(np.float32)


@pytest.mark.parametrize("H,W,N,C", [(4, 57, 228, 64), (16, 225, 3600, 16), (64, 1800, 150000, 0)])
def test_project_spherical_ring(H, W, N, C):
    mu = load_pkg("model_util")
    rng = np.random.default_rng(8)
    B = 2
    pc = _boundary_safe_points(rng, B, N, H, W)
    pc[rng.random((B, N)) < 0.1] = 0                       # empty points all land in one cell and blank it
    feat = rng.normal(0, 1, (B, N, C)).astype(np.float32) if C else None
    got = mu.ProjectPC2SphericalRing(t(pc), t(feat) if C else None, H, W)
    want = O.ProjectPC2SphericalRing(pc, feat, H, W)
    close(got[0], want[0], atol=1e-5)
    if C:
        close(got[1], want[1], atol=1e-5)


def test_projection_cells_differ_from_the_oracle_only_on_cell_borders():
    """The other projection tests keep their points off the cell borders.  Here 150 000 points are drawn with NO such
    care: the kernel's per-point cell (read from the call's scratch, include/elo.h) equals the oracle's for all but a
    handful, and every one of those lies on a border -- its continuous column or row coordinate, recomputed in float64, is
    within 1e-4 of an integer, where one ulp of atan2f / asinf / the division decides the cell (model_util.py:229-242)."""
    ops = load_pkg("_ops")
    rng = np.random.default_rng(21)
    B, N, H, W = 1, 150000, 64, 1800
    az = rng.uniform(-np.pi, np.pi, (B, N))
    beta = np.deg2rad(rng.uniform(-26.0, 3.0, (B, N)))          # a little beyond the sensor's -24.8 .. 2 degrees
    r = rng.uniform(2, 60, (B, N))
    pc = np.stack([r * np.cos(beta) * np.cos(az), r * np.cos(beta) * np.sin(az), r * np.sin(beta)], -1).astype(np.float32)
    buf = ops.ProjectionBuffers(B, N, H, W, 0, DEV)
    ops.warp_project(t(pc), None, None, None, H, W, buffers=buf)
    torch.cuda.synchronize()
    cells = B * H * W
    got = buf.scratch[cells + 4 * B: cells + 4 * B + B * N].cpu().numpy().reshape(B, N)        # cell_of per point
    az_res, vres, voff = (np.float64(x) for x in O.projection_constants(H, W))
    x, y, z = (pc[..., i].astype(np.float64) for i in range(3))
    colf = (np.pi - np.arctan2(y, x)) / az_res
    rowf = np.arcsin(z / np.sqrt(x * x + y * y + z * z)) / vres + voff
    # the oracle's cells, from its own fp32 evaluation
    F = np.float32
    xs, ys, zs = pc[..., 0], pc[..., 1], pc[..., 2]
    rr = np.sqrt((pc * pc).sum(-1)).astype(F)
    ocol = np.trunc(((F(np.pi) - np.arctan2(ys, xs).astype(F)) / F(az_res)).astype(F)).astype(np.int64)
    orow = H - np.trunc((np.arcsin((zs / rr).astype(F)).astype(F) / F(vres) + F(voff)).astype(F)).astype(np.int64)
    want = np.clip(orow, 0, H - 1) * W + np.clip(ocol, 0, W - 1)
    differ = got != want
    assert differ.mean() < 2e-4, differ.mean()                   # measured: a few points in 150 000
    near = lambda v: np.abs(v - np.round(v))
    on_border = (near(colf) < 1e-4) | (near(rowf) < 1e-4)
    assert on_border[differ].all()
    # ... and away from the borders the agreement is exact
    assert (got[~on_border] == want[~on_border]).all()


def test_warp_and_project():
    mu = load_pkg("model_util")
    rng = np.random.default_rng(9)
    B, H, W, C = 2, 8, 113, 32
    N = H * W
    # warp with a small rotation; then nudge points off cell borders by testing only cells that agree
    pc = _boundary_safe_points(rng, B, N, H, W)
    pc[rng.random((B, N)) < 0.1] = 0
    feat = rng.normal(0, 1, (B, N, C)).astype(np.float32)
    q = np.array([[[1.0, 0.002, -0.001, 0.004]], [[0.999, -0.003, 0.002, 0.01]]], np.float32)
    tt = np.array([[[0.5, 0.05, -0.02]], [[-0.3, 0.1, 0.01]]], np.float32)
    warped, xyz_proj, feat_proj = mu.warp_and_project(t(pc), t(feat), t(q), t(tt), H, W)
    want_warped = O.warp(pc, q, tt)
    close(warped, want_warped, atol=1e-5)
    # project the GPU's own warped points with the oracle: isolates the projection from 1-ulp warp noise
    w_xyz, w_feat = O.ProjectPC2SphericalRing(warped.cpu().numpy(), feat, H, W)
    got_xyz, got_feat = xyz_proj.cpu().numpy(), feat_proj.cpu().numpy()
    same = np.isclose(got_xyz, w_xyz, atol=1e-5).all(-1)
    assert same.mean() > 0.995, same.mean()                # cell flips only for points on a cell border
    assert np.isclose(got_feat[same], w_feat[same], atol=1e-5).all()


def test_pose_head_clears_the_next_projection():
    """elo_pose_head_args.clear_* / elo_warp_project_args.prepared: buffers cleared on the side by the pose head's
    partial kernel give the same projection as the self-initialising call, even when they held garbage."""
    ops = load_pkg("_ops")
    rng = np.random.default_rng(12)
    B, H, W, C = 2, 8, 113, 32
    N = H * W
    pc = _boundary_safe_points(rng, B, N, H, W)
    pc[rng.random((B, N)) < 0.1] = 0
    feat = t(rng.normal(0, 1, (B, N, C)).astype(np.float32))
    f, w = (t(rng.normal(0, 1, (B, 116, 64)).astype(np.float32)) for _ in range(2))
    xyz_small = t(rng.normal(0, 5, (B, 116, 3)).astype(np.float32))
    Wb, bb = t(rng.normal(0, .1, (64, 256)).astype(np.float32)), t(rng.normal(0, .1, (256,)).astype(np.float32))
    Wq, bq = t(rng.normal(0, .1, (256, 4)).astype(np.float32)), t(np.array([1, 0, 0, 0], np.float32))
    Wt, bt = t(rng.normal(0, .1, (256, 3)).astype(np.float32)), t(np.zeros(3, np.float32))
    plain = ops.pose_head(f, w, xyz_small, Wb, bb, Wq, bq, Wt, bt)
    want = ops.warp_project(t(pc), feat, plain[0], plain[1], H, W)
    buf = ops.ProjectionBuffers(B, N, H, W, C, DEV)
    buf.out_xyz.fill_(float("nan")); buf.out_feat.fill_(7.0); buf.scratch.fill_(-1)
    q, tt, qn = ops.pose_head(f, w, xyz_small, Wb, bb, Wq, bq, Wt, bt, clear=buf)
    assert buf.cleared and all(torch.equal(a, b) for a, b in zip((q, tt, qn), plain))
    got = ops.warp_project(t(pc), feat, q, tt, H, W, buffers=buf)
    assert not buf.cleared
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    # the zero points all tie at range 0 in one cell and are SUMMED there (scatter_nd adds duplicates): atomic order
    assert torch.allclose(got[2], want[2], atol=1e-4)
    with pytest.raises(ValueError, match="ProjectionBuffers"):
        ops.warp_project(t(pc), feat, q, tt, H, W + 1, buffers=buf)
    # elo_pose_head_warp: the pose head runs the projection itself (every block recomputes the head, warps its points)
    buf2 = ops.ProjectionBuffers(B, N, H, W, C, DEV)
    buf2.out_xyz.fill_(float("nan")); buf2.out_feat.fill_(-3.0); buf2.scratch.fill_(5)
    q2, t2, qn2 = ops.pose_head(f, w, xyz_small, Wb, bb, Wq, bq, Wt, bt, clear=buf2, warp=(t(pc), feat))
    assert all(torch.equal(a, b) for a, b in zip((q2, t2, qn2), plain)) and buf2.result is not None
    got2 = ops.warp_project(t(pc), feat, q2, t2, H, W, buffers=buf2)           # hands the stored result over
    assert buf2.result is None
    assert torch.equal(got2[0], want[0]) and torch.equal(got2[1], want[1]) and torch.allclose(got2[2], want[2], atol=1e-4)
    with pytest.raises(ValueError, match="needs clear="):
        ops.pose_head(f, w, xyz_small, Wb, bb, Wq, bq, Wt, bt, warp=(t(pc), feat))


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 113, 16), (1, 16, 225, 32)])
def test_cost_volume_kernels_with_fp16_storage(B, H, W, C):
    """BASELINE configs[2]: the three cost-volume kernels with fp16 feature storage (ELO_F16: fp16 in HBM, fp32
    arithmetic).  On fp16-representable inputs they must equal the fp32 kernels' outputs rounded to fp16 -- copies and
    masking are exact, the geometry runs the same fp32 instructions; the softmax-pool agrees to fp16 rounding -- and the
    fp32 kernels are the ones the oracle pins (test_cost_volume)."""
    ops, elo, synth = load_pkg("_ops"), load_pkg(), load_pkg("synth")
    rng = np.random.default_rng(31)
    N, Kq, Kp = H * W, 6, 4
    f1, f2 = synth.frame_pair(B, H, W, seed=17)
    x1, x2 = t(f1), t(f2)
    hw = t(synth.hw_index(B, H, W))
    order_q, order_p = t(rng.permutation(5 * 15).astype(np.int32)), t(rng.permutation(3 * 5).astype(np.int32))
    idx_q, _, _, m_q = elo.fused_conv_select_k(x1, x2, hw, order_q, H, W, N, 5, 15, Kq, 0, 6.0, 1, 1, want_valid=False)
    idx_p, _, _, m_p = elo.fused_conv_random_k(x1, x1, hw, order_p, H, W, N, 3, 5, Kp, 0, 6.0, 1, 1, want_valid=False)
    m_q, m_p = m_q.reshape(B, N, Kq), m_p.reshape(B, N, Kp)
    h = lambda *shape: t(rng.normal(0, 1, shape).astype(np.float32)).half()
    ft1, ft2, cost = h(B, H, W, C), h(B, H, W, C), h(B, H, W, 64)

    e1_16 = ops.cv_encode1(x1.reshape(B, N, 3), ft1.reshape(B, N, C), x2, ft2, idx_q, m_q)
    e1_32 = ops.cv_encode1(x1.reshape(B, N, 3), ft1.float().reshape(B, N, C), x2, ft2.float(), idx_q, m_q)
    assert e1_16.dtype == torch.float16 and e1_16.shape == e1_32.shape and torch.equal(e1_16, e1_32.half())

    g16, r16 = ops.cv_encode2(x1, ft1, cost, idx_p, m_p)
    g32, r32 = ops.cv_encode2(x1, ft1.float(), cost.float(), idx_p, m_p)
    assert g16.dtype == r16.dtype == torch.float16
    assert torch.equal(g16, g32.half()) and torch.equal(r16, r32.half())

    for K, mask in ((Kq, m_q), (Kp, m_p)):
        lg, vl = h(B, N, K, 64) * 3, h(B, N, K, 64)
        p16 = ops.masked_softmax_pool(lg, vl, mask)
        p32 = ops.masked_softmax_pool(lg.float(), vl.float(), mask)
        # fp32 arithmetic on the same (fp16-representable) inputs: the fp16 result is the fp32 one rounded once -- up to the ORDER of
        # the K additions, which follows the storage type's lane layout in the wave-per-point kernel (4 lane groups in fp32, 8 in fp16:
        # round 6): the two fp32 sums differ in their last bit now and then, and now and then that decides an fp16 rounding
        assert p16.dtype == torch.float16
        same = p16 == p32.half()
        assert float(same.float().mean()) > 0.995
        assert ((p16.float() - p32).abs() <= 2.0 ** -10 * p32.abs() + 1e-7).all()            # never more than one fp16 ulp apart
        with load_pkg("tuning").override(pool_wave=0):                                        # the quarter-wave form: one layout, equal bits
            assert torch.equal(ops.masked_softmax_pool(lg, vl, mask), ops.masked_softmax_pool(lg.float(), vl.float(), mask).half())
        wide = h(B, N, K, 128)                               # values as a channel slice of a wider tensor (stride 128)
        assert torch.allclose(ops.masked_softmax_pool(lg, wide[..., :64], mask).float(),
                              ops.masked_softmax_pool(lg.float(), wide[..., :64].float().contiguous(), mask),
                              rtol=2e-3, atol=1e-3)
    with pytest.raises(TypeError, match="all float32 or all float16"):
        ops.cv_encode1(x1.reshape(B, N, 3), ft1.reshape(B, N, C), x2, ft2.float(), idx_q, m_q)
    with pytest.raises(RuntimeError, match="fp16 needs C and Cc multiples of 8"):
        ops.cv_encode2(x1, ft1[..., :4].contiguous(), cost, idx_p, m_p)


@pytest.mark.parametrize("rows,w_src,w_before,w_after", [(904, (64, 32), 32, 64), (3600, (64, 16), 16, 64),
                                                         (45, (64, 6), 10, 0), (17, (20,), 0, 7)])
def test_two_stage_mlp_equals_two_launches(rows, w_src, w_before, w_after):
    """elo_mlp_args second stage (set-upconv stage 2 -> flow predictor in one launch): `out` equals the separate launch
    bit for bit (same layers, same summation order); `out2` equals the separate launch over concat[before, out, after]
    to fp32 rounding (the fused kernel keeps its columns as [out | before | after] with the weight rows permuted to
    match: a different summation order) -- for aligned and unaligned widths, ragged last tile, paired jobs."""
    fused, tf_util = load_pkg("fused"), load_pkg("tf_util")
    rng = np.random.default_rng(rows)
    store = tf_util.VariableStore(DEV, seed=rows)
    r = lambda *s: t(rng.normal(0, 1, s).astype(np.float32))

    def job(tag):
        with tf_util.default_store(store), torch.no_grad(), tf_util.variable_scope("two_stage_%s" % tag):
            P = fused.packed_layer
            k1 = sum(w_src)
            layers = [P("a0", k1, 128), P("a1", 128, 64)]
            k2 = w_before + 64 + w_after
            layers2 = [P("b0", k2, 128, row_order=fused.stage2_row_order(w_before, 64, w_after)), P("b1", 128, 64)]
            plain2 = [P("b0", k2, 128), layers2[1]]                  # the same weights in the reference's concat order
        return dict(sources=[r(rows, w) for w in w_src], layers=layers, before=r(rows, w_before) if w_before else None,
                    after=r(rows, w_after) if w_after else None, layers2=layers2), plain2
    (ja, pa), (jb, pb) = job("a"), job("b")
    (o1a, o2a), (o1b, o2b) = fused.mlp2_pair(ja, jb)
    for j, plain2, o1, o2 in ((ja, pa, o1a, o2a), (jb, pb, o1b, o2b)):
        want1 = fused.mlp(j["sources"], j["layers"])
        parts = [p for p in (j["before"], want1, j["after"]) if p is not None]
        want2 = fused.mlp(parts, plain2)
        assert torch.equal(o1, want1)
        assert torch.allclose(o2, want2, rtol=1e-5, atol=1e-5)


@pytest.fixture(params=["split", "half"])
def chain_products(request):
    """The register-resident kernels exist for both products modes (fp32-class hi / lo split, and ONE fp16 product per
    block: `fused.products("half")`): the tests below run under each."""
    fused = load_pkg("fused")
    if fused.fp32_mfma():
        pytest.skip("the fp32-MFMA comparison build has no register-resident kernels")
    with fused.products(request.param):
        yield request.param


@pytest.mark.parametrize("C,dt,rows,pair", [(16, "f32", 700, True), (16, "f16", 1000, True), (32, "f32", 257, False), (64, "f16", 384, True)])
def test_register_resident_two_stage_mlp_equals_the_tile_kernel(C, dt, rows, pair, monkeypatch, chain_products):
    """mlp2_rr_kernel (set-upconv stage 2 + flow predictor on the register-resident chain, taken from ELO_MLP_RR_ROWS rows
    on) gives mlp_kernel's outputs bit for bit -- both stages, fp32 and fp16 storage, ragged last tile, single and paired."""
    fused, tf_util, lib = load_pkg("fused"), load_pkg("tf_util"), load_pkg("_lib")
    if fused.fp32_mfma():
        pytest.skip("the fp32-MFMA comparison build has no register-resident kernels")
    rng = np.random.default_rng(C + rows)
    store = tf_util.VariableStore(DEV, seed=rows)
    tdt = torch.float16 if dt == "f16" else torch.float32
    r = lambda *s: t(rng.normal(0, 1, s).astype(np.float32)).to(tdt)

    def job(tag):
        with tf_util.default_store(store), torch.no_grad(), tf_util.variable_scope("rr_two_stage_%s" % tag):
            P = fused.packed_layer
            layers = [P("a0", 64 + C, 128), P("a1", 128, 64)]
            layers2 = [P("b0", C + 64 + 64, 128, row_order=fused.stage2_row_order(C, 64, 64)), P("b1", 128, 64)]
            for p_ in layers + layers2:
                p_.b.copy_(torch.from_numpy(rng.normal(0, 0.1, p_.b.shape).astype(np.float32)))
        return dict(sources=[r(rows, 64), r(rows, C)], layers=layers, before=r(rows, C), after=r(rows, 64), layers2=layers2)
    ja, jb = job("a"), job("b")

    def run():
        if pair:
            (o1a, o2a), (o1b, o2b) = fused.mlp2_pair(ja, jb)
            return [o1a, o2a, o1b, o2b]
        a_, out, out2, _keep = fused._mlp2_args(**ja)
        lib.call("elo_mlp_fused", a_, out)
        return [out, out2]
    try:
        lib.lib().elo_debug_rr_rows(-1, 1 << 40)
        tile = run()
        lib.lib().elo_debug_rr_rows(-1, 0)
        lib.lib().elo_debug_rr_launches(None, 1)
        rr = run()
        counts = (ctypes.c_ulonglong * 4)()
        lib.lib().elo_debug_rr_launches(counts, 1)
        assert counts[3] == 1, "the chain kernel is the one under test"
    finally:
        lib.lib().elo_debug_rr_rows(-1, -1)
    torch.cuda.synchronize()
    for a_, b_ in zip(rr, tile):
        assert a_.dtype == b_.dtype == tdt and torch.equal(a_, b_)
    assert float(rr[1].float().abs().max()) > 0


def test_register_resident_kernels_are_stable_under_repetition(kernel_path):
    """tools/rr_stress.py: cost-volume stage 1 on the chain kernel, repeated on shapes with more workgroups than CUs, in both
    products modes: every run equals the tile kernel bit for bit (a race between the waves of a workgroup shows up as a
    run that differs: DESIGN.md section 3b, finding 4)."""
    if kernel_path != "fused":
        pytest.skip("fused kernels only")
    if load_pkg("fused").fp32_mfma():
        pytest.skip("the fp32-MFMA comparison build has no register-resident kernels")
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "rr_stress.py"), "12"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if "runs differing" in l]
    assert len(lines) == 8 and all(": 0 of 12" in l for l in lines), out.stdout


def test_dense_layers_keep_fp32_class_accuracy_across_magnitudes():
    """The fused 1x1 convolutions run as hi*hi + hi*lo + lo*hi of fp16-split operands on the fp16 matrix cores: the
    result must stay at fp32-class accuracy (not fp16's 1e-3) for inputs from 1e-6 (below fp16's normal range) to 1e3,
    mixed in one row, against a float64 reference."""
    fused, tf_util = load_pkg("fused"), load_pkg("tf_util")
    rng = np.random.default_rng(5)
    rows, K, N = 333, 144, 128
    mag = 10.0 ** rng.uniform(-6, 3, (rows, K))
    x = (rng.normal(0, 1, (rows, K)) * mag).astype(np.float32)
    W = rng.normal(0, 0.2, (K, N)).astype(np.float32) * (10.0 ** rng.uniform(-3, 0, (K, 1))).astype(np.float32)
    b = rng.normal(0, 1, (N,)).astype(np.float32)
    layer = fused.PackedDense(t(W), t(b), relu=False)
    got = fused.mlp([t(x)], [layer]).cpu().numpy().astype(np.float64)
    want = x.astype(np.float64) @ W.astype(np.float64) + b
    scale = np.abs(x.astype(np.float64)) @ np.abs(W.astype(np.float64)) + np.abs(b)      # sum of |terms|: the error scale
    err = np.abs(got - want) / scale
    assert err.max() < 1e-5 and np.median(err) < 5e-7, (err.max(), np.median(err))   # ~2^-20 per product; fp16 alone: 5e-4
    # exact zeros and a zero row stay exact
    z = np.zeros((16, K), np.float32)
    assert np.array_equal(fused.mlp([t(z)], [layer]).cpu().numpy(), np.tile(b, (16, 1)))


def test_half_products_are_one_fp16_product_with_fp32_accumulation():
    """ELO_PRODUCTS_HALF (BASELINE configs[2]'s fp16 arithmetic): a layer computes fp16(x) @ fp16(W) + b with fp32
    accumulation -- checked against exactly that in float64 (tolerance: fp32 summation), and against the unrounded
    product at fp16's 1e-3.  Mixing modes inside one launch is an argument error."""
    fused = load_pkg("fused")
    rng = np.random.default_rng(6)
    rows, K, N = 333, 144, 128
    x = (rng.normal(0, 1, (rows, K)) * 10.0 ** rng.uniform(-2, 3, (rows, K))).astype(np.float32)
    W = rng.normal(0, 0.2, (K, N)).astype(np.float32)
    b = rng.normal(0, 1, (N,)).astype(np.float32)
    half = fused.PackedDense(t(W), t(b), relu=False, half=True)
    got = fused.mlp([t(x)], [half]).cpu().numpy().astype(np.float64)
    x16, W16 = x.astype(np.float16).astype(np.float64), W.astype(np.float16).astype(np.float64)
    scale = np.abs(x16) @ np.abs(W16) + np.abs(b)
    assert (np.abs(got - (x16 @ W16 + b)) / scale).max() < 2e-6              # the spec: rounded operands, fp32 sums
    exact = x.astype(np.float64) @ W.astype(np.float64) + b
    err = np.abs(got - exact) / scale
    assert 1e-5 < err.max() < 2e-3                                          # fp16-class, and really not the split path
    full = fused.PackedDense(t(rng.normal(0, 0.2, (N, 64)).astype(np.float32)), t(np.zeros(64, np.float32)))
    with pytest.raises(RuntimeError, match="products mode"):
        fused.mlp([t(x)], [half, full])


@pytest.mark.parametrize("B,N,stride,H,W", [(3, 5000, 3, 64, 900), (1, 20000, 6, 64, 1800), (2, 777, 4, 16, 225)])
def test_input_stage_is_preprocess_plus_both_projections(B, N, stride, H, W):
    """elo_input_stage: the point half of PreProcess against the oracle (crop at 35 m, T_trans on
    the frame aug_frame names, validity re-mask; tolerance = the 4-term dot product's rounding), and its range images
    against the product's own projection of those points (same kernel code: equal up to the order duplicates add)."""
    mu = load_pkg("model_util")
    rng = np.random.default_rng(B * 1000 + N)
    cloud = rng.normal(0, 15, (B, 2 * N, stride)).astype(np.float32)
    cloud[:, :N // 20, :3] = 0                                    # invalid points in frame 1 ...
    cloud[:, -N // 10:, :3] = 0                                   # ... and frame 2

    def rigid(seed):
        r = np.random.default_rng(seed)
        a = r.normal(0, 0.05, 3)
        Rz = np.array([[np.cos(a[0]), -np.sin(a[0]), 0], [np.sin(a[0]), np.cos(a[0]), 0], [0, 0, 1]])
        Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
        T = np.eye(4); T[:3, :3] = Rz @ Ry; T[:3, 3] = r.normal(0, 0.5, 3)
        return T.astype(np.float32)
    T_tr = np.stack([rigid(10 + i) for i in range(B)])
    aug = np.array([1, 2, 1][:B])
    eye = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    pts, proj = mu.input_stage(t(cloud), t(T_tr), aug, H, W)
    w1, w2, _q, _t = O.PreProcess(cloud[:, :N, :3], cloud[:, N:, :3], eye, T_tr, np.linalg.inv(T_tr).astype(np.float32), aug)
    assert pts.shape == (2 * B, N, 3) and proj.shape == (2 * B, H, W, 3)
    close(pts[:B], w1, atol=2e-4)
    close(pts[B:], w2, atol=2e-4)
    assert (pts[:B, :N // 20] == 0).all() and (pts[B:, -N // 10:] == 0).all()            # translated zeros are re-masked
    far = np.sqrt((cloud[:, :N, :2] ** 2).sum(-1)) > 35
    assert far.any() and (pts[:B].cpu().numpy()[far] == 0).all()
    with torch.no_grad():
        again = mu.ProjectPC2SphericalRing(pts, None, H, W)[0]
    assert torch.allclose(proj, again, rtol=0, atol=1e-5)
    # no augmentation: T_trans = None
    pts0, proj0 = mu.input_stage(t(cloud), None, None, H, W)
    crop = cloud[..., :3].copy()
    crop[np.sqrt((crop[..., :2] ** 2).sum(-1)) > 35] = 0
    assert np.array_equal(pts0.cpu().numpy(), np.concatenate([crop[:, :N], crop[:, N:]], 0))


def test_input_stage_edges():
    """Empty batch, malformed clouds, and a T_trans without aug_frame are argument errors, not launches."""
    mu, ops = load_pkg("model_util"), load_pkg("_ops")
    pts, proj = mu.input_stage(torch.zeros((0, 200, 3), device=DEV), None, None, 16, 64)
    assert pts.shape == (0, 100, 3) and proj.shape == (0, 16, 64, 3)
    with pytest.raises(ValueError, match="2\\*N"):
        mu.input_stage(torch.zeros((1, 201, 3), device=DEV), None, None, 16, 64)
    with pytest.raises(ValueError, match="2\\*N"):
        mu.input_stage(torch.zeros((1, 200, 2), device=DEV), None, None, 16, 64)
    with pytest.raises(RuntimeError, match="AMD GPU"):
        mu.input_stage(torch.zeros((1, 200, 3)), None, None, 16, 64)
    # all-zero clouds: every cell stays empty
    pts, proj = mu.input_stage(torch.zeros((2, 200, 3), device=DEV), None, None, 16, 64)
    assert not pts.any() and not proj.any()


@pytest.mark.parametrize("B,two", [(1, True), (2, False), (8, True)])
def test_heterogeneous_launch_equals_separate_launches(B, two):
    """elo_cv_stage1_setconv_fused: cost-volume stage 1 and one / two set-conv jobs in ONE grid give the bits of the
    separate launches (every tile-height combination the launcher picks by size)."""
    fused, tf_util, synth = load_pkg("fused"), load_pkg("tf_util"), load_pkg("synth")
    rng = np.random.default_rng(40 + B)
    H, W, C = 8, 113, 32
    f1, f2 = synth.frame_pair(B, H, W, seed=61)
    fa, fb = (t(rng.normal(0, 1, (B, H, W, C)).astype(np.float32)) for _ in range(2))
    sparse = np.ascontiguousarray(f1[:, ::2, ::2])
    sf = [t(rng.normal(0, 1, sparse.shape[:3] + (64,)).astype(np.float32)) for _ in range(2)]
    store = tf_util.VariableStore(DEV, seed=B)
    with tf_util.default_store(store), torch.no_grad():
        P = fused.packed_layer
        layers = (P("c0", 10 + 2 * C, 128, row_order=fused.cv0_row_order(C)), P("c1", 128, 64), P("c2", 64, 64), P("cx", 10, 64),
                  P("s0", 128, 128), P("s1", 128, 64))
        perm_q, perm_u = t(rng.permutation(175).astype(np.int32)), t(rng.permutation(105).astype(np.int32))
        cv = lambda side=None: fused.cv_stage1(t(f1).reshape(B, H * W, 3), fa.reshape(B, H * W, C), t(f2), fb, None, None, *layers,
                                               group=fused.Grouping(perm_q, [7, 25], 1000), K=6, side=side)
        jobs = [dict(src_xyz=t(sparse), src_feat=sf[i], idx=None, mask=None,
                     layers=[P("u%d0" % i, 67, 128, row_order=fused.setconv_row_order(64)), P("u%d1" % i, 128, 64)],
                     xyz1_grid=t(f1), K=8, group=fused.Grouping(perm_u, [7, 15], 6.0, 2, 2)) for i in range(2 if two else 1)]
        want_cv = cv()
        want_side = [fused.setconv(**j) for j in jobs]
        got_cv, got_side = cv(side=jobs)
    torch.cuda.synchronize()
    assert torch.equal(got_cv, want_cv)
    for (g, gx), (w_, wx) in zip(got_side, want_side):
        assert torch.equal(g, w_) and gx is None and wx is None


@pytest.mark.parametrize("B,H,W,C,win,features", [(2, 16, 225, 16, (11, 41), "f32"), (1, 8, 113, 32, (7, 25), "f32"), (3, 4, 57, 64, (5, 15), "f32"),
                                                  (2, 16, 225, 16, (11, 41), "f16"), (1, 4, 57, 64, (5, 15), "f16"), (1, 5, 33, 16, (3, 9), "f32")])
def test_register_resident_cost_volume_equals_the_tile_kernel(kernel_path, B, H, W, C, win, features, chain_products):
    """cv1_rr_kernel (a wave owns 32 rows and all columns; the chain stays in registers) against the tile kernel
    (cv1_tile: four waves share the rows, activations in LDS) from the same idx / mask: the same products in the same
    order, so the outputs are equal BIT FOR BIT (fp32 and fp16 feature storage, C = 16 / 32 / 64, ragged last tiles, a
    cloud with empty pixels and masked slots); the in-kernel-grouping launch gives the same bits again."""
    if kernel_path != "fused":
        pytest.skip("fused kernels only")
    fused, tf_util, synth, elo, lib = load_pkg("fused"), load_pkg("tf_util"), load_pkg("synth"), load_pkg(), load_pkg("_lib")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    f1, f2 = synth.frame_pair(B, H, W, seed=H * W + C)
    rng = np.random.default_rng(C)
    dt = np.float16 if features == "f16" else np.float32
    fa, fb = (rng.normal(0, 1, (B, H, W, C)).astype(dt) for _ in range(2))
    N, K = H * W, 6
    perm = rng.permutation(win[0] * win[1]).astype(np.int32)
    store = tf_util.VariableStore(DEV, seed=3)
    with tf_util.default_store(store), torch.no_grad():
        P = fused.packed_layer
        layers = (P("c0", 10 + 2 * C, 128, row_order=fused.cv0_row_order(C)), P("c1", 128, 64), P("c2", 64, 64),
                  P("cx", 10, 64), P("s0", 128, 128), P("s1", 128, 64))
        for p_ in layers:                                          # non-trivial biases
            p_.b.copy_(torch.from_numpy(rng.normal(0, 0.1, p_.b.shape).astype(np.float32)))
        hw = t(synth.hw_index(B, H, W))
        idx, _, _, m = elo.fused_conv_select_k(t(f1), t(f2), hw, t(perm), H, W, N, win[0], win[1], K, 0, 1000.0, 1, 1, want_valid=False)
        m = m.reshape(B, N, K)
        run = lambda: fused.cv_stage1(t(f1).reshape(B, N, 3), t(fa).reshape(B, N, C), t(f2), t(fb), idx, m, *layers)
        try:
            lib.lib().elo_debug_cv1_rr(0)
            tile = run()
            lib.lib().elo_debug_cv1_rr(1)
            rr = run()                                               # the whole of CV_0 per row: the tile kernel's sums in its order
        finally:
            lib.lib().elo_debug_cv1_rr(-1)
        own = fused.cv_stage1(t(f1).reshape(B, N, 3), t(fa).reshape(B, N, C), t(f2), t(fb), None, None, *layers,
                              group=fused.Grouping(t(perm), list(win), 1000), K=K)
    torch.cuda.synchronize()
    assert rr.dtype == tile.dtype and torch.isfinite(rr.float()).all()
    assert torch.equal(rr, tile) and torch.equal(rr, own)
    assert m.mean() > 0.5 and m.min() == 0                           # masked slots are in the mix


@pytest.mark.parametrize("B,H,W,C,features", [(2, 16, 225, 16, "f32"), (1, 8, 113, 32, "f32"), (3, 4, 57, 64, "f32"), (2, 16, 225, 16, "f16"),
                                              (1, 4, 57, 64, "f16"), (1, 5, 33, 32, "f16")])
def test_register_resident_cost_volume_stage2_equals_the_tile_kernel(kernel_path, B, H, W, C, features, chain_products):
    """cv2_rr_kernel against cv2_kernel from the same idx / mask (random-k 3x5, K = 4 of the warped cloud on itself), and
    against the launch that groups in-kernel: bit for bit, fp32 and fp16 storage, C = 16 (feat1 is a 16-k tail block) /
    32 / 64."""
    if kernel_path != "fused":
        pytest.skip("fused kernels only")
    fused, tf_util, synth, elo, lib = load_pkg("fused"), load_pkg("tf_util"), load_pkg("synth"), load_pkg(), load_pkg("_lib")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    f1, _ = synth.frame_pair(B, H, W, seed=H * W + C + 1)
    rng = np.random.default_rng(C + 1)
    dt = np.float16 if features == "f16" else np.float32
    fa = rng.normal(0, 1, (B, H, W, C)).astype(dt)
    cost = rng.normal(0, 1, (B, H, W, 64)).astype(dt)
    N, K = H * W, 4
    perm = rng.permutation(15).astype(np.int32)
    store = tf_util.VariableStore(DEV, seed=4)
    with tf_util.default_store(store), torch.no_grad():
        P = fused.packed_layer
        order = list(range(64 + C, 128 + C)) + list(range(64)) + list(range(64, 64 + C))
        layers = (P("xe", 10, 64), P("s0", 128 + C, 128, row_order=order), P("s1", 128, 64))
        for p_ in layers:
            p_.b.copy_(torch.from_numpy(rng.normal(0, 0.1, p_.b.shape).astype(np.float32)))
        hw = t(synth.hw_index(B, H, W))
        idx, _, _, m = elo.fused_conv_random_k(t(f1), t(f1), hw, t(perm), H, W, N, 3, 5, K, 0, 2.0, 1, 1, want_valid=False)
        m = m.reshape(B, N, K)
        run = lambda: fused.cv_stage2(t(f1), t(fa), t(cost), idx, m, *layers)
        try:
            lib.lib().elo_debug_cv1_rr(0)
            tile = run()
            lib.lib().elo_debug_cv1_rr(1)
            rr = run()
        finally:
            lib.lib().elo_debug_cv1_rr(-1)
        own = fused.cv_stage2(t(f1), t(fa), t(cost), None, None, *layers, group=fused.Grouping(t(perm), [3, 5], 2.0), K=K)
    torch.cuda.synchronize()
    assert rr.dtype == tile.dtype and torch.isfinite(rr.float()).all()
    assert torch.equal(rr, tile) and torch.equal(rr, own)
    assert 0.3 < m.mean() and m.min() == 0


@pytest.mark.parametrize("case", [dict(B=2, H=16, W=225, sh=2, sw=2, K=8, mlp=[128, 64], win=(7, 15), d=3.0, feat="f32", pair=True),    # set-upconv l0 <- l1
                                  dict(B=3, H=8, W=113, sh=2, sw=2, K=8, mlp=[128, 64], win=(7, 15), d=6.0, feat="f16", pair=True),
                                  dict(B=2, H=4, W=57, sh=1, sw=2, K=8, mlp=[128, 64], win=(7, 15), d=9.0, feat="f32", pair=False),
                                  dict(B=4, H=4, W=57, sh=1, sw=2, K=16, mlp=[64, 64, 128], win=(5, 9), d=12.0, feat="f32", pair=False, down=True),   # sa1/layer3
                                  dict(B=2, H=4, W=57, sh=1, sw=2, K=16, mlp=[128, 64, 64], win=(5, 9), d=12.0, feat="f16", pair=False, down=True),   # new_layer3
                                  dict(B=2, H=8, W=113, sh=2, sw=2, K=32, mlp=[128, 64], win=(7, 15), d=20.0, feat="f32", pair=True)])   # K = 32: a point spans two waves, the pooling goes through LDS
def test_register_resident_setconv_equals_the_tile_kernel(kernel_path, case, monkeypatch, chain_products):
    """setconv_rr_kernel (in-kernel random-k by the wave that owns the rows, chain in registers, masked max through LDS)
    against setconv_kernel on the same call: bit for bit -- set-upconv shapes (every dense pixel a centre, strided sparse
    grid, K = 8, single and paired launches) and the two wide down_conv shapes (strided centre list, K = 16), fp32 and
    fp16 feature storage."""
    if kernel_path != "fused":
        pytest.skip("fused kernels only")
    fused, tf_util, synth, lib, mu = load_pkg("fused"), load_pkg("tf_util"), load_pkg("synth"), load_pkg("_lib"), load_pkg("model_util")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    B, H, W, sh, sw, K = case["B"], case["H"], case["W"], case["sh"], case["sw"], case["K"]
    f1, _ = synth.frame_pair(B, H, W, seed=H + W + K)
    rng = np.random.default_rng(K + H)
    dt = np.float16 if case["feat"] == "f16" else np.float32
    kH, kW = case["win"]
    perm = [rng.permutation(kH * kW).astype(np.int32) for _ in range(2)]
    store = tf_util.VariableStore(DEV, seed=6)
    if case.get("down"):                                        # down_conv: strided centres on the grid itself
        src_xyz = f1
        H2, W2 = H, W
        oh, ow = -(-H // sh), -(-W // sw)
        sel = mu.get_selected_idx(t(f1), sh, sw, oh, ow)
        centre_hw = sel.reshape(B, -1, 3)[:, :, 1:].contiguous()
        stride = (1, 1)
    else:                                                       # up_conv: dense centres, sparse source grid
        src_xyz = np.ascontiguousarray(f1[:, ::sh, ::sw])
        H2, W2 = src_xyz.shape[1:3]
        centre_hw = None
        stride = (sh, sw)
    feats = [rng.normal(0, 1, (B, H2, W2, 64)).astype(dt) for _ in range(2)]
    with tf_util.default_store(store), torch.no_grad():
        P = fused.packed_layer
        chains = []
        for j in range(2):
            widths = [67] + case["mlp"]
            layers = [P("j%d_l%d" % (j, i), widths[i], widths[i + 1], row_order=fused.setconv_row_order(64) if i == 0 else None)
                      for i in range(len(case["mlp"]))]
            for p_ in layers:
                p_.b.copy_(torch.from_numpy(rng.normal(0, 0.1, p_.b.shape).astype(np.float32)))
            chains.append(layers)
        job = lambda j: dict(src_xyz=t(src_xyz), src_feat=t(feats[j]), idx=None, mask=None, layers=chains[j], xyz1_grid=t(f1),
                             centre_hw=centre_hw, group=fused.Grouping(t(perm[j]), [kH, kW], case["d"], *stride), K=K)
        def run():
            if case["pair"]:
                (oa, _), (ob, _) = fused.setconv_pair(job(0), job(1))
                return [oa, ob]
            o, nx = fused.setconv(**job(0))
            return [o] + ([nx] if nx is not None else [])
        try:
            lib.lib().elo_debug_cv1_rr(0)
            tile = run()
            lib.lib().elo_debug_cv1_rr(1)
            lib.lib().elo_debug_rr_rows(0, -1)
            lib.lib().elo_debug_rr_launches(None, 1)
            rr = run()
            counts = (ctypes.c_ulonglong * 4)()
            lib.lib().elo_debug_rr_launches(counts, 1)
            assert counts[2] == 1, "the chain kernel is the one under test"
        finally:
            lib.lib().elo_debug_cv1_rr(-1)
            lib.lib().elo_debug_rr_rows(-1, -1)
    torch.cuda.synchronize()
    for a_, b_ in zip(rr, tile):
        assert a_.dtype == b_.dtype and torch.equal(a_, b_)
    assert float(rr[0].float().abs().max()) > 0


@pytest.mark.parametrize("B,H,W,C,ks,dist,mlp,feat", [(3, 16, 225, 16, [7, 11], 3.0, [16, 16, 32], "f32"), (2, 16, 113, 16, [7, 11], 3.0, [16, 16, 32], "f16"),
                                                      (5, 7, 21, 16, [7, 11], 2.0, [16, 16, 32], "f16")])
def test_narrow_setconv_on_the_matrix_cores_equals_the_valu_kernel(kernel_path, B, H, W, C, ks, dist, mlp, feat):
    """setconv_narrow_kernel (round 4: the MLP of the 19 -> 16 -> 16 -> 32 set-conv layer as 16-row MFMA blocks, weights built into
    fragments in-kernel) against setconv_small_kernel (fp32 FMAs, a lane per row): the same grouping (new_xyz equal bit for
    bit), the pooled features equal to fp32-class rounding (2e-6 of the value scale; in fp16 storage the same stored half
    but for rounding boundaries), ragged centre counts (not a multiple of the 8 centres of a workgroup), fp32 / fp16 storage."""
    if kernel_path != "fused":
        pytest.skip("fused kernels only")
    pu, mu, fused, lib = load_pkg("pointnet_util"), load_pkg("model_util"), load_pkg("fused"), load_pkg("_lib")
    if fused.fp32_mfma():
        pytest.skip("the fp32-MFMA comparison build keeps the VALU kernel")
    _, _, store, perms = _ctx()
    f1, _, feats = _scene(B, H, W, 13, C)
    oh, ow = (H + 1) // 2, (W + 1) // 2
    xyz = t(f1)
    pts = t(feats[0]).half() if feat == "f16" else t(feats[0])
    sel = mu.get_selected_idx(xyz, 2, 2, oh, ow)
    run = lambda: pu.down_conv(xyz, pts, sel, K_sample=32, kernel_size=ks, distance=dist, mlp=mlp, mlp2=None, flag_add=False,
                               is_training=False, bn_decay=None, scope='layerN')
    try:
        lib.lib().elo_debug_narrow_mfma(0)
        valu = _run(run, store, perms)
        lib.lib().elo_debug_narrow_mfma(1)
        with load_pkg("tf_util").default_store(store), load_pkg("perm").default_perm_source(perms), torch.no_grad():
            mfma = run()
    finally:
        lib.lib().elo_debug_narrow_mfma(-1)
    torch.cuda.synchronize()
    assert torch.equal(mfma[1], valu[1]) and mfma[0].dtype == valu[0].dtype == pts.dtype
    a_, b_ = mfma[0].float(), valu[0].float()
    scale = float(b_.abs().max())
    assert scale > 0 and torch.isfinite(a_).all()
    if feat == "f32":
        assert float((a_ - b_).abs().max()) <= 2e-6 * scale
    else:
        assert float((a_ - b_).abs().max()) <= 2.0 ** -10 * scale and float((a_ != b_).float().mean()) < 0.01
