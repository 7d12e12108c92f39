"""The register-resident ("chain") kernels -- cv1_rr_kernel, cv2_rr_kernel, setconv_rr_kernel, mlp2_rr_kernel: the forms
that carry every level of BASELINE configs[2] (batch 8) -- against the ORACLE, element-wise on the operator's whole
feature tensor, at north_star's 1e-4 (fp32 storage) and in the oracle's fp16 storage mode (fp16 storage).

tests/test_ops_gpu.py compares the operators with oracle/ops_np.py at batch <= 2 (below the chain kernels' row
thresholds: the tile kernels) and the chain kernels with the tile kernels bit for bit; the model tests reach the chain
kernels only through a softmax-pooled 7-float pose.  Here each operator runs at batch 8 on the pyramid's level shapes
(16x225, 8x113, 4x57; K = 4 / 6 / 8 / 16 / 32), the launch counters of the library (elo_debug_rr_launches) prove that the
chain kernel is the one that produced the tensor, and every element is compared
(reference semantics: utils/pointnet_util.py:33-149, 153-175, 179-316)."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_pkg
from oracle import ops_np as O
from util_params import export, randomise, shuffle_fn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)

CV1, CV2, SETCONV, MLP2 = 0, 1, 2, 3


@pytest.fixture(autouse=True)
def chain_regime(monkeypatch):
    """Every level takes the chain kernels, whatever its row count (the product's thresholds are throughput choices)."""
    fused, lib = load_pkg("fused"), load_pkg("_lib")
    if fused.fp32_mfma():
        pytest.skip("the fp32-MFMA comparison build has no register-resident kernels")
    load_pkg("pointnet_util").use_fused(True)
    with load_pkg("tuning").override(cv_prepass=1, chain_forms=1, setconv_chain_rows=0, mlp_chain_rows=0):
        lib.lib().elo_debug_rr_launches(None, 1)
        yield


def launches():
    counts = (ctypes.c_ulonglong * 4)()
    load_pkg("_lib").lib().elo_debug_rr_launches(counts, 1)
    return list(counts)


def run_twice(fn, store, perms):
    """Once to create the variables, randomise them, once more for the checked result (launch counters cover the 2nd run)."""
    tf_util, perm = load_pkg("tf_util"), load_pkg("perm")
    with tf_util.default_store(store), perm.default_perm_source(perms), torch.no_grad():
        fn()
        randomise(store, seed=3)
        launches()
        out = fn()
    torch.cuda.synchronize()
    return out


def ctx():
    tf_util, perm = load_pkg("tf_util"), load_pkg("perm")
    return tf_util.VariableStore(DEV, seed=1), perm.PermSource(fn=shuffle_fn)


def features(rng, shape, storage):
    """Feature tensors as the product would hold them: fp16-representable values when the storage is fp16."""
    x = rng.normal(0, 1, shape).astype(np.float32)
    return x.astype(np.float16).astype(np.float32) if storage == "f16" else x


def dev_feat(x, storage):
    return t(x).half() if storage == "f16" else t(x)


def compare(got, want, storage, what):
    """fp32 storage: north_star's 1e-4.  fp16 storage: the stored value is the fp16 rounding of a 1e-4-class value -- equal
    to the oracle's rounding of ITS value up to one fp16 ulp where the two fall on either side of a rounding boundary."""
    got = got.detach().float().cpu().numpy()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), what
    err = np.abs(got - want)
    tol = 1e-4 + 1e-4 * np.abs(want)
    if storage == "f16":
        tol = tol + 2.0 ** -10 * np.abs(want)            # one ulp of fp16 (10 mantissa bits below the leading one)
    bad = err > tol
    assert not bad.any(), "%s: %d of %d elements differ, worst %.3e at %s (want %.6f got %.6f)" % (
        what, int(bad.sum()), bad.size, float(err.max()), np.unravel_index(err.argmax(), err.shape), want.flat[err.argmax()],
        got.flat[err.argmax()])
    # and nearly all of them are far inside: a handful of wrong rows cannot hide in a mean
    assert float(np.mean(err <= 2e-5 + (2.0 ** -11 * np.abs(want) if storage == "f16" else 0))) > 0.99, what


LEVELS = [(8, 16, 225, 16, 6, [11, 41], 1.0), (8, 8, 113, 32, 6, [7, 25], 2.0), (8, 4, 57, 64, 6, [5, 15], 4.0),
          (8, 4, 57, 64, 32, [5, 35], 4.0), (4, 32, 256, 16, 6, [11, 41], 1.0)]       # the last: l0 of BASELINE configs[4]'s 128x2048 scans


@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("B,H,W,C,Kq,ks2,dist", LEVELS, ids=["l0", "l1", "l2", "l2_origin", "hires_l0"])
def test_cost_volume_on_the_chain_kernels_matches_the_oracle(B, H, W, C, Kq, ks2, dist, storage):
    """cost_volume (utils/pointnet_util.py:33-149): select-k pre-pass + cv1_rr_kernel, random-k pre-pass + cv2_rr_kernel."""
    pu, synth = load_pkg("pointnet_util"), load_pkg("synth")
    store, perms = ctx()
    f1, f2 = synth.frame_pair(B, H, W, seed=41)
    rng = np.random.default_rng(H * W + Kq)
    fa, fb = features(rng, (B, H, W, C), storage), features(rng, (B, H, W, C), storage)
    a = [t(f1), t(f2), dev_feat(fa, storage), dev_feat(fb, storage)]
    fused = load_pkg("fused")

    def forward():
        with fused.recording() as calls:                      # (keeps the stage-1 tensor the product handed to stage 2)
            out = pu.cost_volume(a[0], a[1], a[2], a[3], kernel_size1=[3, 5], kernel_size2=ks2, nsample=4, nsample_q=Kq,
                                 distance=dist, mlp1=[128, 64, 64], mlp2=[128, 64], is_training=False, bn_decay=None,
                                 scope='flow_embedding_c', bn=True, pooling='max', knn=True, corr_func='concat')
        return out, [c for c in calls if c[0] == 2][0][1][2]
    got, got_stage1 = run_twice(forward, store, perms)
    n = launches()
    assert n[CV1] == 1 and n[CV2] == 1, n
    assert got.dtype == got_stage1.dtype == (torch.float16 if storage == "f16" else torch.float32)
    params, taps = export(store), {}
    args = (params, shuffle_fn, f1, f2, fa, fb, [3, 5], ks2, 4, Kq, dist, [128, 64, 64], [128, 64], "flow_embedding_c")
    with O.feature_storage(np.float16 if storage == "f16" else None):
        O.cost_volume(*args, taps=taps)
        compare(got_stage1, taps["stage1"], storage, "cost volume stage 1 (cv1_rr_kernel)")
        want = O.cost_volume(*args, stage1=got_stage1.float().cpu().numpy())
    compare(got, want, storage, "cost volume (cv2_rr_kernel on the product's stage-1 tensor)")


@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("B,H,W,sh,sw,C1,dist", [(8, 16, 225, 2, 2, 16, 3.0), (8, 8, 113, 2, 2, 32, 6.0), (8, 4, 57, 1, 2, 64, 9.0)],
                         ids=["l0", "l1", "l2"])
def test_up_conv_and_predictor_on_the_chain_kernels_match_the_oracle(B, H, W, sh, sw, C1, dist, storage):
    """The embedding / embedding-mask set-upconvs of a refinement level and the flow predictors they feed
    (utils/pointnet_util.py:254-316, 153-175; pwclo_model.py:247-254) as the model runs them at batch 8: ONE
    setconv_rr_kernel launch (both stage-1 jobs) and ONE mlp2_rr_kernel launch (both stage-2 MLPs + both predictors)."""
    pu, synth = load_pkg("pointnet_util"), load_pkg("synth")
    store, perms = ctx()
    f1, _ = synth.frame_pair(B, H, W, seed=43)
    H2, W2 = -(-H // sh), -(-W // sw)
    sparse_xyz = np.ascontiguousarray(f1[:, ::sh, ::sw][:, :H2, :W2])
    rng = np.random.default_rng(H * W + C1)
    feat1 = features(rng, (B, H, W, C1), storage)
    sparse = [features(rng, (B, H2, W2, 64), storage) for _ in range(2)]
    cost = features(rng, (B, H * W, 64), storage)
    d = lambda x: dev_feat(x, storage)
    common = dict(xyz1_proj=t(f1), xyz2_proj=t(sparse_xyz), feat1_proj=d(feat1), kernel_size=[7, 15], stride_h=sh, stride_w=sw,
                  nsample=8, distance=dist, mlp=[128, 64], mlp2=[128, 64])
    ups = [dict(common, feat2_proj=d(sparse[j]), scope="up_%s" % tag) for j, tag in enumerate("ab")]
    fps = [dict(points_f1=d(feat1).reshape(B, H * W, C1), cost_volume=d(cost), mlp=[128, 64], scope="fp_%s" % tag) for tag in "ab"]
    fused = load_pkg("fused")

    def forward():
        jobs = pu.up_conv_stage1_jobs(ups[0], ups[1])
        (pa, _), (pb, _) = fused.setconv_pair(jobs[0], jobs[1])
        return (pa, pb) + tuple(pu.up_conv_predict_finish(ups[0], ups[1], fps[0], fps[1], pa, pb))
    got = run_twice(forward, store, perms)
    n = launches()
    assert n[SETCONV] == 1 and n[MLP2] == 1, n
    params = export(store)
    with O.feature_storage(np.float16 if storage == "f16" else None):
        for j, tag in enumerate("ab"):
            args = (params, shuffle_fn, f1, sparse_xyz, feat1, sparse[j], [7, 15], sh, sw, 8, dist, [128, 64], [128, 64], "up_%s" % tag)
            taps = {}
            O.up_conv(*args, taps=taps)
            compare(got[j], taps["pooled"], storage, "set-upconv %s stage 1 (setconv_rr_kernel)" % tag)
            up = O.up_conv(*args, pooled=got[j].float().cpu().numpy())
            compare(got[2 + 2 * j], up, storage, "set-upconv %s (mlp2_rr_kernel, first MLP)" % tag)
            # the predictor continues with the set-upconv's output AS STORED (a fused pair does not change the numbers)
            pred = O.flow_predictor(params, feat1.reshape(B, H * W, C1), got[2 + 2 * j].float().cpu().numpy(), cost, [128, 64], "fp_%s" % tag)
            compare(got[3 + 2 * j], pred, storage, "flow predictor %s (mlp2_rr_kernel, second MLP)" % tag)


@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("B,H,W,K,ks,dist,mlp", [(8, 8, 57, 16, [5, 9], 12.0, [64, 64, 128]), (8, 8, 57, 16, [5, 9], 12.0, [128, 64, 64]),
                                                 (8, 4, 57, 32, [5, 9], 6.0, [128, 64, 64])], ids=["layer3", "layer3_cost", "K32"])
def test_down_conv_on_the_chain_kernel_matches_the_oracle(B, H, W, K, ks, dist, mlp, storage):
    """down_conv (set-conv, utils/pointnet_util.py:179-250) on setconv_rr_kernel: the pyramid's wide layers (64 input
    channels; mlp [64,64,128] = layer 3, [128,64,64] = the set-conv on the initial cost volume, pwclo_model.py:138,177)."""
    pu, mu, synth = load_pkg("pointnet_util"), load_pkg("model_util"), load_pkg("synth")
    store, perms = ctx()
    f1, _ = synth.frame_pair(B, H, W, seed=47)
    feat = features(np.random.default_rng(K + mlp[0]), (B, H, W, 64), storage)
    oh, ow = (H + 1) // 2, (W + 1) // 2
    sel_np = O.get_selected_idx(B, 2, 2, oh, ow)
    xyz = t(f1)
    sel = mu.get_selected_idx(xyz, 2, 2, oh, ow)
    got = run_twice(lambda: pu.down_conv(xyz, dev_feat(feat, storage), sel, K_sample=K, kernel_size=ks, distance=dist, mlp=mlp,
                                         mlp2=None, flag_add=False, is_training=False, bn_decay=None, scope='layerC'), store, perms)
    n = launches()
    assert n[SETCONV] == 1, n
    with O.feature_storage(np.float16 if storage == "f16" else None):
        want = O.down_conv(export(store), shuffle_fn, f1, feat, sel_np, K, ks, dist, mlp, "layerC")
    compare(got[0], want[0], storage, "down_conv")
    assert np.array_equal(got[1].cpu().numpy(), want[1])


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_narrow_down_conv_on_the_matrix_cores_matches_the_oracle(storage):
    """down_conv of the pyramid's second layer (19 -> 16 -> 16 -> 32 on the 16x225 level, K = 32; utils/pointnet_util.py:179-250)
    at batch 8 on setconv_narrow_kernel (round 4: the MLP as 16-row MFMA blocks): EVERY element of the pooled feature tensor
    against the oracle -- until round 5 this kernel met the oracle at batch 1 only and the VALU kernel otherwise -- with the
    library's launch counter asserting that the matrix-core form is the one that ran."""
    pu, mu, synth, lib = load_pkg("pointnet_util"), load_pkg("model_util"), load_pkg("synth"), load_pkg("_lib")
    B, H, W, K, ks, dist, mlp = 8, 16, 225, 32, [7, 11], 3.0, [16, 16, 32]
    store, perms = ctx()
    f1, _ = synth.frame_pair(B, H, W, seed=49)
    feat = features(np.random.default_rng(19), (B, H, W, 16), storage)
    oh, ow = (H + 1) // 2, (W + 1) // 2
    sel_np = O.get_selected_idx(B, 2, 2, oh, ow)
    xyz = t(f1)
    sel = mu.get_selected_idx(xyz, 2, 2, oh, ow)
    counts = (ctypes.c_ulonglong * 2)()
    try:
        lib.lib().elo_debug_narrow_mfma(1)
        lib.lib().elo_debug_narrow_launches(None, 1)
        got = run_twice(lambda: pu.down_conv(xyz, dev_feat(feat, storage), sel, K_sample=K, kernel_size=ks, distance=dist, mlp=mlp,
                                             mlp2=None, flag_add=False, is_training=False, bn_decay=None, scope='layerN'), store, perms)
        lib.lib().elo_debug_narrow_launches(counts, 1)
    finally:
        lib.lib().elo_debug_narrow_mfma(-1)
    assert counts[0] == 2 and counts[1] == 0, list(counts)        # (run_twice: two forwards, both on setconv_narrow_kernel)
    with O.feature_storage(np.float16 if storage == "f16" else None):
        want = O.down_conv(export(store), shuffle_fn, f1, feat, sel_np, K, ks, dist, mlp, "layerN")
    compare(got[0], want[0], storage, "down_conv 19 -> 16 -> 16 -> 32 (setconv_narrow_kernel)")
    assert np.array_equal(got[1].cpu().numpy(), want[1])


def test_setconv_launcher_rejects_a_queried_grid_smaller_than_the_strided_centres():
    """the raw-pointer window walk has no final clamp, so the launcher must refuse a queried grid that does not
    cover ceil(H / stride) x ceil(W / stride) of the centres' grid instead of reading out of bounds."""
    fused, lib = load_pkg("fused"), load_pkg("_lib")
    tf_util = load_pkg("tf_util")
    store = tf_util.VariableStore(DEV, seed=0)
    B, H, W, C = 1, 16, 64, 16
    xyz1 = torch.randn((B, H, W, 3), device=DEV)
    src = torch.randn((B, 4, 16, 3), device=DEV)                   # stride 2 x 2 would need an 8 x 32 queried grid
    feat = torch.randn((B, 4, 16, C), device=DEV)
    order = torch.randperm(15, device=DEV).to(torch.int32)
    with tf_util.default_store(store), torch.no_grad():
        layers = [fused.packed_layer("rej0", 3 + C, 16, row_order=fused.setconv_row_order(C)), fused.packed_layer("rej1", 16, 16)]
    with pytest.raises(lib.EloError, match="smaller than the centres' grid"):
        fused.setconv(src, feat, None, None, layers, xyz1_grid=xyz1, group=fused.Grouping(order, [3, 5], 2.0, 2, 2), K=8)


@pytest.mark.parametrize("profile", ["dense", "kitti"])
@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("B", [1, 2])
def test_the_heterogeneous_chain_launch_matches_the_oracle_elementwise(B, storage, profile):
    """cv1_setconv_rr_kernel (elo_cv_stage1_setconv_chain): cost-volume stage 1 behind its select-k pre-pass AND the level's two
    set-upconv stage-1 jobs in ONE launch whose workgroups are partitioned between three jobs (n_cv | n_sc | n_sc) -- at l0 of a
    64 x 1800 forward, batch 1 and 2, the shape the product takes it at.  All THREE output tensors element-wise against the
    oracle (utils/pointnet_util.py:33-100, :254-298), on the dense scene and on the KITTI-density one (all-masked windows,
    masked slots gathering batch 0 pixel (0,0)); the pair-launch counter proves the kernel produced them.  (Round 5 pinned this
    launch only by 'forward with == forward without'.)"""
    pu, synth, lib, fused = load_pkg("pointnet_util"), load_pkg("synth"), load_pkg("_lib"), load_pkg("fused")
    H, W, C, Kq, ks2, dist = 16, 225, 16, 6, [11, 41], 1.0
    sh = sw = 2
    store, perms = ctx()
    g1, g2 = synth.frame_pair(B, 64, 1800, seed=61, profile=profile)
    f1, f2 = np.ascontiguousarray(g1[:, ::4, ::8]), np.ascontiguousarray(g2[:, ::4, ::8])      # the l0 grid of that pair (strides 4 x 8)
    assert f1.shape == (B, H, W, 3)
    H2, W2 = -(-H // sh), -(-W // sw)
    sparse_xyz = np.ascontiguousarray(f1[:, ::sh, ::sw][:, :H2, :W2])
    rng = np.random.default_rng(7 + B)
    fa, fb = features(rng, (B, H, W, C), storage), features(rng, (B, H, W, C), storage)
    sparse = [features(rng, (B, H2, W2, 64), storage) for _ in range(2)]
    d = lambda x: dev_feat(x, storage)
    common = dict(xyz1_proj=t(f1), xyz2_proj=t(sparse_xyz), feat1_proj=d(fa), kernel_size=[7, 15], stride_h=sh, stride_w=sw,
                  nsample=8, distance=3.0, mlp=[128, 64], mlp2=[128, 64])
    ups = [dict(common, feat2_proj=d(sparse[j]), scope="up_%s" % tag) for j, tag in enumerate("ab")]
    n_pair = ctypes.c_ulonglong(0)

    def forward():
        jobs = pu.up_conv_stage1_jobs(ups[0], ups[1])
        with fused.recording() as calls:
            out, sides = pu.cost_volume(t(f1), t(f2), d(fa), d(fb), kernel_size1=[3, 5], kernel_size2=ks2, nsample=4, nsample_q=Kq,
                                        distance=dist, mlp1=[128, 64, 64], mlp2=[128, 64], is_training=False, bn_decay=None,
                                        scope='flow_embedding_c', bn=True, pooling='max', knn=True, corr_func='concat',
                                        side_jobs=jobs, side_chain=True)
        assert sides is not None, "the chain-pair form was not taken"
        return out, [c for c in calls if c[0] == 2][0][1][2], sides[0][0], sides[1][0]
    lib.check(lib.lib().elo_debug_chain_pair_launches(ctypes.byref(n_pair), 1))
    got, got_stage1, up_a, up_b = run_twice(forward, store, perms)
    lib.check(lib.lib().elo_debug_chain_pair_launches(ctypes.byref(n_pair), 1))
    assert n_pair.value == 2, n_pair.value                          # (run_twice: two forwards, one heterogeneous launch each)
    params, taps = export(store), {}
    args = (params, shuffle_fn, f1, f2, fa, fb, [3, 5], ks2, 4, Kq, dist, [128, 64, 64], [128, 64], "flow_embedding_c")
    with O.feature_storage(np.float16 if storage == "f16" else None):
        O.cost_volume(*args, taps=taps)
        compare(got_stage1, taps["stage1"], storage, "cost-volume stage 1 inside cv1_setconv_rr_kernel")
        want = O.cost_volume(*args, stage1=got_stage1.float().cpu().numpy())
        compare(got, want, storage, "cost volume (stage 2 on the product's stage-1 tensor)")
        for j, (tag, up) in enumerate(zip("ab", (up_a, up_b))):
            taps_u = {}
            O.up_conv(params, shuffle_fn, f1, sparse_xyz, fa, sparse[j], [7, 15], sh, sw, 8, 3.0, [128, 64], [], "up_%s" % tag, taps=taps_u)   # (stage 1 only: no up_2_* variables exist)
            compare(up, taps_u["pooled"], storage, "set-upconv %s stage 1 inside cv1_setconv_rr_kernel" % tag)
