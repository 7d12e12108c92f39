"""GPU parity of the two grouping ops: HIP kernels (through the C ABI) against the
committed golden vectors (reference outputs), against the CPU oracle on seeded
sweeps, and against the reference digests at full 64x1800 / 128x2048 size.
Bar: bit-exact on all four outputs (integer indices and 0/1 masks)."""
import hashlib
import json
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, expand_prefix, load_pkg
from oracle import grouping as G

pytestmark = pytest.mark.gpu


def _hip(op, xyz1, xyz2, idx, perm, kH, kW, K, fc, dist, sh, sw, want_valid=True):
    elo = load_pkg()
    fn = elo.fused_conv_random_k if op == "random" else elo.fused_conv_select_k
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    B, H, W, _ = xyz1.shape
    out = fn(t(xyz1), t(xyz2), t(idx), t(perm), H, W, idx.shape[1], kH, kW, K, flag_copy=fc,
             distance=dist, stride_h=sh, stride_w=sw, want_valid=want_valid)
    torch.cuda.synchronize()
    return [o.cpu().numpy() if o is not None else None for o in out]


def test_golden_cases_bit_exact(golden_cases):
    meta, blobs = golden_cases
    for c in meta:
        k = c["name"]
        kH, kW = c["window"]
        sel, valid, indis, mask = _hip(c["op"], blobs[k + "/xyz1"], blobs[k + "/xyz2"], blobs[k + "/idx_n2"],
                                       blobs[k + "/random_hw"], kH, kW, c["K"], c["flag_copy"], c["distance"],
                                       c["stride"][0], c["stride"][1])
        assert sel.dtype == np.int32 and mask.dtype == np.float32
        assert np.array_equal(sel, blobs[k + "/sel"]), k
        assert np.array_equal(mask, blobs[k + "/mask"]), k
        assert np.array_equal(valid, expand_prefix(blobs[k + "/n_valid"], kH * kW)), k
        assert np.array_equal(indis, expand_prefix(blobs[k + "/n_indis"], kH * kW)), k


@pytest.mark.parametrize("seed", range(24))
def test_sweep_against_oracle(seed):
    rng = np.random.default_rng(100 + seed)
    B = int(rng.integers(1, 4))
    H, W = int(rng.integers(1, 20)), int(rng.integers(6, 80))
    sh, sw = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    H2, W2 = math.ceil(H / sh), math.ceil(W / sw)
    kH = int(rng.integers(1, 12))
    kW = int(rng.integers(1, min(2 * W2, 30)))
    K = int(rng.integers(1, 40))
    lattice = seed % 2 == 0

    def cloud(h, w):
        x = rng.normal(0, 2.0, (B, h, w, 3))
        if lattice:
            x = np.round(x)
        x[rng.random((B, h, w)) < 0.15] = 0
        return x.astype(np.float32)
    xyz1, xyz2 = cloud(H, W), cloud(H2, W2)
    N = int(rng.integers(1, 300))
    idx = np.stack([rng.integers(0, H, (B, N)), rng.integers(0, W, (B, N))], -1).astype(np.int32)
    perm = rng.permutation(kH * kW).astype(np.int32)
    dist = float(rng.choice([0.5, 2.0, 5.0, 1000.0]))
    fc = int(rng.integers(0, 2))
    for op, fn in (("random", G.fused_conv_random_k), ("select", G.fused_conv_select_k)):
        want = fn(xyz1, xyz2, idx, perm, H, W, N, kH, kW, K, fc, dist, sh, sw)
        got = _hip(op, xyz1, xyz2, idx, perm, kH, kW, K, fc, dist, sh, sw)
        for g, w_ in zip(got, want):
            assert np.array_equal(g, w_), (op, seed)


@pytest.mark.parametrize("kH,kW,K", [(5, 15, 6), (7, 25, 6), (11, 41, 6), (5, 15, 32), (5, 35, 32), (5, 35, 16), (3, 60, 63),
                                     (9, 15, 8), (13, 39, 7)])
@pytest.mark.parametrize("lattice", [False, True, "sparse"])
def test_select_k_rank_forms_keep_the_reference_order(kH, kW, K, lattice):
    """The register form of select-k ranks the window (three forms by window size / K: csrc/elo_group_device.h) instead of
    running the reference's K swap rounds, and falls back to the rounds on an exact tie that touches the selected set.  On
    smooth clouds (no ties), on an integer lattice (ties everywhere) and on a 90 %-empty cloud (fewer than K candidates)
    all four outputs equal the oracle's, valid_idx / valid_in_dis_idx included."""
    rng = np.random.default_rng(kH * 1000 + kW * 10 + K)
    B, H, W = 2, 12, 70
    x = rng.normal(0, 2.0, (2, B, H, W, 3))
    if lattice is True:
        x = np.round(x)
    x[rng.random((2, B, H, W)) < (0.9 if lattice == "sparse" else 0.1)] = 0
    xyz1, xyz2 = x.astype(np.float32)
    idx = np.stack([rng.integers(0, H, (B, 200)), rng.integers(0, W, (B, 200))], -1).astype(np.int32)
    perm = rng.permutation(kH * kW).astype(np.int32)
    for dist in (1.5, 1000.0):
        want = G.fused_conv_select_k(xyz1, xyz2, idx, perm, H, W, 200, kH, kW, K, 0, dist, 1, 1)
        got = _hip("select", xyz1, xyz2, idx, perm, kH, kW, K, 0, dist, 1, 1)
        for g, w_ in zip(got, want):
            assert np.array_equal(g, w_)
        got = _hip("select", xyz1, xyz2, idx, perm, kH, kW, K, 0, dist, 1, 1, want_valid=False)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[3], want[3])


def test_edge_cases():
    # all-empty clouds, one centre, K larger than the window, widest legal window
    z = np.zeros((2, 3, 9, 3), np.float32)
    idx = np.zeros((2, 1, 2), np.int32)
    for op in ("random", "select"):
        sel, valid, indis, mask = _hip(op, z, z, idx, np.arange(15, dtype=np.int32), 3, 5, 4, 1, 1.0, 1, 1)
        assert not sel.any() and not mask.any() and not valid.any() and not indis.any()
    rng = np.random.default_rng(5)
    x = rng.normal(0, 1, (1, 2, 5, 3)).astype(np.float32)
    idx = np.array([[[1, 4], [0, 0]]], np.int32)
    perm = rng.permutation(2 * 11).astype(np.int32)       # kW/2 = 5 == W2: every column wraps
    for op, fn in (("random", G.fused_conv_random_k), ("select", G.fused_conv_select_k)):
        for fc in (0, 1):
            want = fn(x, x, idx, perm, 2, 5, 2, 2, 11, 40, fc, 100.0, 1, 1)
            got = _hip(op, x, x, idx, perm, 2, 11, 40, fc, 100.0, 1, 1)
            for g, w_ in zip(got, want):
                assert np.array_equal(g, w_)
        got = _hip(op, x, x, idx, perm, 2, 11, 40, 0, 100.0, 1, 1, want_valid=False)
        assert got[1] is None and got[2] is None and np.array_equal(got[0], want[0] if fc == 0 else got[0])


def test_torch_ops_dispatch_to_the_hip_kernels():
    load_pkg("torch_ops")
    rng = np.random.default_rng(2)
    x = rng.normal(0, 2, (1, 6, 20, 3)).astype(np.float32)
    idx = np.stack([rng.integers(0, 6, (1, 9)), rng.integers(0, 20, (1, 9))], -1).astype(np.int32)
    perm = rng.permutation(15).astype(np.int32)
    t = lambda a: torch.from_numpy(a).to("cuda:0")
    for name, ofn in (("fused_conv_random_k", G.fused_conv_random_k), ("fused_conv_select_k", G.fused_conv_select_k)):
        got = getattr(torch.ops.elo, name)(t(x), t(x), t(idx), t(perm), 6, 20, 9, 3, 5, 4, 0, 2.0, 1, 1)
        want = ofn(x, x, idx, perm, 6, 20, 9, 3, 5, 4, 0, 2.0, 1, 1)
        for g, w_ in zip(got, want):
            assert np.array_equal(g.cpu().numpy(), w_)


def test_error_behaviour():
    elo = load_pkg()
    dev = "cuda:0"
    x = torch.zeros(1, 4, 8, 3, device=dev)
    idx = torch.zeros(1, 2, 2, dtype=torch.int32, device=dev)
    perm = torch.arange(15, dtype=torch.int32, device=dev)
    with pytest.raises(ValueError, match="positive K"):
        elo.fused_conv_random_k(x, x, idx, perm, 4, 8, 2, 3, 5, 0, 0, 1.0, 1, 1)
    with pytest.raises(ValueError, match="positive distance"):
        elo.fused_conv_select_k(x, x, idx, perm, 4, 8, 2, 3, 5, 4, 0, 0.0, 1, 1)
    with pytest.raises(ValueError, match="xyz2 shape"):
        elo.fused_conv_random_k(x, x, idx, perm, 4, 8, 2, 3, 5, 4, 0, 1.0, 2, 1)
    with pytest.raises(ValueError, match="random_hw shape"):
        elo.fused_conv_random_k(x, x, idx, perm[:14], 4, 8, 2, 3, 5, 4, 0, 1.0, 1, 1)
    with pytest.raises(Exception, match="no CPU fallback"):
        elo.fused_conv_random_k(x.cpu(), x.cpu(), idx.cpu(), perm.cpu(), 4, 8, 2, 3, 5, 4, 0, 1.0, 1, 1)
    big = torch.arange(71 * 71, dtype=torch.int32, device=dev)          # 5041 > 5000 slots
    xx = torch.zeros(1, 80, 80, 3, device=dev)
    with pytest.raises(Exception, match="exceeds 5000"):
        elo.fused_conv_select_k(xx, xx, idx, big, 80, 80, 2, 71, 71, 4, 0, 1.0, 1, 1)


def _digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("i", range(4))
def test_full_size_digests(i):
    """BASELINE config 1 (64x1800 all-pixel random-k, 9x15, K=16, d=0.5) and the other
    full-resolution cases: sha256 of the REFERENCE outputs."""
    with open(os.path.join(GOLDEN, "large_digests.json")) as f:
        d = json.load(f)[i]
    synth = load_pkg("synth")
    f1, f2 = synth.frame_pair(1, d["H"], d["W"], seed=d["seed"])
    idx = synth.hw_index(1, d["H"], d["W"])
    kH, kW = d["window"]
    perm = np.random.default_rng(d["seed"]).permutation(kH * kW).astype(np.int32)
    sel, valid, indis, mask = _hip(d["op"], f1, f2, idx, perm, kH, kW, d["K"], 0, d["distance"], 1, 1)
    if _digest(f1, f2, idx, perm) == d["inputs_sha256"]:
        assert _digest(sel) == d["sel_sha256"]
        assert _digest(mask) == d["mask_sha256"]
        assert _digest(valid.sum(2).astype(np.int32), indis.sum(2).astype(np.int32)) == d["counts_sha256"]
    # independent of the RNG stream: equality with the oracle on the same inputs
    fn = G.fused_conv_random_k if d["op"] == "random" else G.fused_conv_select_k
    want = fn(f1, f2, idx, perm, d["H"], d["W"], idx.shape[1], kH, kW, d["K"], 0, d["distance"], 1, 1, threads=16)
    for g, w_ in zip((sel, valid, indis, mask), want):
        assert np.array_equal(g, w_)
    # size-independent properties: every selected index is inside the window and within the radius
    b, h, w = sel[..., 0], sel[..., 1], sel[..., 2]
    m = mask[..., 0] == 1
    assert (b[m] == 0).all() and (h[m] >= 0).all() and (h[m] < d["H"]).all()
    ch = idx[0, :, 0][:, None]
    assert (np.abs(h - ch)[m] <= kH // 2).all()
    p = f2[0][h, w]
    cpt = f1[0][idx[0, :, 0], idx[0, :, 1]][:, None, :]
    dd = ((cpt - p) ** 2).sum(-1)
    assert (dd[m] <= d["distance"] ** 2 * (1 + 1e-5)).all()


def test_in_kernel_grouping_of_fused_kernels_is_bit_exact():
    """The fused kernels group in-kernel (elo_group_spec): their idx_out / mask_out must equal the oracle's
    random-k / select-k outputs bit for bit, for strided centres, all-pixel centres with a stride, and select-k."""
    fused, tf_util = load_pkg("fused"), load_pkg("tf_util")
    synth = load_pkg("synth")
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    f1, f2 = synth.frame_pair(2, 16, 225, seed=77)
    rng = np.random.default_rng(3)
    store = tf_util.VariableStore(dev, seed=0)
    with tf_util.default_store(store), torch.no_grad():
        # (a) set-conv: strided centres, 7x11 window, K=32
        C = 16
        feat = rng.normal(0, 1, (2, 16, 225, C)).astype(np.float32)
        hw = synth.strided_index(2, 8, 113, 2, 2)
        perm = rng.permutation(77).astype(np.int32)
        g = fused.Grouping(t(perm), [7, 11], 3.0, want_indices=True)
        fused.setconv(t(f1), t(feat), None, None, [fused.packed_layer("a0", 3 + C, 32)], xyz1_grid=t(f1),
                      centre_hw=t(hw), K=32, group=g)
        want = G.fused_conv_random_k(f1, f1, hw, perm, 16, 225, hw.shape[1], 7, 11, 32, 0, 3.0, 1, 1)
        assert np.array_equal(g.idx.cpu().numpy(), want[0]) and np.array_equal(g.mask.cpu().numpy()[..., None], want[3])
        # (b) set-upconv: every pixel of the dense grid, sparse grid at stride (2,2), K=8
        sparse = np.ascontiguousarray(f2[:, ::2, ::2])
        sfeat = rng.normal(0, 1, sparse.shape[:3] + (64,)).astype(np.float32)
        perm = rng.permutation(105).astype(np.int32)
        g = fused.Grouping(t(perm), [7, 15], 3.0, 2, 2, want_indices=True)
        fused.setconv(t(sparse), t(sfeat), None, None, [fused.packed_layer("b0", 67, 64)], xyz1_grid=t(f1), K=8, group=g)
        hw_all = synth.hw_index(2, 16, 225)
        want = G.fused_conv_random_k(f1, sparse, hw_all, perm, 16, 225, 3600, 7, 15, 8, 0, 3.0, 2, 2)
        assert np.array_equal(g.idx.cpu().numpy(), want[0]) and np.array_equal(g.mask.cpu().numpy()[..., None], want[3])
        # (c) cost volume stage 1 (select-k, 11x41, K=6) and stage 2 (random-k on itself, 3x5, K=4)
        C = 16
        fa, fb = (rng.normal(0, 1, (2, 16, 225, C)).astype(np.float32) for _ in range(2))
        perm = rng.permutation(451).astype(np.int32)
        P = fused.packed_layer
        g = fused.Grouping(t(perm), [11, 41], 1000, want_indices=True)
        cost = fused.cv_stage1(t(f1).reshape(2, 3600, 3), t(fa).reshape(2, 3600, C), t(f2), t(fb), None, None,
                               P("c0", 10 + 2 * C, 128), P("c1", 128, 64), P("c2", 64, 64), P("cx", 10, 64),
                               P("s0", 128, 128), P("s1", 128, 64), group=g, K=6)
        want = G.fused_conv_select_k(f1, f2, hw_all, perm, 16, 225, 3600, 11, 41, 6, 0, 1000.0, 1, 1)
        assert np.array_equal(g.idx.cpu().numpy(), want[0]) and np.array_equal(g.mask.cpu().numpy()[..., None], want[3])
        perm = rng.permutation(15).astype(np.int32)
        g = fused.Grouping(t(perm), [3, 5], 1.0, want_indices=True)
        fused.cv_stage2(t(f1), t(fa), cost.reshape(2, 16, 225, 64), None, None, P("e0", 10, 64), P("e1", 128 + C, 128),
                        P("e2", 128, 64), group=g, K=4)
        want = G.fused_conv_random_k(f1, f1, hw_all, perm, 16, 225, 3600, 3, 5, 4, 0, 1.0, 1, 1)
        assert np.array_equal(g.idx.cpu().numpy(), want[0]) and np.array_equal(g.mask.cpu().numpy()[..., None], want[3])
    torch.cuda.synchronize()


def test_select_k32_window_5x35_at_batch8_is_bit_exact():
    """BASELINE configs[2]'s select-k: the l2_origin cost volume's call (5x35 window, K = 32, distance 1000, every pixel
    of the 4x57 l2 grid of a 64x1800 scan; pwclo_model.py:170-172, utils/pointnet_util.py:49-51) at batch 8, run inside
    the fused stage-1 kernel with fp16 feature storage: indices and mask equal the oracle's bit for bit."""
    fused, tf_util = load_pkg("fused"), load_pkg("tf_util")
    synth = load_pkg("synth")
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    B = 8
    f1, f2 = synth.frame_pair(B, 64, 1800, seed=31)
    x1, x2 = np.ascontiguousarray(f1[:, ::16, ::32]), np.ascontiguousarray(f2[:, ::16, ::32])   # the l2 grids: 4 x 57
    assert x1.shape == (B, 4, 57, 3)
    rng = np.random.default_rng(8)
    C = 64
    fa, fb = (rng.normal(0, 1, (B, 4, 57, C)).astype(np.float16) for _ in range(2))
    perm = rng.permutation(175).astype(np.int32)
    store = tf_util.VariableStore(dev, seed=0)
    with tf_util.default_store(store), torch.no_grad():
        P = fused.packed_layer
        g = fused.Grouping(t(perm), [5, 35], 1000, want_indices=True)
        out = fused.cv_stage1(t(x1).reshape(B, 228, 3), t(fa).reshape(B, 228, C), t(x2), t(fb), None, None,
                              P("c0", 10 + 2 * C, 128, row_order=fused.cv0_row_order(C)), P("c1", 128, 64), P("c2", 64, 64),
                              P("cx", 10, 64), P("s0", 128, 128), P("s1", 128, 64), group=g, K=32)
    assert out.dtype == torch.float16
    want = G.fused_conv_select_k(x1, x2, synth.hw_index(B, 4, 57), perm, 4, 57, 228, 5, 35, 32, 0, 1000.0, 1, 1)
    assert np.array_equal(g.idx.cpu().numpy(), want[0]) and np.array_equal(g.mask.cpu().numpy()[..., None], want[3])
    assert want[3].sum() > 0.5 * want[3].size          # a populated case, not an all-masked one


@pytest.mark.parametrize("case", [dict(H=64, W=1800, win=(9, 15), K=16, d=0.5, stride=(1, 1), fc=0),      # BASELINE configs[0]
                                  dict(H=16, W=225, win=(3, 5), K=4, d=1.0, stride=(1, 1), fc=0),          # cost-volume stage 2
                                  dict(H=16, W=225, win=(7, 15), K=8, d=3.0, stride=(2, 2), fc=0),         # set-upconv, strided
                                  dict(H=7, W=70, win=(5, 9), K=6, d=2.0, stride=(1, 2), fc=1),            # ragged tile, flag_copy
                                  dict(H=5, W=33, win=(3, 41), K=5, d=1000.0, stride=(1, 1), fc=0),        # window wider than the image
                                  dict(H=128, W=2048, win=(11, 41), K=6, d=4.0, stride=(1, 1), fc=0)])      # configs[4]'s grid
def test_dense_lds_tiled_random_k_is_bit_exact(case):
    """elo_fused_conv_random_k_dense (every pixel a centre, window union of a 4 x 64 tile staged in LDS, one thread per
    centre walking it in the visiting order) against the oracle: all four outputs, bit for bit, incl. the cylindrical
    wrap, rows outside the grid, empty pixels, strides, flag_copy, ragged last tiles; and it equals the general kernel."""
    elo, synth = load_pkg(), load_pkg("synth")
    H, W, (kH, kW), K, (sh, sw) = case["H"], case["W"], case["win"], case["K"], case["stride"]
    B = 2 if H * W < 20000 else 1
    f1, f2 = synth.frame_pair(B, H, W, seed=H + W)
    x2 = np.ascontiguousarray(f2[:, ::sh, ::sw])
    rng = np.random.default_rng(kH * kW)
    perm = rng.permutation(kH * kW).astype(np.int32)
    hw = synth.hw_index(B, H, W)
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    args = (t(f1), t(x2), t(hw), t(perm), H, W, H * W, kH, kW, K, case["fc"], case["d"], sh, sw)
    got = elo.fused_conv_random_k(*args, dense=True)
    gen = elo.fused_conv_random_k(*args, dense=False)
    want = G.fused_conv_random_k(f1, x2, hw, perm, H, W, H * W, kH, kW, K, case["fc"], case["d"], sh, sw, threads=8)
    for g, o, w_ in zip(got, gen, want):
        assert np.array_equal(g.cpu().numpy(), w_) and torch.equal(g, o)
    # the index tensors get_hw_idx hands out take the dense kernel on their own
    pu = load_pkg("pointnet_util")
    auto = elo.fused_conv_random_k(args[0], args[1], pu.get_hw_idx(B, H, W, dev), *args[3:], want_valid=False)
    assert torch.equal(auto[0], got[0]) and torch.equal(auto[3], got[3])


@pytest.mark.parametrize("case", [dict(H=16, W=225, win=(11, 41), K=6, d=1000.0, stride=(1, 1), cloud="scan"),     # l0 cost volume
                                  dict(H=8, W=113, win=(7, 25), K=6, d=1000.0, stride=(1, 1), cloud="scan"),      # l1
                                  dict(H=4, W=57, win=(5, 15), K=6, d=1000.0, stride=(1, 1), cloud="scan"),       # l2: window union wider than the grid
                                  dict(H=12, W=70, win=(9, 15), K=7, d=1.5, stride=(1, 1), cloud="lattice"),      # exact ties on every centre: the fallback
                                  dict(H=12, W=70, win=(5, 15), K=4, d=1000.0, stride=(1, 1), cloud="sparse"),    # fewer than K (and than 8) neighbours
                                  dict(H=9, W=130, win=(13, 39), K=5, d=2.5, stride=(1, 1), cloud="normal"),      # 507 slots, ragged last tile, small radius
                                  dict(H=16, W=64, win=(5, 9), K=6, d=1000.0, stride=(2, 2), cloud="normal"),     # strided query grid
                                  dict(H=32, W=256, win=(11, 41), K=6, d=1000.0, stride=(1, 1), cloud="scan")])   # configs[4]'s l0 grid
@pytest.mark.parametrize("waves", [4, 8, 16])
def test_dense_lds_tiled_select_k_is_bit_exact(case, waves):
    """elo_fused_conv_select_k_dense (every pixel a centre, K <= 7: the window union of 64 centres staged in LDS, a lane per
    centre walking it twice -- class minima -> bound, then candidates -- and K + 1 minima pulled from the candidate list;
    exact ties and overflowing lists redone by the wave-per-centre form) against the oracle and the general kernel: all
    four outputs bit for bit, for 4, 8 and 16 waves per tile."""
    elo, synth = load_pkg(), load_pkg("synth")
    H, W, (kH, kW), K, (sh, sw) = case["H"], case["W"], case["win"], case["K"], case["stride"]
    B = 2
    rng = np.random.default_rng(kH * kW + K)
    if case["cloud"] == "scan":
        f1, f2 = synth.frame_pair(B, H, W, seed=H + W)
    else:
        x = rng.normal(0, 2.0, (2, B, H, W, 3))
        if case["cloud"] == "lattice":
            x = np.round(x)
        x[rng.random((2, B, H, W)) < (0.93 if case["cloud"] == "sparse" else 0.1)] = 0
        f1, f2 = x.astype(np.float32)
    x2 = np.ascontiguousarray(f2[:, ::sh, ::sw])
    perm = rng.permutation(kH * kW).astype(np.int32)
    hw = synth.hw_index(B, H, W)
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    args = (t(f1), t(x2), t(hw), t(perm), H, W, H * W, kH, kW, K, 0, case["d"], sh, sw)
    want = G.fused_conv_select_k(f1, x2, hw, perm, H, W, H * W, kH, kW, K, 0, case["d"], sh, sw, threads=8)
    # (the waves-per-tile choice follows the grid size; the library's debugging hook forces each of the three forms)
    load_pkg("_lib").lib().elo_debug_select_dense_waves(waves)
    try:
        got = elo.fused_conv_select_k(*args, dense=True)
        lean = elo.fused_conv_select_k(*args, dense=True, want_valid=False)
    finally:
        load_pkg("_lib").lib().elo_debug_select_dense_waves(0)
    gen = elo.fused_conv_select_k(*args, dense=False)
    for g, o, w_ in zip(got, gen, want):
        assert np.array_equal(g.cpu().numpy(), w_) and torch.equal(g, o)
    assert torch.equal(lean[0], got[0]) and torch.equal(lean[3], got[3])
    if case["cloud"] == "scan":
        assert want[3].mean() > 0.8                       # populated windows, not a trivially empty case
    pu = load_pkg("pointnet_util")
    auto = elo.fused_conv_select_k(args[0], args[1], pu.get_hw_idx(B, H, W, dev), *args[3:], want_valid=False)
    assert torch.equal(auto[0], got[0]) and torch.equal(auto[3], got[3])


def test_dense_select_k_refuses_what_it_cannot_do():
    elo, synth = load_pkg(), load_pkg("synth")
    f1, f2 = synth.frame_pair(1, 8, 64, seed=1)
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    perm = np.arange(15, dtype=np.int32)
    with pytest.raises(ValueError, match="K <= 7"):
        elo.fused_conv_select_k(t(f1), t(f2), t(synth.hw_index(1, 8, 64)), t(perm), 8, 64, 512, 3, 5, 8, 0, 1.0, 1, 1, dense=True)


@pytest.mark.parametrize("case", [
    dict(B=2, H=64, W=450, C=3, cs=(4, 8), win=(9, 15), d=0.5, mlp=[8, 8, 16], holes=0.05),       # the pyramid's first set-conv (64x1800 at full size)
    dict(B=2, H=64, W=450, C=3, cs=(4, 8), win=(9, 15), d=0.12, mlp=[8, 8, 16], holes=0.5),        # sparse cloud, small radius: several rounds, short lists
    dict(B=3, H=16, W=225, C=16, cs=(2, 2), win=(7, 11), d=3.0, mlp=[16, 16, 32], holes=0.1),      # the second one (8x113 centres: a ragged last tile)
    dict(B=1, H=6, W=21, C=3, cs=(1, 1), win=(9, 15), d=2.0, mlp=[8, 8, 16], holes=0.2),           # window taller than the grid, columns wrap both ways
    dict(B=2, H=9, W=70, C=16, cs=(2, 3), win=(5, 9), d=1000.0, mlp=[16, 16, 32], holes=0.0),      # everything in range: the walk stops after one round
    dict(B=1, H=8, W=64, C=3, cs=(2, 2), win=(9, 15), d=0.5, mlp=[8, 8, 16], holes=1.0)])          # an EMPTY cloud: every centre invalid
@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_narrow_setconv_in_kernel_grouping_is_bit_exact(case, storage):
    """setconv_small_kernel's in-kernel random-k over STRIDED centres (half a wave per centre, a lane per window slot: the two narrow
    set-conv layers of the pyramid): its neighbour indices / masks equal the ORACLE's random-k bit for bit on dense, sparse,
    wrapping and EMPTY clouds.  (Round 5's LDS-staged window form of this kernel, setconv_tiled_kernel, was measured slower and
    lives in tools/micro/patches/r06_rejected_forms.patch; these are its cases.)"""
    fused, tf_util, pu, mu, synth, tuning, lib = (load_pkg("fused"), load_pkg("tf_util"), load_pkg("pointnet_util"), load_pkg("model_util"),
                                                  load_pkg("synth"), load_pkg("tuning"), load_pkg("_lib"))
    import ctypes
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    B, H, W, C, (sh, sw), (kH, kW) = case["B"], case["H"], case["W"], case["C"], case["cs"], case["win"]
    f1, _ = synth.frame_pair(B, H, W, seed=H * W + C)
    rng = np.random.default_rng(kH * kW + C)
    f1 = f1.copy()
    f1[rng.random((B, H, W)) < case["holes"]] = 0                              # empty pixels: skipped as neighbours, invalid as centres
    oh, ow = -(-H // sh), -(-W // sw)
    feat = rng.normal(0, 1, (B, H, W, C)).astype(np.float32)
    feat_dev = t(feat).half() if storage == "f16" else t(feat)
    perm = rng.permutation(kH * kW).astype(np.int32)
    xyz = t(f1)
    sel = mu.get_selected_idx(xyz, sh, sw, oh, ow)
    centre_hw = pu._centre_hw(sel)
    store = tf_util.VariableStore(dev, seed=2)
    widths = [3 + C] + case["mlp"]
    with tf_util.default_store(store), torch.no_grad():
        layers = [fused.packed_layer("t%d" % i, widths[i], widths[i + 1], row_order=fused.setconv_row_order(C) if i == 0 else None)
                  for i in range(3)]
        for p_ in layers:
            p_.b.copy_(torch.from_numpy(rng.normal(0, 0.1, p_.b.shape).astype(np.float32)))

        def run():
            g = fused.Grouping(t(perm), [kH, kW], case["d"], want_indices=True)
            out, new_xyz = fused.setconv(xyz, feat_dev, None, None, layers, xyz1_grid=xyz, centre_hw=centre_hw, K=32, group=g)
            return out, new_xyz, g.idx, g.mask
        counts = (ctypes.c_ulonglong * 2)()
        lib.lib().elo_debug_narrow_launches(None, 1)
        with tuning.override(narrow_mfma=0):
            tiled = run()
        lib.lib().elo_debug_narrow_launches(counts, 1)
    torch.cuda.synchronize()
    assert list(counts) == [0, 1], list(counts)
    hw = synth.strided_index(B, oh, ow, sh, sw)
    want = G.fused_conv_random_k(f1, f1, hw, perm, H, W, oh * ow, kH, kW, 32, 0, case["d"], 1, 1)
    assert np.array_equal(tiled[2].cpu().numpy(), want[0]) and np.array_equal(tiled[3].cpu().numpy()[..., None], want[3])
    if case["holes"] < 1.0:
        assert 0 < float(want[3].mean()) and float(tiled[0].float().abs().max()) > 0
    else:
        assert float(want[3].max()) == 0
