"""GPU: the hand-written backward kernels (csrc/elo_backward.hip, reached through the autograd Functions of _ops.py)
against ANALYTIC gradients in double precision: torch.autograd over the float64 restatements of tests/twins_torch.py
(gather_nd -> scatter-add, reduce_max, softmax, scatter_nd -> gather: TensorFlow's gradients of the reference's ops,
utils/pointnet_util.py:54-55,110-111,203-204,277-278, model_util.py:264-273).  Tolerance 1e-4 of the gradient scale.
Also: the product holds no second implementation -- the same kernels run forward whether or not autograd records."""
import numpy as np
import pytest
import torch

import twins_torch as twin
from conftest import load_pkg
from oracle import ops_np as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _setup(seed=0, B=2, H=8, W=113, C=32, K=6):
    elo, synth = load_pkg(), load_pkg("synth")
    rng = np.random.default_rng(seed)
    f1, f2 = synth.frame_pair(B, H, W, seed=9 + seed)
    x1, x2 = t(f1), t(f2)
    hw = t(synth.hw_index(B, H, W))
    perm = t(rng.permutation(35).astype(np.int32))
    idx, _, _, m = elo.fused_conv_select_k(x1, x2, hw, perm, H, W, H * W, 5, 7, K, 0, 1000.0, 1, 1, want_valid=False)
    return rng, x1, x2, idx, m.reshape(B, H * W, K).contiguous()


def _check(fn_hip, fn_twin, inputs, wrt, seed=0, tol=1e-4):
    """inputs: list of fp32 tensors / non-tensors; wrt: indices of the tensors to differentiate.  The upstream gradient
    is random; every gradient is compared against the float64 autograd of the twin, relative to its own scale."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    a32 = [x.clone().requires_grad_(True) if i in wrt else x for i, x in enumerate(inputs)]
    a64 = [(x.double().clone().requires_grad_(True) if i in wrt else (x.double() if torch.is_tensor(x) and x.is_floating_point() else x))
           for i, x in enumerate(inputs)]
    out32, out64 = fn_hip(*a32), fn_twin(*a64)
    outs32 = [o for o in (out32 if isinstance(out32, (tuple, list)) else [out32]) if o is not None]
    outs64 = [o for o in (out64 if isinstance(out64, (tuple, list)) else [out64]) if o is not None]
    ups = [torch.randn(o.shape, generator=g).to(DEV) for o in outs32]
    for o32, o64 in zip(outs32, outs64):
        assert torch.allclose(o32.double(), o64, atol=1e-4, rtol=1e-4)
    g32 = torch.autograd.grad(outs32, [a32[i] for i in wrt], ups, allow_unused=True)
    g64 = torch.autograd.grad(outs64, [a64[i] for i in wrt], [u.double() for u in ups], allow_unused=True)
    for i, a, b in zip(wrt, g32, g64):
        assert (a is None) == (b is None), i
        if a is None:
            continue
        scale = float(b.abs().max()) + 1e-12
        err = float((a.double() - b).abs().max()) / scale
        assert err < tol, "input %d: max err %.3e of scale %.3e" % (i, err, scale)
        assert float(a.abs().max()) > 0, i                       # a real gradient, not zeros against zeros


def test_gather_kernels_backward():
    ops = load_pkg("_ops")
    rng, x1, x2, idx, m = _setup()
    B, H, W, _ = x1.shape
    C, K = 32, idx.shape[2]
    fa, fb = (t(rng.normal(0, 1, (B, H, W, C)).astype(np.float32)) for _ in range(2))
    p1 = x1.reshape(B, -1, 3).contiguous()
    _check(ops.group_concat, twin.group_concat, [p1, x2, fb, idx, m], wrt=[0, 1, 2])
    _check(ops.cv_encode1, twin.cv_encode1, [p1, fa.reshape(B, -1, C).contiguous(), x2, fb, idx, m], wrt=[0, 1, 2, 3])
    cost = t(rng.normal(0, 1, (B, H, W, 64)).astype(np.float32))
    _check(ops.cv_encode2, twin.cv_encode2, [x1, fa, cost, idx, m], wrt=[0, 1, 2])
    _check(ops.cv_encode1, twin.cv_encode1, [p1, fa.reshape(B, -1, C).contiguous(), x2, fb, idx, m], wrt=[1, 3])   # only features wanted


def test_pooling_kernels_backward():
    ops = load_pkg("_ops")
    rng, x1, x2, idx, m = _setup(1)
    B, N, K = m.shape
    big = t(rng.normal(0, 1, (B, N, K, 64)).astype(np.float32))
    val = t(rng.normal(0, 1, (B, N, K, 64)).astype(np.float32))
    _check(ops.masked_maxpool, twin.masked_maxpool, [torch.relu(big) + 0.01 * big.abs(), m], wrt=[0])
    _check(ops.masked_softmax_pool, twin.masked_softmax_pool, [big, val, m], wrt=[0, 1])
    wide = t(rng.normal(0, 1, (B, N, K, 96)).astype(np.float32))            # values as a channel slice of a wider tensor
    _check(lambda l, w_: ops.masked_softmax_pool(l, w_[..., 32:], m), lambda l, w_: twin.masked_softmax_pool(l, w_[..., 32:], m.double()),
           [big, wide], wrt=[0, 1])
    f, w = (t(rng.normal(0, 1, (B, N, 64)).astype(np.float32)) for _ in range(2))
    xyz = x1.reshape(B, N, 3).clone()
    xyz[1] = 0                                                              # a batch element without valid points: zero gradients
    a = [f.clone().requires_grad_(True), w.clone().requires_grad_(True)]
    out = ops.softmax_valid(a[0], a[1], xyz)
    gf, gw = torch.autograd.grad(out, a, torch.ones_like(out))
    assert float(gf[1].abs().max()) == 0 and float(gw[1].abs().max()) == 0 and float(gf[0].abs().max()) > 0
    _check(ops.softmax_valid, twin.softmax_valid, [f, w, x1.reshape(B, N, 3).contiguous()], wrt=[0, 1])


def _boundary_safe_points(rng, B, N, H, W):
    az_res, vres, voff = (float(x) for x in O.projection_constants(H, W))
    col = rng.integers(0, W, (B, N)) + rng.uniform(0.3, 0.7, (B, N))
    rowf = rng.integers(1, H, (B, N)) + rng.uniform(0.3, 0.7, (B, N))
    az, beta, r = np.pi - col * az_res, (rowf - voff) * vres, rng.uniform(3, 30, (B, N))
    return np.stack([r * np.cos(beta) * np.cos(az), r * np.cos(beta) * np.sin(az), r * np.sin(beta)], -1).astype(np.float32)


def test_warp_project_backward():
    """Gradients of the re-projection: to the scattered features and, through the quaternion warp, to the pose (q, t)
    and to the cloud; cell indices and the minimum-range mask carry none.  The pose is near identity so that the
    double-precision twin and the kernel put every (border-safe) point in the same cell."""
    ops = load_pkg("_ops")
    rng = np.random.default_rng(5)
    B, H, W, C = 2, 8, 113, 16
    N = H * W
    pc = _boundary_safe_points(rng, B, N, H, W)
    pc[rng.random((B, N)) < 0.1] = 0
    feat = rng.normal(0, 1, (B, N, C)).astype(np.float32)
    q = np.array([[1.0, 2e-4, -1e-4, 3e-4], [1.0, -2e-4, 1e-4, 2e-4]], np.float32)
    tt = np.array([[0.02, 0.005, -0.002], [-0.01, 0.01, 0.001]], np.float32)
    hip = lambda x, f, q_, t_: ops.warp_project(x, f, q_, t_, H, W)
    ref = lambda x, f, q_, t_: twin.warp_project(x, f, q_, t_, H, W)
    _check(hip, ref, [t(pc), t(feat), t(q), t(tt)], wrt=[0, 1, 2, 3], tol=2e-4)
    # without a warp, and without zero points: a zero point that wins its cell receives that cell's gradient from
    # tf.scatter_nd's adjoint (and from the kernel); the torch restatement routes such rows to dummy targets and loses it
    full = _boundary_safe_points(rng, B, N, H, W)
    _check(lambda x, f: ops.warp_project(x, f, None, None, H, W), lambda x, f: twin.warp_project(x, f, None, None, H, W),
           [t(full), t(feat)], wrt=[0, 1])


def test_the_same_kernels_run_with_and_without_autograd():
    """One implementation: an operator called on tensors that require grad returns bit-identical outputs to the plain
    call, and inference under torch.enable_grad() takes the fused path (no grad_fn anywhere, same bits as under no_grad)."""
    ops, model, synth = load_pkg("_ops"), load_pkg("model"), load_pkg("synth")
    rng, x1, x2, idx, m = _setup(2)
    B, H, W, _ = x1.shape
    fb = t(rng.normal(0, 1, (B, H, W, 32)).astype(np.float32))
    p1 = x1.reshape(B, -1, 3).contiguous()
    plain = ops.group_concat(p1, x2, fb, idx, m)
    rec = ops.group_concat(p1, x2, fb.clone().requires_grad_(True), idx, m)
    assert rec.grad_fn is not None and plain.grad_fn is None and torch.equal(plain, rec.detach())
    f1, f2 = synth.frame_pair(1, 64, 900, seed=4)
    a, b = t(f1), t(f2)
    net = model.PWCLONet(DEV, seed=2)
    with torch.no_grad():
        want = net.forward(a, b)
    with torch.enable_grad():
        got = pkg_forward = load_pkg("pwclo_model")
        with load_pkg("tf_util").default_store(net.store), load_pkg("perm").default_perm_source(net.perms):
            got = pkg_forward.get_model_from_projection(a, b, False)
    for g, w_ in zip(got, want):
        assert g.grad_fn is None and torch.equal(g, w_)
