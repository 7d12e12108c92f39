"""GPU: the training layer conv2d -> batch norm (batch statistics) -> ReLU on the row-reduction kernels of
csrc/elo_train.hip (elo_bn_stats / elo_bn_apply / elo_bn_backward / elo_dense_weight_grad, through tf_util.conv2d with
is_training=True) against the same layer written with torch in float64: outputs, the moving averages, and the gradients
of the input, the weights, the bias, gamma and beta.  Reference: utils/tf_util.py:120-185 (conv2d), :512-563
(batch_norm_template)."""
import numpy as np
import pytest
import torch

from conftest import load_pkg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference(x, W, b, gamma, beta, rm, rv, momentum, eps, relu, gy, mask):
    """`mask`: the ReLU decisions of the fp32 forward (y > 0).  The float64 reference uses THE SAME decisions: among
    millions of pre-activations one lies within fp32 rounding of zero now and then, and a reference that decides it the
    other way differs by that element's whole gradient (seen: 0.08 on a scale of 4.9), which says nothing about the kernels."""
    x, W, b, gamma, beta = (t.detach().double().requires_grad_(True) for t in (x, W, b, gamma, beta))
    z = x @ W + b
    mean, var = z.mean(0), z.var(0, unbiased=False)
    y = (z - mean) / torch.sqrt(var + eps) * gamma + beta
    if relu:
        y = y * mask.double()
    (y * gy.double()).sum().backward()
    M = z.shape[0]
    new_rm = (1 - momentum) * rm.double() + momentum * mean.detach()
    new_rv = (1 - momentum) * rv.double() + momentum * var.detach() * M / (M - 1)
    return y.detach(), new_rm, new_rv, [t.grad for t in (x, W, b, gamma, beta)]


@pytest.mark.parametrize("rows,cin,cout,relu", [(4096, 6, 8, True), (921, 3, 4, True), (30000, 42, 128, True), (5000, 138, 64, False),
                                                (777, 272, 256, True), (65536, 16, 16, True), (13, 19, 32, True), (1, 8, 8, True),
                                                (120001, 67, 128, True), (150000, 128, 64, True), (100003, 8, 16, False)])
def test_training_layer_matches_float64(rows, cin, cout, relu):
    tf_util = load_pkg("tf_util")
    rng = np.random.default_rng(rows + cin)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    store = tf_util.VariableStore(DEV, seed=3)
    x = t(rng.normal(0.3, 1.5, (1, rows, 1, cin))).requires_grad_(True)
    gy = t(rng.normal(0, 1, (rows, cout)))
    with tf_util.default_store(store):
        y = tf_util.conv2d(x, cout, [1, 1], scope="layer", bn=True, is_training=True, bn_decay=0.7,
                           activation_fn=tf_util.relu if relu else None)                       # creates the variables
        P = store.params
        with torch.no_grad():                                                                 # non-trivial parameters
            P["layer/biases"].copy_(t(rng.normal(0, 0.5, (cout,))))
            P["layer/bn/gamma"].copy_(t(rng.normal(1, 0.3, (cout,))))
            P["layer/bn/beta"].copy_(t(rng.normal(0, 0.3, (cout,))))
        rm0, rv0 = store.buffers["layer/bn/moving_mean"].clone(), store.buffers["layer/bn/moving_variance"].clone()
        y = tf_util.conv2d(x, cout, [1, 1], scope="layer", bn=True, is_training=True, bn_decay=0.7,
                           activation_fn=tf_util.relu if relu else None)
    leaves = [x, P["layer/weights"], P["layer/biases"], P["layer/bn/gamma"], P["layer/bn/beta"]]
    for l in leaves:
        l.grad = None
    (y.reshape(rows, cout) * gy).sum().backward()
    if rows > 1:
        want_y, want_rm, want_rv, want_g = _reference(x.reshape(rows, cin), leaves[1], leaves[2], leaves[3], leaves[4], rm0, rv0, 0.3,
                                                      tf_util.BN_EPS, relu, gy, (y.reshape(rows, cout) > 0).detach())
        scale = lambda w: float(w.abs().max()) + 1e-6
        assert float((y.detach().reshape(rows, cout).double() - want_y).abs().max()) <= 1e-4 * scale(want_y)
        assert torch.allclose(store.buffers["layer/bn/moving_mean"].double(), want_rm, atol=1e-5, rtol=1e-5)
        assert torch.allclose(store.buffers["layer/bn/moving_variance"].double(), want_rv, atol=1e-5, rtol=1e-5)
        for name, leaf, w in zip(("x", "W", "b", "gamma", "beta"), leaves, want_g):
            got = leaf.grad.reshape(w.shape).double()
            tol = 1e-4 * scale(w) if name != "b" else 1e-3 * scale(want_g[4])      # db is a sum that cancels to ~0 under batch norm
            assert float((got - w).abs().max()) <= tol, (name, float((got - w).abs().max()), scale(w))
    else:
        assert torch.isfinite(y).all() and all(torch.isfinite(l.grad).all() for l in leaves)


def test_training_layer_is_what_torch_computes():
    """The kernel path and the torch path of tf_util._dense (ELO_TRAIN_KERNELS=0: addmm + F.batch_norm + relu) agree to fp32
    rounding on the same variables, moving averages included."""
    tf_util, tuning = load_pkg("tf_util"), load_pkg("tuning")
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.normal(0, 1, (2, 500, 6, 16)).astype(np.float32)).to(DEV)
    outs = []
    for kernels in (True, False):
        store = tf_util.VariableStore(DEV, seed=9)
        with tuning.override(train_kernels=kernels), tf_util.default_store(store):     # (the field is read at the point of use)
            xi = x.clone().requires_grad_(True)
            y = tf_util.conv2d(xi, 32, [1, 1], scope="l", bn=True, is_training=True, bn_decay=0.9)
            y.square().sum().backward()
        outs.append((y.detach(), xi.grad, store.params["l/weights"].grad, store.params["l/bn/gamma"].grad,
                     store.buffers["l/bn/moving_mean"].clone(), store.buffers["l/bn/moving_variance"].clone()))
    for a, b in zip(*outs):
        assert torch.allclose(a, b, atol=2e-4 * float(b.abs().max()), rtol=0)


def test_training_layer_matches_the_oracle():
    """SURVEY a-12, training mode: tf_util.conv2d(is_training=True) -- batch statistics, ReLU, moving-average update --
    against the numpy restatement oracle/ops_np.conv_train (utils/tf_util.py:120-185, :512-531) at 1e-4."""
    from oracle import ops_np as O
    tf_util = load_pkg("tf_util")
    rng = np.random.default_rng(11)
    x = rng.normal(0.2, 1.3, (2, 300, 6, 19)).astype(np.float32)
    store = tf_util.VariableStore(DEV, seed=4)
    with tf_util.default_store(store):
        tf_util.conv2d(torch.from_numpy(x).to(DEV), 32, [1, 1], scope="l", bn=True, is_training=True, bn_decay=0.8)   # variables
        with torch.no_grad():
            store.params["l/bn/gamma"].copy_(torch.from_numpy(rng.normal(1, 0.2, 32).astype(np.float32)))
            store.params["l/bn/beta"].copy_(torch.from_numpy(rng.normal(0, 0.2, 32).astype(np.float32)))
            store.buffers["l/bn/moving_mean"].copy_(torch.from_numpy(rng.normal(0, 0.5, 32).astype(np.float32)))
            store.buffers["l/bn/moving_variance"].copy_(torch.from_numpy(rng.uniform(0.5, 2, 32).astype(np.float32)))
        params = {k: v.detach().cpu().numpy() for k, v in store.state_dict().items()}
        y = tf_util.conv2d(torch.from_numpy(x).to(DEV), 32, [1, 1], scope="l", bn=True, is_training=True, bn_decay=0.8)
    want_y, want_mm, want_mv = O.conv_train(params, "l", x, bn_decay=0.8)
    assert np.abs(y.detach().cpu().numpy() - want_y).max() <= 1e-4 * max(1.0, np.abs(want_y).max())
    assert np.allclose(store.buffers["l/bn/moving_mean"].cpu().numpy(), want_mm, atol=1e-5, rtol=1e-5)
    assert np.allclose(store.buffers["l/bn/moving_variance"].cpu().numpy(), want_mv, atol=1e-5, rtol=1e-5)


def _literal_pose(q_raw, t_det, q_coarse, t_coarse):
    """The reference's operator chain (pwclo_model.py:206-208, :271-280 on model_util.py:17-69) in torch, any dtype."""
    pm, mu = load_pkg("pwclo_model"), load_pkg("model_util")
    B = q_raw.shape[0]
    q_det = pm._normalise_q(q_raw.reshape(B, 1, 4))
    if q_coarse is None:
        q = q_det.squeeze(1)
        return q, t_det, pm._normalise_q(q)
    tc = torch.cat([torch.zeros((B, 1, 1), dtype=q_raw.dtype), t_coarse.reshape(B, 1, 3)], -1)
    tc = mu.mul_q_point(q_det, tc, B)
    tc = mu.mul_point_q(tc, mu.inv_q(q_det, B), B)[:, :, 1:]
    q = mu.mul_point_q(q_det, q_coarse.reshape(B, 1, 4), B).squeeze(1)
    t = (tc + t_det.reshape(B, 1, 3)).squeeze(1)
    return q, t, pm._normalise_q(q)


@pytest.mark.parametrize("coarse", [True, False])
def test_pose_compose_matches_the_literal_chain_and_its_gradients(coarse):
    """_ops.pose_compose (elo_pose_compose: the pose algebra of a training step in one launch each way, adjoints written by
    hand) against the reference's operator chain under torch.autograd in float64: values to 2e-6, every input gradient to 1e-5
    of its scale, for arbitrary incoming gradients on all three outputs."""
    ops = load_pkg("_ops")
    rng = np.random.default_rng(11)
    B = 9
    mk = lambda *s: rng.normal(0, 1, s)
    q_raw, t_det, q_c, t_c = mk(B, 4) * 2.0, mk(B, 3), mk(B, 4), mk(B, 3) * 3.0
    gq, gt, gqn = mk(B, 4), mk(B, 3), mk(B, 4)
    ins64 = [torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in ((q_raw, t_det) if coarse else (q_raw, t_det, q_c, t_c))]
    want = _literal_pose(ins64[0], ins64[1], None if coarse else ins64[2], None if coarse else ins64[3])
    (want[0] * torch.tensor(gq) + 0).sum().add((want[1] * torch.tensor(gt)).sum()).add((want[2] * torch.tensor(gqn)).sum()).backward()
    ins = [torch.tensor(x, dtype=torch.float32, device=DEV, requires_grad=True) for x in ((q_raw, t_det) if coarse else (q_raw, t_det, q_c, t_c))]
    got = ops.pose_compose(ins[0], ins[1], None if coarse else ins[2], None if coarse else ins[3])
    dev = lambda x: torch.tensor(x, dtype=torch.float32, device=DEV)
    ((got[0] * dev(gq)).sum() + (got[1] * dev(gt)).sum() + (got[2] * dev(gqn)).sum()).backward()
    for g, w in zip(got, want):
        assert float((g.detach().cpu().double() - w.detach()).abs().max()) <= 2e-6 * max(1.0, float(w.detach().abs().max()))
    for a, b in zip(ins, ins64):
        scale = max(1e-3, float(b.grad.abs().max()))
        assert float((a.grad.cpu().double() - b.grad).abs().max()) <= 1e-5 * scale, (coarse, a.shape)


def test_pose_loss_matches_get_loss_and_its_gradients():
    """_ops.pose_loss (elo_pose_loss) against get_loss's literal torch chain in float64: the value, and the gradients of all
    eight pose tensors and of the two learnable loss weights (pwclo_model.py:437-481)."""
    ops, pm = load_pkg("_ops"), load_pkg("pwclo_model")
    rng = np.random.default_rng(12)
    B = 8
    poses = [rng.normal(0, 1, (B, 4 if i % 2 == 0 else 3)) for i in range(8)]
    q_gt, t_gt = rng.normal(0, 1, (B, 4)), rng.normal(0, 1, (B, 3, 1))
    p64 = [torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in poses]
    w64 = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (0.3, -2.5)]
    want = pm.get_loss(*p64, torch.tensor(q_gt), torch.tensor(t_gt), *w64)       # (CPU tensors: the literal chain)
    (want * 1.7).backward()
    p32 = [torch.tensor(x, dtype=torch.float32, device=DEV, requires_grad=True) for x in poses]
    w32 = [torch.tensor(v, dtype=torch.float32, device=DEV, requires_grad=True) for v in (0.3, -2.5)]
    got = pm.get_loss(*p32, torch.tensor(q_gt, dtype=torch.float32, device=DEV), torch.tensor(t_gt, dtype=torch.float32, device=DEV), *w32)
    (got * 1.7).backward()
    assert abs(float(got) - float(want)) <= 2e-6 * abs(float(want))
    for a, b in zip(p32 + w32, p64 + w64):
        scale = max(1e-4, float(b.grad.abs().max()))
        assert float((a.grad.cpu().double() - b.grad).abs().max()) <= 1e-5 * scale


def test_flat_adam_is_torch_adam():
    """training.FlatAdam (elo_adam_flat: one launch over the flat parameter buffer, the step's scalars computed by the host)
    against torch.optim.Adam on the same tensors and gradients over several steps with a changing learning rate: the
    variables and the moments stay equal to fp32 rounding (torch forms the moments with lerp / addcmul and the bias corrections
    in fp32, the kernel with plain multiply-adds and host doubles); and the variables ARE views of the flat buffer afterwards."""
    training, dist_mod = load_pkg("training"), load_pkg("distributed")
    rng = np.random.default_rng(5)
    shapes = [(1, 1, 6, 8), (8,), (1, 1, 138, 128), (128,), (), (3, 5)]
    mine = [torch.nn.Parameter(torch.tensor(rng.normal(0, 1, s_), dtype=torch.float32, device=DEV)) for s_ in shapes]
    ref = [torch.nn.Parameter(p_.detach().clone()) for p_ in mine]
    bucket = dist_mod.FlatGradBucket(mine)
    opt = training.FlatAdam(mine, bucket, lr=1e-3, epsilon="torch")
    topt = torch.optim.Adam(ref, lr=1e-3)
    for step in range(6):
        lr = 1e-3 * 0.7 ** step
        opt.lr = lr
        for g_ in topt.param_groups:
            g_["lr"] = lr
        for p_, r_ in zip(mine, ref):
            g = torch.tensor(rng.normal(0, 1, p_.shape), dtype=torch.float32, device=DEV)
            p_.grad.copy_(g)                              # (.grad is the bucket's view)
            r_.grad = g.clone()
        opt.step()
        topt.step()
    torch.cuda.synchronize()
    state = opt.state_dict()["state"]
    for i, (p_, r_) in enumerate(zip(mine, ref)):
        assert float((p_.detach() - r_.detach()).abs().max()) <= 2e-6 * max(1.0, float(r_.detach().abs().max())), i
        for mine_m, ref_m in ((state[i]["exp_avg"], topt.state[r_]["exp_avg"]), (state[i]["exp_avg_sq"], topt.state[r_]["exp_avg_sq"])):
            assert float((mine_m - ref_m).abs().max()) <= 1e-6 * max(1e-3, float(ref_m.abs().max())), i   # (torch: lerp / addcmul)
        assert p_.data_ptr() >= opt.flat.data_ptr() and p_.data_ptr() < opt.flat.data_ptr() + opt.flat.numel() * 4
    assert opt.t == 6
    with pytest.raises(RuntimeError, match="orphaned"):     # a second optimiser re-seats the variables: the first must refuse to step
        training.FlatAdam(mine, bucket, lr=1e-3)
        opt.launch()


def test_flat_adam_default_is_tensorflows_adam_and_survives_a_host_that_runs_ahead():
    """The default epsilon placement is tf.train.AdamOptimizer's (the optimiser the reference trains with, main.py:174):
    p -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps), against a float64 restatement -- with gradients of the order
    of eps, where the two placements differ visibly.  150 steps are enqueued WITHOUT a synchronisation and with a learning
    rate that changes every step: every step must read ITS scalars (a single pinned staging buffer was overwritten by the
    host while earlier copies were still queued)."""
    training, dist_mod = load_pkg("training"), load_pkg("distributed")
    rng = np.random.default_rng(9)
    mine = [torch.nn.Parameter(torch.tensor(rng.normal(0, 1, (257,)), dtype=torch.float32, device=DEV))]
    bucket = dist_mod.FlatGradBucket(mine)
    opt = training.FlatAdam(mine, bucket)
    steps, b1, b2, eps = 150, 0.9, 0.999, 1e-8
    grads = rng.normal(0, 1, (steps, 257)) * np.where(rng.random(257) < 0.5, 1e-8, 1.0)
    gdev = torch.tensor(grads, dtype=torch.float32, device=DEV)
    p = mine[0].detach().cpu().double().numpy().copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    filler = torch.zeros((1 << 22,), device=DEV)
    for t in range(1, steps + 1):
        lr = 1e-3 * (1.0 + (t % 7))
        opt.lr = lr
        filler.add_(1.0)                                       # keeps the stream behind the host
        mine[0].grad.copy_(gdev[t - 1])
        opt.step()
        g = grads[t - 1].astype(np.float32).astype(np.float64)
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        p -= lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m / (np.sqrt(v) + eps)
    torch.cuda.synchronize()
    got = mine[0].detach().cpu().double().numpy()
    assert np.abs(got - p).max() <= 5e-5 * max(1.0, np.abs(p).max()), float(np.abs(got - p).max())
    assert opt.t == steps


# ---- the layer's dense products on elo_dense_rows (csrc/elo_train_dense.hip) ---------------------------------------------------
@pytest.mark.parametrize("rows,cin,cout", [(1, 3, 4), (15, 6, 8), (33, 8, 8), (1000, 19, 16), (4097, 16, 32), (2049, 35, 32), (3001, 67, 64),
                                           (5003, 64, 128), (2000, 138, 128), (999, 192, 128), (777, 10, 64), (1234, 128, 138), (555, 64, 6),
                                           (640, 128, 192), (300000, 8, 8), (200001, 128, 128)])
@pytest.mark.parametrize("transposed", [False, True])
def test_dense_rows_matches_float64(rows, cin, cout, transposed):
    """x W + b and x W^T for every width class of the kernel (1, 2, 4, 8, 12 tiles; Cin % 4 == 0 or not; Cout % 4 == 0 or not; rows
    that end inside a 16-row block; W as a misaligned view, which a weight of the flat parameter buffer is), with the fused
    batch moments and moving averages where there is a bias (the forward product of a training layer)."""
    ops = load_pkg("_ops")
    g = torch.Generator(device="cpu").manual_seed(rows * 7 + cin)
    x = torch.randn(rows, cin, generator=g).to(DEV)
    flat = torch.randn(cin * cout + 3, generator=g).to(DEV) * 0.3
    W = flat[1:1 + cin * cout].view((cout, cin) if transposed else (cin, cout))      # 4-byte offset: the unaligned staging path
    b = None if transposed else torch.randn(cout, generator=g).to(DEV)
    want = x.double() @ (W.double().t() if transposed else W.double()) + (0 if b is None else b.double())
    got = ops.dense_rows(x, W, b, transposed=transposed)
    assert got.shape == want.shape
    tol = 2e-6 * (float(want.abs().max()) + 1.0) * max(1.0, cin ** 0.5 / 4)
    assert float((got.double() - want).abs().max()) <= tol
    W2 = W.contiguous()                                                              # 16-byte aligned: the vector staging path
    assert torch.equal(ops.dense_rows(x, W2, b, transposed=transposed), got)
    if b is not None and rows > 1:
        mean, invstd = torch.empty(cout, device=DEV), torch.empty(cout, device=DEV)
        rm, rv = torch.full((cout,), 0.25, device=DEV), torch.full((cout,), 2.0, device=DEV)
        z = ops.dense_rows(x, W, b, moments=(1e-3, 0.3, mean, invstd, rm, rv))
        assert torch.equal(z, got)
        m, v = want.mean(0), want.var(0, unbiased=False)
        assert float((mean.double() - m).abs().max()) <= 1e-5 * (float(m.abs().max()) + 1.0)
        assert float(((invstd.double() - 1 / torch.sqrt(v + 1e-3)) * torch.sqrt(v + 1e-3)).abs().max()) <= 2e-5
        assert float((rm.double() - (0.7 * 0.25 + 0.3 * m)).abs().max()) <= 1e-5 * (float(m.abs().max()) + 1.0)
        assert float(((rv.double() - (0.7 * 2.0 + 0.3 * v * rows / (rows - 1))) / (v + 1.0)).abs().max()) <= 2e-5


def test_dense_rows_rejects_what_it_cannot_run():
    ops, L = load_pkg("_ops"), load_pkg("_lib")
    x = torch.randn(64, 16, device=DEV)
    with pytest.raises(ValueError):
        ops.dense_rows(x, torch.randn(8, 32, device=DEV))                          # inner sizes differ
    with pytest.raises(RuntimeError):
        ops.dense_rows(x, torch.randn(16, 200, device=DEV))                        # > 192 output columns
    assert L.lib().elo_dense_rows_supported(1000, 512, 192) == 0                   # W beyond 160 KB of LDS
    with pytest.raises((RuntimeError, TypeError, ValueError)):
        ops.dense_rows(x.cpu(), torch.randn(16, 8))                                # no CPU path


@pytest.mark.parametrize("rows,c,cin,relu", [(1000, 8, 6, True), (5001, 16, 19, True), (3000, 32, 16, False), (2049, 64, 128, True), (4100, 128, 67, True),
                                             (777, 128, 138, True), (130001, 64, 64, True)])
def test_dense_rows_forms_dz_on_its_operand_load(rows, c, cin, relu):
    """dx = dz W^T with dz = batch norm's backward formed inside the kernel (elo_bn_backward(dz=NULL) + elo_dense_rows(bn_*)) against the three-
    launch elo_bn_backward followed by the plain product: the same dz (same arithmetic per element: a few ulp) and the same dx."""
    ops, L = load_pkg("_ops"), load_pkg("_lib")
    g = torch.Generator(device="cpu").manual_seed(rows + c)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)
    dy, z, W = r(rows, c), r(rows, c) * 1.5 + 0.3, r(cin, c) * 0.2
    gamma, beta = r(c) * 0.3 + 1.0, r(c) * 0.3
    mean, var = z.mean(0), z.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-3)
    scratch = torch.empty(L.lib().elo_bn_scratch_floats(c, 1), device=DEV)
    sums, dz_ref = torch.empty(2 * c, device=DEV), torch.empty_like(z)
    args = lambda dzp, s: L.BnBackwardArgs(rows, c, dy.data_ptr(), z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                           1 if relu else 0, scratch.data_ptr(), s.data_ptr(), dzp, 1)
    L.call("elo_bn_backward", args(dz_ref.data_ptr(), sums), z)
    dx_ref = ops.dense_rows(dz_ref, W, None, transposed=True)
    sums2, dz = torch.empty(2 * c, device=DEV), torch.full_like(z, float("nan"))
    L.call("elo_bn_backward", args(None, sums2), z)
    assert torch.equal(sums, sums2)
    dx = ops.dense_rows(dy, W, None, transposed=True, bn_backward=(z, mean, invstd, gamma, beta, sums2, relu, dz))
    scale = float(dz_ref.abs().max())
    assert float((dz - dz_ref).abs().max()) <= 2e-6 * scale
    assert float((dx - dx_ref).abs().max()) <= 5e-6 * float(dx_ref.abs().max())


def test_the_layer_is_the_same_on_both_dense_paths():
    """tuning.train_dense off (library GEMM + elo_bn_stats) and on (elo_dense_rows, moments fused): outputs, moving averages and
    all five gradients of one layer agree to fp32 rounding."""
    tf_util, tuning = load_pkg("tf_util"), load_pkg("tuning")
    rows, cin, cout = 130000, 67, 128
    g = torch.Generator(device="cpu").manual_seed(5)
    x0 = torch.randn(1, rows, 1, cin, generator=g).to(DEV)
    gy = torch.randn(rows, cout, generator=g).to(DEV)
    res = []
    for own in (False, True):
        store = tf_util.VariableStore(DEV, seed=3)
        x = x0.clone().requires_grad_(True)
        with tuning.override(train_dense=own), tf_util.default_store(store):
            y = tf_util.conv2d(x, cout, [1, 1], scope="layer", bn=True, is_training=True, bn_decay=0.7, activation_fn=tf_util.relu)
            (y.reshape(rows, cout) * gy).sum().backward()
        P = store.params
        res.append([y.detach(), store.buffers["layer/bn/moving_mean"], store.buffers["layer/bn/moving_variance"], x.grad,
                    P["layer/weights"].grad, P["layer/biases"].grad, P["layer/bn/gamma"].grad, P["layer/bn/beta"].grad])
    for i, (a, b) in enumerate(zip(*res)):
        # (the bias gradient of a batch-normalised layer is zero in exact arithmetic: what is left is rounding on the scale of dW)
        scale = float(res[0][4].abs().max()) if i == 5 else float(a.abs().max()) + 1e-3
        assert float((a - b).abs().max()) <= 3e-5 * scale, i


@pytest.mark.parametrize("rows,cin,cout", [(2 * 600, 6, 8), (2 * 20000, 19, 16), (2 * 70000, 16, 32), (2 * 60001, 128, 64), (2 * 333, 67, 128)])
def test_a_two_group_layer_is_two_calls_with_shared_variables(rows, cin, cout):
    """tf_util.bn_groups(2) (the Siamese training batch): ONE call on the 2B batch against the layer called once per frame with shared
    variables (pwclo_model.py:117-143): outputs, moving averages (frame 1's update, then frame 2's) and all five gradients -- on the library
    GEMM + elo_bn_stats path (few rows), on elo_dense_rows with the moments from its accumulators, and with dz formed in the dx kernel."""
    tf_util = load_pkg("tf_util")
    g = torch.Generator(device="cpu").manual_seed(rows + cin)
    x0 = (torch.randn(2, rows // 2, 1, cin, generator=g) * torch.tensor([1.0, 2.5]).view(2, 1, 1, 1) + torch.tensor([0.0, 0.7]).view(2, 1, 1, 1)).to(DEV)
    gy = torch.randn(rows, cout, generator=g).to(DEV)
    res = []
    for joint in (False, True):
        store = tf_util.VariableStore(DEV, seed=3)
        x = x0.clone().requires_grad_(True)
        layer = lambda inp: tf_util.conv2d(inp, cout, [1, 1], scope="layer", bn=True, is_training=True, bn_decay=0.7, activation_fn=tf_util.relu)
        with tf_util.default_store(store):
            if joint:
                with tf_util.bn_groups(2):
                    y = layer(x.reshape(1, rows, 1, cin)).reshape(rows, cout)
            else:
                y = torch.cat([layer(x[0:1]), layer(x[1:2])], 1).reshape(rows, cout)
            (y * gy).sum().backward()
        P = store.params
        res.append([y.detach(), store.buffers["layer/bn/moving_mean"], store.buffers["layer/bn/moving_variance"], x.grad,
                    P["layer/weights"].grad, P["layer/biases"].grad, P["layer/bn/gamma"].grad, P["layer/bn/beta"].grad])
    for i, (a, b) in enumerate(zip(*res)):
        scale = float(res[0][4].abs().max()) if i == 5 else float(a.abs().max()) + 1e-3      # (the bias gradient is zero in exact arithmetic)
        assert float((a - b).abs().max()) <= 3e-5 * scale, i
    assert torch.equal(res[0][0], res[1][0]) or float((res[0][0] - res[1][0]).abs().max()) <= 1e-6 * float(res[0][0].abs().max())


def test_dense_rows_and_weight_grad_on_random_shapes():
    """A seeded sweep over shapes nobody chose: rows 1..6000 (blocks that end anywhere), Cin 1..200, Cout 1..192, either orientation,
    the fused moments with 1 or 2 groups, misaligned W -- against float64; and elo_dense_weight_grad (vector and scalar operand loads,
    ragged trips) on the same operands."""
    ops, L = load_pkg("_ops"), load_pkg("_lib")
    rng = np.random.default_rng(77)
    for case in range(40):
        groups = int(rng.integers(1, 3))
        rows = int(rng.integers(1, 3000)) * groups
        cin, cout = int(rng.integers(1, 201)), int(rng.integers(1, 193))
        if not L.lib().elo_dense_rows_supported(rows, cin, cout) or not L.lib().elo_dense_rows_supported(rows, cout, cin):
            continue
        g = torch.Generator(device="cpu").manual_seed(case)
        x = torch.randn(rows, cin, generator=g).to(DEV)
        off = int(rng.integers(0, 4))
        flat = (torch.randn(cin * cout + 4, generator=g) * 0.3).to(DEV)
        W = flat[off:off + cin * cout].view(cin, cout)
        b = torch.randn(cout, generator=g).to(DEV)
        want = x.double() @ W.double() + b.double()
        tol = 3e-6 * (float(want.abs().max()) + 1.0) * max(1.0, cin ** 0.5 / 4)
        mean, invstd = torch.empty(groups * cout, device=DEV), torch.empty(groups * cout, device=DEV)
        rm, rv = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
        z = ops.dense_rows(x, W, b, moments=(1e-3, 0.25, mean, invstd, rm, rv), groups=groups)
        assert float((z.double() - want).abs().max()) <= tol, (case, rows, cin, cout)
        per = want.view(groups, rows // groups, cout)
        m, v = per.mean(1), per.var(1, unbiased=False)
        assert float((mean.view(groups, cout).double() - m).abs().max()) <= 2e-5 * (float(m.abs().max()) + 1.0), (case, rows, cin, cout, groups)
        # (the variance is E[z^2] - mean^2 from fp32 partial sums, as in elo_bn_stats: where mean^2 >> var the subtraction cancels --
        #  the tolerance carries that conditioning per channel)
        cond = (m * m + v) / (v + 1e-3)
        assert bool((((invstd.view(groups, cout).double() * torch.sqrt(v + 1e-3)) - 1).abs() <= 5e-5 + 2e-6 * cond).all()), (case, rows, cin, cout, groups)
        dz = torch.randn(rows, cout, generator=g).to(DEV)
        dx = ops.dense_rows(dz, W, None, transposed=True)
        dx_want = dz.double() @ W.double().t()
        assert float((dx.double() - dx_want).abs().max()) <= 3e-6 * (float(dx_want.abs().max()) + 1.0) * max(1.0, cout ** 0.5 / 4), (case, rows, cin, cout)
        dW, db = torch.empty(cin, cout, device=DEV), torch.empty(cout, device=DEV)
        scratch = torch.empty(L.lib().elo_weight_grad_slices(rows, cin, cout) * (cin * cout + cout), device=DEV)
        L.call("elo_dense_weight_grad", L.WeightGradArgs(rows, cin, cout, x.data_ptr(), dz.data_ptr(), dW.data_ptr(), db.data_ptr(), scratch.data_ptr()), x)
        dW_want = x.double().t() @ dz.double()
        assert float((dW.double() - dW_want).abs().max()) <= 1e-5 * (float(dW_want.abs().max()) + 1.0), (case, rows, cin, cout)
        assert float((db.double() - dz.double().sum(0)).abs().max()) <= 1e-5 * (float(dz.abs().sum(0).max()) + 1.0), (case, rows, cin, cout)
