"""GPU: the training path -- the torch restatements used as gradient references (tests/twins_torch.py) equal the HIP
kernels in the forward direction, gradients reach every layer, get_loss / PreProcess / the full get_model signature match the oracle, and a
few optimisation steps reduce the loss."""
import copy

import numpy as np
import pytest
import torch

from conftest import load_pkg
from oracle import ops_np as O
from util_params import close, export, randomise, shuffle_fn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_torch_twins_equal_hip_kernels():
    import twins_torch as twin
    ops, elo, synth = load_pkg("_ops"), load_pkg(), load_pkg("synth")
    rng = np.random.default_rng(0)
    B, H, W, C, K = 2, 8, 113, 32, 6
    f1, f2 = synth.frame_pair(B, H, W, seed=9)
    x1, x2 = t(f1), t(f2)
    fa, fb = t(rng.normal(0, 1, (B, H, W, C)).astype(np.float32)), t(rng.normal(0, 1, (B, H, W, C)).astype(np.float32))
    hw = t(synth.hw_index(B, H, W))
    perm = t(rng.permutation(35).astype(np.int32))
    idx, _, _, m = elo.fused_conv_select_k(x1, x2, hw, perm, H, W, H * W, 5, 7, K, 0, 1000.0, 1, 1, want_valid=False)
    m = m.reshape(B, H * W, K)
    p1, ff1 = x1.reshape(B, -1, 3), fa.reshape(B, -1, C)
    with torch.no_grad():
        for name, args in (("group_concat", (p1, x2, fb, idx, m)), ("cv_encode1", (p1, ff1, x2, fb, idx, m))):
            close(getattr(twin, name)(*args), getattr(ops, name)(*args).cpu().numpy(), atol=1e-6)
        big = t(rng.normal(0, 1, (B, H * W, K, 64)).astype(np.float32))
        val = t(rng.normal(0, 1, (B, H * W, K, 64)).astype(np.float32))
        close(twin.masked_maxpool(big, m), ops.masked_maxpool(big, m).cpu().numpy(), atol=1e-6)
        close(twin.masked_softmax_pool(big, val, m), ops.masked_softmax_pool(big, val, m).cpu().numpy(), atol=1e-5)
        cost = t(rng.normal(0, 1, (B, H, W, 64)).astype(np.float32))
        for a_, b_ in zip(twin.cv_encode2(x1, fa, cost, idx, m), ops.cv_encode2(x1, fa, cost, idx, m)):
            close(a_, b_.cpu().numpy(), atol=1e-6)
        close(twin.softmax_valid(ff1, ff1 * 2, p1), ops.softmax_valid(ff1, ff1 * 2, p1).cpu().numpy(), atol=1e-5)
        q = t(np.array([[1.0, 0.002, -0.001, 0.004], [0.999, -0.003, 0.002, 0.01]], np.float32))
        tt = t(np.array([[0.5, 0.05, -0.02], [-0.3, 0.1, 0.01]], np.float32))
        a_, b_ = twin.warp_project(p1, ff1, q, tt, H, W), ops.warp_project(p1, ff1, q, tt, H, W)
        close(a_[0], b_[0].cpu().numpy(), atol=1e-5)
        same = torch.isclose(a_[1], b_[1], atol=1e-5).all(-1)
        assert same.float().mean() > 0.995                                  # cell flips only on cell borders
        assert torch.isclose(a_[2][same], b_[2][same], atol=1e-5).all()


def test_get_loss_and_preprocess_match_oracle():
    pm, mu = load_pkg("pwclo_model"), load_pkg("model_util")
    rng = np.random.default_rng(1)
    B = 3
    qs = [rng.normal(0, 1, (B, 4)).astype(np.float32) for _ in range(4)]
    ts = [rng.normal(0, 1, (B, 3)).astype(np.float32) for _ in range(4)]
    q_gt, t_gt = rng.normal(0, 1, (B, 4)).astype(np.float32), rng.normal(0, 1, (B, 3, 1)).astype(np.float32)
    args = [x for pair in zip(qs, ts) for x in pair]
    want = O.get_loss(*args, q_gt, t_gt, 0.0, -2.5)
    got = pm.get_loss(*[t(a) for a in args], t(q_gt), t(t_gt), torch.tensor(0.0, device=DEV), torch.tensor(-2.5, device=DEV))
    assert abs(float(got) - want) < 1e-4 * max(1.0, abs(want))
    # PreProcess: crop, per-sample augmentation of frame 1 / frame 2, q_gt / t_gt
    N = 5000
    pc1 = rng.normal(0, 15, (B, N, 3)).astype(np.float32); pc1[:, :200] = 0
    pc2 = rng.normal(0, 15, (B, N, 3)).astype(np.float32); pc2[:, -100:] = 0

    def rigid(seed):
        r = np.random.default_rng(seed)
        a = r.normal(0, 0.05, 3)
        Rz = np.array([[np.cos(a[0]), -np.sin(a[0]), 0], [np.sin(a[0]), np.cos(a[0]), 0], [0, 0, 1]])
        Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
        T = np.eye(4); T[:3, :3] = Rz @ Ry; T[:3, 3] = r.normal(0, 0.5, 3)
        return T.astype(np.float32)
    T_gt = np.stack([rigid(i) for i in range(B)])
    T_tr = np.stack([rigid(10 + i) for i in range(B)])
    T_inv = np.linalg.inv(T_tr).astype(np.float32)
    aug = np.array([1, 2, 1])
    want = O.PreProcess(pc1, pc2, T_gt, T_tr, T_inv, aug)
    got = mu.PreProcess(t(pc1), t(pc2), t(T_gt), t(T_tr), t(T_inv), aug)
    for g, w_ in zip(got, want):
        close(g, w_, atol=2e-4)


def test_forward_points_matches_oracle_composition():
    """get_model with the reference's full signature: PreProcess + input projection + pyramid."""
    model, perm_mod, synth = load_pkg("model"), load_pkg("perm"), load_pkg("synth")
    H, W, B = 64, 900, 1
    f1, f2 = synth.frame_pair(B, H, W, seed=12)
    # a raw "cloud" = the range-image points (plus padding zeros), so that re-projecting reproduces a range image
    pad = np.zeros((B, 1000, 3), np.float32)
    cloud = np.concatenate([np.concatenate([f1.reshape(B, -1, 3), pad], 1), np.concatenate([f2.reshape(B, -1, 3), pad], 1)], 1)
    cloud6 = np.concatenate([cloud, np.zeros_like(cloud)], -1)
    eye = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    net = model.PWCLONet(DEV, seed=2, perm_source=perm_mod.PermSource(fn=shuffle_fn))
    out = net.forward_points(t(cloud6), H, W, t(eye), t(eye), t(eye), aug_frame=np.array([1]))
    randomise(net.store, seed=4)
    out = net.forward_points(t(cloud6), H, W, t(eye), t(eye), t(eye), aug_frame=np.array([1]))
    n = cloud.shape[1] // 2
    p1, p2, q_gt, t_gt = O.PreProcess(cloud[:, :n], cloud[:, n:], eye, eye, eye, np.array([1]))
    # project with the product's own kernel (cell flips at borders are the projection test's business) ...
    mu = load_pkg("model_util")
    with torch.no_grad():                 # inference = the HIP projection kernel (autograd on would pick the torch twin;
        x1 = mu.ProjectPC2SphericalRing(t(p1), None, H, W)[0].cpu().numpy()      # these synthetic points sit exactly on
        x2 = mu.ProjectPC2SphericalRing(t(p2), None, H, W)[0].cpu().numpy()      # row borders, where 1 ulp of asin flips cells)
    # ... so the composition is checked exactly: get_model == pyramid(project(PreProcess(cloud))), with PreProcess
    # and q_gt/t_gt against the oracle (the pyramid itself is checked against the oracle in test_model_gpu.py)
    direct = net.forward(t(x1), t(x2))
    for g, w_ in zip(out[:9], direct):
        assert torch.equal(g, w_)
    close(out[9], q_gt, atol=1e-5)
    close(out[10], t_gt, atol=1e-5)


def test_training_steps_reduce_the_loss_and_reach_every_layer():
    model, training, synth = load_pkg("model"), load_pkg("training"), load_pkg("synth")
    torch.manual_seed(0)                                   # dropout draws
    net = model.PWCLONet(DEV, seed=3)
    tr = training.Trainer(net)
    assert len(tr.params) == 382 and tr.bucket.flat.numel() == 899134
    f1, f2 = synth.frame_pair(2, 64, 900, seed=20)
    a, b = t(f1), t(f2)
    q_gt = t(np.array([[0.99995, 0.0, 0.0, 0.01]] * 2, np.float32))
    t_gt = t(np.array([[[0.8], [0.0], [0.0]]] * 2, np.float32))
    losses = [float(tr.step(a, b, q_gt, t_gt)) for _ in range(8)]
    assert all(np.isfinite(losses)) and min(losses[4:]) < losses[0], losses
    tr.bucket.zero()
    with torch.enable_grad():
        tf_util, perm = load_pkg("tf_util"), load_pkg("perm")
        pm = load_pkg("pwclo_model")
        with tf_util.default_store(net.store), perm.default_perm_source(net.perms):
            out = pm.get_model_from_projection(a, b, True, 0.5)
        pm.get_loss(*out[:8], q_gt, t_gt, tr.w_x, tr.w_q).backward()
    dead = [n for n, p in net.store.params.items() if float(p.grad.abs().sum()) == 0.0]
    assert torch.isfinite(tr.bucket.flat).all()
    assert len(dead) <= 4, dead                    # every conv of the pyramid receives gradient
    # inference after training: folded / packed weights were invalidated and rebuilt
    out = net.forward(a, b)
    assert all(torch.isfinite(o).all() for o in out)


def test_trainer_checkpoints_resume(tmp_path):
    """save -> load into a fresh trainer -> the next step equals the original's next step (variables, moving
    statistics, loss weights, Adam moments, step count) up to the summation order of the atomic scatter-adds in the
    backward pass; the TensorFlow-bundle form restores the variables exactly."""
    model, training, synth = load_pkg("model"), load_pkg("training"), load_pkg("synth")
    f1, f2 = synth.frame_pair(1, 64, 900, seed=21)
    a, b = t(f1), t(f2)
    q_gt, t_gt = t(np.array([[0.99995, 0.0, 0.0, 0.01]], np.float32)), t(np.array([[[0.8], [0.0], [0.0]]], np.float32))

    def fresh(seed):
        return training.Trainer(model.PWCLONet(DEV, seed=seed))
    tr = fresh(3)
    torch.manual_seed(0)
    for _ in range(2):
        tr.step(a, b, q_gt, t_gt)
    tr.save(str(tmp_path / "ck.npz"))
    tr.save(str(tmp_path / "ck_tf"), tf_bundle=True)
    perm_state = copy.deepcopy(tr.net.perms)              # the visiting-order stream is not part of a checkpoint
    torch.manual_seed(5)
    want = float(tr.step(a, b, q_gt, t_gt))
    want_vars = {k: v.clone() for k, v in tr.net.store.state_dict().items()}

    tr2 = fresh(9).load(str(tmp_path / "ck.npz"))
    assert tr2.step_count == 2
    saved = dict(np.load(str(tmp_path / "ck.npz")))
    for k, v in tr2.net.store.state_dict().items():          # the restore itself is exact
        assert np.array_equal(v.cpu().numpy(), saved[k]), k
    for i, p in enumerate(tr2.params):
        assert np.array_equal(tr2.opt.state_dict()["state"][i]["exp_avg"].cpu().numpy(), saved["adam_m/%d" % i])
    tr2.net.perms = perm_state
    torch.manual_seed(5)
    got = float(tr2.step(a, b, q_gt, t_gt))
    assert abs(got - want) <= 1e-5 * max(1.0, abs(want))
    # Adam turns the summation-order noise of a (mathematically zero) gradient -- a conv bias in front of a batch
    # norm -- into a full +-lr step, so variables are compared to within two steps of lr = 1e-3
    for k, v in tr2.net.store.state_dict().items():
        assert torch.allclose(v, want_vars[k], rtol=1e-3, atol=2.1e-3), k

    tr3 = fresh(11).load(str(tmp_path / "ck_tf"))
    for k, v in tr3.net.store.state_dict().items():
        assert np.array_equal(v.cpu().numpy(), saved[k]), k
    assert float(tr3.w_q.detach()) == float(saved["w_q"])


def test_runs_at_128x2048():
    """BASELINE config 5 resolution: the re-projection sizes come from the pyramid, not from literals."""
    model, synth, perm_mod = load_pkg("model"), load_pkg("synth"), load_pkg("perm")
    f1, f2 = synth.frame_pair(1, 128, 2048, seed=30)
    net = model.PWCLONet(DEV, seed=6, perm_source=perm_mod.PermSource(fn=shuffle_fn))
    net.forward(t(f1), t(f2))
    randomise(net.store, seed=8)
    got = net.forward(t(f1), t(f2))
    want = O.get_model_from_projection(export(net.store), shuffle_fn, f1, f2)
    for g, w_ in zip(got, want):
        close(g, w_, atol=1e-4, rtol=1e-4)


def test_captured_training_step_trains():
    """Trainer(capturable=True).capture / step_graph: the whole optimisation step as one hipGraph.  Same data every
    step: the loss falls, the step counter and the device-side learning rate follow the schedule, and the graph reads
    new inputs through its static buffers (a different batch gives a different loss)."""
    model, training, synth = load_pkg("model"), load_pkg("training"), load_pkg("synth")
    torch.manual_seed(0)
    net = model.PWCLONet(DEV, seed=3)
    tr = training.Trainer(net, capturable=True)
    f1, f2 = synth.frame_pair(2, 64, 900, seed=20)
    a, b = t(f1), t(f2)
    q_gt = t(np.array([[0.99995, 0.0, 0.0, 0.01]] * 2, np.float32))
    t_gt = t(np.array([[[0.8], [0.0], [0.0]]] * 2, np.float32))
    tr.capture(a, b, q_gt, t_gt)
    start = tr.step_count
    losses = [float(tr.step_graph(a, b, q_gt, t_gt)) for _ in range(8)]
    assert tr.step_count == start + 8 and all(np.isfinite(losses)) and min(losses[4:]) < losses[0], losses
    lr = training.learning_rate(tr.step_count - 1, 2)
    assert abs(tr.opt.lr - lr) < 1e-12 and tr.opt.t == tr.step_count           # the schedule; one Adam step per optimisation step
    # ... and what the captured launch reads: tf.train.AdamOptimizer's step size lr sqrt(1 - b2^t) / (1 - b1^t) (FlatAdam's default)
    assert abs(float(tr.opt.hyper[0]) - lr * (1 - 0.999 ** tr.opt.t) ** 0.5 / (1 - 0.9 ** tr.opt.t)) < 1e-9 and float(tr.opt.hyper[1]) == 1.0
    g1, g2 = synth.frame_pair(2, 64, 900, seed=21)
    other = float(tr.step_graph(t(g1), t(g2), q_gt, t_gt))
    assert np.isfinite(other) and other != losses[-1]
    with pytest.raises(RuntimeError, match="capturable"):
        training.Trainer(model.PWCLONet(DEV, seed=3)).capture(a, b, q_gt, t_gt)


def test_the_siamese_training_batch_is_the_two_pyramids_of_the_reference():
    """tuning.train_siamese_batch (the two frames' feature pyramids as ONE 2B batch with per-frame batch-norm statistics) against the
    pyramid run once per frame with shared variables (pwclo_model.py:117-143, :143 reuse_variables): the four poses, the loss, every
    gradient and every moving average of one training forward + backward."""
    model, training, synth, tuning = load_pkg("model"), load_pkg("training"), load_pkg("synth"), load_pkg("tuning")
    tf_util, perm, pm = load_pkg("tf_util"), load_pkg("perm"), load_pkg("pwclo_model")
    f1, f2 = synth.frame_pair(2, 64, 900, seed=21)
    a, b = t(f1), t(f2)
    q_gt = t(np.array([[0.99995, 0.0, 0.0, 0.01]] * 2, np.float32))
    t_gt = t(np.array([[[0.8], [0.0], [0.0]]] * 2, np.float32))
    res = []
    for joint in (False, True):
        torch.manual_seed(0)
        net = model.PWCLONet(DEV, seed=3)
        tr = training.Trainer(net)
        tr.bucket.zero()
        with tuning.override(train_siamese_batch=joint), torch.enable_grad():
            with tf_util.default_store(net.store), perm.default_perm_source(net.perms):
                out = pm.get_model_from_projection(a, b, True, 0.5)
            loss = pm.get_loss(*out[:8], q_gt, t_gt, tr.w_x, tr.w_q)
            loss.backward()
        res.append(([o.detach().clone() for o in out[:8]], float(loss), tr.bucket.flat.clone(),
                    {k: v.clone() for k, v in net.store.buffers.items()}))
    (p0, l0, g0, m0), (p1, l1, g1, m1) = res
    # The layers are equal to rounding (tests/test_train_kernels_gpu.py::test_a_two_group_layer_is_two_calls_with_shared_variables); the
    # library GEMM of the small layers may pick another kernel for 2B rows, and a 1e-7 change of a coarse pose can move a point across a
    # projection cell at the finer levels (DESIGN.md "numerics"): tight at l3 / l2, bounded at l1 / l0 (tools/siamese_batch_check.py
    # prints the same comparison for three scenes, and the sequential form against itself).
    tol = [2e-2, 2e-2, 2e-3, 2e-3, 2e-4, 2e-4, 2e-5, 2e-5]                      # l0_q, l0_t, l1_q, l1_t, l2_q, l2_t, l3_q, l3_t
    for x, y, e in zip(p0, p1, tol):
        assert float((x - y).abs().max()) <= e * (float(x.abs().max()) + 1e-3)
    assert abs(l0 - l1) <= 2e-3 * abs(l0)
    cos = float((g0 * g1).sum() / (g0.norm() * g1.norm()))
    assert cos > 0.9995, cos
    for k in m0:
        assert float((m0[k] - m1[k]).abs().max()) <= 2e-3 * (float(m0[k].abs().max()) + 1e-2), k
