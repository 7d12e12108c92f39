"""GPU: softmax_valid's partial sums riding on the launch that produces the pose head's inputs (elo_mlp_args.sv_*,
mlp_sv_kernel; elo_pose_head_args.ready_parts).  The row-wise MLP outputs must be the bits of the plain launch, the pose
must equal the oracle's softmax_valid -> pose head -> composition (model_util.py:319-343, pwclo_model.py:262-280) on those
outputs and the two-launch form to summation order; launch counters prove which kernel ran."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_pkg
from oracle import ops_np as O
from util_params import close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _rides(reset=False):
    lib = load_pkg("_lib")
    n = ctypes.c_ulonglong(0)
    lib.check(lib.lib().elo_debug_sv_ride_launches(ctypes.byref(n), 1 if reset else 0))
    return n.value


def _tiles(B, N, jobs):
    """Row tiles per batch element of a launch over B x N rows (elo_tuning.small_tile_units: 16-row tiles while 32-row tiles would
    give fewer workgroups than that)."""
    units = -(-B * N // 32) * jobs
    tile = 16 if units < load_pkg("_lib").get_tuning()["small_tile_units"] else 32
    return -(-N // tile)


def _head(rng, level=1):
    names = ("l%d_big", "l%d_q_det", "l%d_t_det")
    params = {}
    for n, s in zip(names, ((64, 256), (256, 4), (256, 3))):
        params[n % level + "/weights"] = rng.normal(0, 0.1, s).astype(np.float32)
        params[n % level + "/biases"] = rng.normal(0, 0.1, s[1:]).astype(np.float32)
    order = [n % level + k for n in names for k in ("/weights", "/biases")]
    return params, [t(params[k]) for k in order]


def _cloud(rng, B, N, empty_element):
    xyz = rng.normal(0, 5, (B, N, 3)).astype(np.float32)
    xyz[rng.random((B, N)) < 0.2] = 0
    if empty_element and B > 1:
        xyz[B - 1] = 0                                   # a batch element without any valid point
    return xyz


def _want(params, f, w, xyz, q_c, t_c, level=1):
    pooled = O.softmax_valid(f, w, ~np.all(xyz == 0, -1))
    q_det, t_det = O.pose_head(params, pooled, level, False)
    return O.compose(q_det, t_det, q_c, t_c)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
@pytest.mark.parametrize("B,N,C", [(1, 228, 64), (1, 904, 32), (1, 3600, 16), (2, 450, 32), (3, 57, 64)])
def test_paired_two_stage_mlp_leaves_softmax_valids_partial_sums(B, N, C, dt):
    """A refinement level's tail: set-upconv stage 2 + predictor of the embedding-mask (job a: the logits) and the embedding
    (job b: the features) branch in ONE 512-thread launch, two 4-wave groups side by side, plus the per-tile triples."""
    fused, tf_util, ops = load_pkg("fused"), load_pkg("tf_util"), load_pkg("_ops")
    rng = np.random.default_rng(B * 1000 + N)
    store = tf_util.VariableStore(DEV, seed=N)
    r = lambda *s: t(rng.normal(0, 1, s).astype(np.float32)).to(dt)

    def job(tag):
        with tf_util.default_store(store), torch.no_grad(), tf_util.variable_scope("sv_ride_%s" % tag):
            P = fused.packed_layer
            layers = [P("a0", 64 + C, 128), P("a1", 128, 64)]
            layers2 = [P("b0", C + 64 + 64, 128, row_order=fused.stage2_row_order(C, 64, 64)), P("b1", 128, 64)]
            for p_ in layers + layers2:
                p_.b.copy_(torch.from_numpy(rng.normal(0, 0.1, p_.b.shape).astype(np.float32)))
        return dict(sources=[r(B, N, 64), r(B, N, C)], layers=layers, before=r(B, N, C), after=r(B, N, 64), layers2=layers2)
    ja, jb = job("w"), job("c")
    xyz = _cloud(rng, B, N, empty_element=True)
    params, head = _head(rng)
    q_c = O.normalise_q(rng.normal(0, 1, (B, 1, 4)).astype(np.float32))
    t_c = rng.normal(0, 1, (B, 1, 3)).astype(np.float32)
    kw = dict(q_coarse=t(q_c), t_coarse=t(t_c))

    (o1a, weight), (o1b, predict) = fused.mlp2_pair(ja, jb)                       # the plain paired launch
    _rides(reset=True)
    sv = ops.SvPartials(t(xyz))
    H, W, Cf = 8, 29, 16
    buf = ops.ProjectionBuffers(B, H * W, H, W, Cf, DEV, dt)
    buf.out_xyz.fill_(float("nan")); buf.out_feat.fill_(3.0); buf.scratch.fill_(-1)
    (r1a, r_weight), (r1b, r_predict) = fused.mlp2_pair(ja, jb, clear=buf, sv=sv)
    assert _rides() == 1 and sv.parts == _tiles(B, N, 2) and buf.cleared
    for got, want in ((r1a, o1a), (r_weight, weight), (r1b, o1b), (r_predict, predict)):
        assert torch.equal(got, want)                                             # same layers, same order: same bits

    two = ops.pose_head(predict, weight, t(xyz), *head, **kw)                      # partial-sums launch + head
    pc, pf = r(B, H * W, 3) * 8, r(B, H * W, Cf)
    one = ops.pose_head(r_predict, r_weight, t(xyz), *head, clear=buf, warp=(pc.float(), pf), partials=sv, **kw)
    want_q, want_t = _want(params, predict.float().cpu().numpy(), weight.float().cpu().numpy(), xyz, q_c, t_c)
    for got, ref, want in zip(one, two, (want_q, want_t, O.normalise_q(want_q))):
        close(got, want, atol=2e-5, rtol=1e-4)                                     # the oracle on the stored rows
        close(got, ref.cpu().numpy(), atol=2e-6, rtol=1e-5)                        # the two-launch form: summation order only
    # the projection the head ran inside its launch on the buffers the MLP launch cleared = the stand-alone one
    want_proj = ops.warp_project(pc.float(), pf, one[0], one[1], H, W)
    got = ops.warp_project(pc.float(), pf, one[0], one[1], H, W, buffers=buf)
    assert torch.equal(got[0], want_proj[0]) and torch.equal(got[1], want_proj[1])
    assert torch.allclose(got[2].float(), want_proj[2].float(), atol=1e-3 if dt == torch.float16 else 1e-4)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
@pytest.mark.parametrize("B,N", [(1, 58), (2, 58), (1, 512)])
def test_single_mlp_leaves_softmax_valids_partial_sums(B, N, dt):
    """The coarse level: the predictor's output are the logits, the features are an existing tensor (SvPartials.feature)."""
    fused, tf_util, ops = load_pkg("fused"), load_pkg("tf_util"), load_pkg("_ops")
    rng = np.random.default_rng(N + B)
    store = tf_util.VariableStore(DEV, seed=N)
    r = lambda *s: t(rng.normal(0, 1, s).astype(np.float32)).to(dt)
    with tf_util.default_store(store), torch.no_grad(), tf_util.variable_scope("sv_ride_single"):
        layers = [fused.packed_layer("p0", 128 + 64, 128), fused.packed_layer("p1", 128, 64)]
    srcs = [r(B, N, 128), r(B, N, 64)]
    feature = srcs[1]
    xyz = _cloud(rng, B, N, empty_element=False)
    params, head = _head(rng, level=3)
    weight = fused.mlp(srcs, layers)
    _rides(reset=True)
    sv = ops.SvPartials(t(xyz), feature)
    r_weight = fused.mlp(srcs, layers, sv=sv)
    assert _rides() == 1 and sv.parts == _tiles(B, N, 1) and torch.equal(r_weight, weight)
    two = ops.pose_head(feature, weight, t(xyz), *head)
    one = ops.pose_head(feature, r_weight, t(xyz), *head, partials=sv)
    pooled = O.softmax_valid(feature.float().cpu().numpy(), weight.float().cpu().numpy(), ~np.all(xyz == 0, -1))
    names = {"l3_q_coarse": "l3_q_det", "l3_t_coarse": "l3_t_det"}
    coarse_params = dict(params, **{k.replace(v, c): params[k] for c, v in names.items() for k in list(params) if k.startswith(v)})
    q_det, t_det = O.pose_head(coarse_params, pooled, 3, True)
    for got, ref, want in zip(one, two, (q_det[:, 0], t_det[:, 0], O.normalise_q(q_det[:, 0]))):
        close(got, want, atol=2e-5, rtol=1e-4)
        close(got, ref.cpu().numpy(), atol=2e-6, rtol=1e-5)


def test_the_ride_is_refused_where_it_cannot_be_taken():
    """elo_mlp_sv_parts == 0 (a 32-wide final layer; the register-resident chain regime): fused.mlp* leave sv.parts at 0 and do
    NOT clear on the side, so the pose head runs its own partial-sums launch; sv_* set by hand on such a launch is an error."""
    fused, tf_util, ops, lib = load_pkg("fused"), load_pkg("tf_util"), load_pkg("_ops"), load_pkg("_lib")
    rng = np.random.default_rng(5)
    store = tf_util.VariableStore(DEV, seed=5)
    r = lambda *s: t(rng.normal(0, 1, s).astype(np.float32))
    with tf_util.default_store(store), torch.no_grad(), tf_util.variable_scope("sv_ride_refused"):
        layers = [fused.packed_layer("p0", 64, 32)]
    B, N = 1, 100
    sv = ops.SvPartials(t(_cloud(rng, B, N, False)), r(B, N, 64))
    buf = ops.ProjectionBuffers(B, 4 * 5, 4, 5, 0, DEV)
    _rides(reset=True)
    fused.mlp([r(B, N, 64)], layers, clear=buf, sv=sv)
    assert _rides() == 0 and sv.parts == 0 and not buf.cleared
    a, _out, _keep = fused._mlp_args([r(B, N, 64)], layers)
    a.sv_npoints, a.sv_scratch, a.sv_xyz, a.sv_feature = N, sv.scratch.data_ptr(), sv.xyz.data_ptr(), sv.feature.data_ptr()
    with pytest.raises(lib.EloError, match="elo_mlp_sv_parts returns 0"):
        lib.call("elo_mlp_fused", a, sv.scratch)


@pytest.mark.parametrize("H,W", [(64, 1800), (128, 2048)])
@pytest.mark.parametrize("feat", [torch.float32, torch.float16])
def test_a_forward_with_the_ride_equals_the_forward_without(feat, H, W):
    """Whole pyramid at batch 1, 64 x 1800 and 128 x 2048: four launches less (one per level).  The coarse pose agrees to summation order;
    the refinement levels follow it through DISCRETE decisions (a warped point's projection cell, a neighbour entering a
    window: tests/test_parity_flips_gpu.py), so they are held to a loose bound here -- their parity against the oracle,
    level by level on the same coarse poses, is tests/test_model_gpu.py's (which runs with the ride on: the default)."""
    from util_params import randomise, shuffle_fn
    model, synth, tuning, perm = load_pkg("model"), load_pkg("synth"), load_pkg("tuning"), load_pkg("perm")
    f1, f2 = synth.frame_pair(1, H, W, seed=3)                  # (128 x 2048: the ride at l3 / l2 / l1; l1 = 2048 points)
    both = torch.from_numpy(np.concatenate([f1, f2], 0)).to(DEV)
    net = model.PWCLONet(DEV, seed=5, perm_source=perm.PermSource(fn=shuffle_fn), feature_dtype=feat)
    net.forward(both[:1], both[1:])
    randomise(net.store, seed=7)
    outs = {}
    for ride in (True, False):
        with tuning.override(sv_ride=ride):
            _rides(reset=True)
            outs[ride] = [x.clone() for x in net.forward(both[:1], both[1:])[:8]]
            # (128 x 2048: the l0 row-wise MLP is 2 x 8192 rows -- the register-resident kernel's regime: no ride there)
            assert _rides() == ((4 if (H, W) == (64, 1800) else 3) if ride else 0)
    for i, (a, b) in enumerate(zip(outs[True], outs[False])):
        assert torch.isfinite(a).all()
        tight = i >= 6                                       # (l3_q, l3_t)
        assert torch.allclose(a, b, atol=2e-5 if tight else 2e-2, rtol=1e-4), (i, a, b)


@pytest.mark.parametrize("feat", [torch.float32, torch.float16])
@pytest.mark.parametrize("B", [1, 2])
def test_the_chain_kernels_heterogeneous_launch_leaves_a_forward_unchanged(B, feat):
    """cv1_setconv_rr_kernel (elo_cv_stage1_setconv_chain: cost-volume stage 1 from the select-k pre-pass + the level's two
    set-upconv stage-1 jobs as ONE launch of register-resident workgroups, taken below the throughput batch where both are chain
    forms -- l0 of a 64 x 1800 forward) runs the plain kernels' bodies: every pose output of the forward is BIT-identical
    with tuning.chain_pair off, and the launch counter says the kernel ran."""
    from util_params import randomise, shuffle_fn
    model, synth, tuning, perm, lib = load_pkg("model"), load_pkg("synth"), load_pkg("tuning"), load_pkg("perm"), load_pkg("_lib")
    if load_pkg("fused").fp32_mfma():
        pytest.skip("the fp32-MFMA comparison build has no register-resident kernels")
    f1, f2 = synth.frame_pair(B, 64, 1800, seed=11)
    both = torch.from_numpy(np.concatenate([f1, f2], 0)).to(DEV)
    net = model.PWCLONet(DEV, seed=5, perm_source=perm.PermSource(fn=shuffle_fn), feature_dtype=feat)
    net.forward(both[:B], both[B:])
    randomise(net.store, seed=7)

    def pairs(reset=False):
        n = ctypes.c_ulonglong(0)
        lib.check(lib.lib().elo_debug_chain_pair_launches(ctypes.byref(n), 1 if reset else 0))
        return n.value
    outs = {}
    for on in (True, False):
        with tuning.override(chain_pair=on):
            pairs(reset=True)
            outs[on] = [x.clone() for x in net.forward(both[:B], both[B:])[:8]]
            assert pairs() == (1 if on else 0)                        # l0 (the smaller levels ride on the tile kernels' merged launch)
    for a, b in zip(outs[True], outs[False]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_the_native_submit_replays_what_torch_replays():
    """model.submit through elo_graph_submit (one C call: hipMemcpyAsync + hipGraphLaunch on the raw exec handle; tuning
    native_submit) against torch's copy_ + CUDAGraph.replay() on the same lanes: the same poses bit for bit, pair after pair,
    with the in-place form (no copy) and the checked graph every second replay in between."""
    model, synth, tuning = load_pkg("model"), load_pkg("synth"), load_pkg("tuning")
    pairs = []
    for i in range(6):
        f1, f2 = synth.frame_pair(1, 64, 1800, seed=70 + i)
        pairs.append(torch.from_numpy(np.concatenate([f1, f2], 0)).to(DEV))
    got = {}
    for native in (True, False):
        with tuning.override(native_submit=native):
            net = model.PWCLONet(DEV, seed=3)
            net.capture(1, 64, 1800, lanes=2, check_every=2)
            assert (net._lanes[0]["native"] is not None) == native
            rows = []
            for i, p in enumerate(pairs):
                out = net.submit(i % 2, p)
                torch.cuda.synchronize()
                rows.append(torch.cat([out[0].reshape(-1), out[1].reshape(-1)]).clone())
            net.lane_input(0).copy_(pairs[3])                    # in place: no copy in the submit
            torch.cuda.synchronize()
            out = net.submit(0)
            torch.cuda.synchronize()
            rows.append(torch.cat([out[0].reshape(-1), out[1].reshape(-1)]).clone())
            net.collect(0); net.collect(1)                       # (no range violation on healthy inputs)
            got[native] = rows
    for a, b in zip(got[True], got[False]):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    assert torch.equal(got[True][6], got[True][3])               # the in-place replay of pair 3 = the copied one


@pytest.mark.parametrize("native", [True, False])
def test_submit_waits_for_the_producer_of_its_input(native):
    """submit()'s default ordering (the contract replaced is the reference's synchronous sess.run(feed_dict=...), main.py:372-381):
    the pair is filled by work still QUEUED on the caller's current stream -- a long spin kernel, then the copy that writes the real
    range images over zeros -- and submitted immediately.  The lane's stream must wait for that work (elo_graph_submit's
    hipEventRecord + hipStreamWaitEvent on the native path, Event.record / Stream.wait_event through torch): the pose is the one of
    a synchronised submit, bit for bit.  With ready=False (the caller owns the ordering) the same sequence reads the zeros: the test
    asserts that too, so it would have failed on the unordered submit of round 5.  submit_points likewise."""
    model, synth, tuning = load_pkg("model"), load_pkg("synth"), load_pkg("tuning")
    f1, f2 = synth.frame_pair(1, 64, 1800, seed=91)
    real = torch.from_numpy(np.concatenate([f1, f2], 0)).to(DEV)
    with tuning.override(native_submit=native):
        net = model.PWCLONet(DEV, seed=3)
        net.capture(1, 64, 1800, lanes=2)
        assert (net._lanes[0]["native"] is not None) == native
        flat = lambda out: torch.cat([out[0].reshape(-1), out[1].reshape(-1)]).clone()
        out = net.submit(0, real)
        torch.cuda.synchronize()
        want = flat(out)
        out = net.submit(0, torch.zeros_like(real))
        torch.cuda.synchronize()
        zeros = flat(out)
        assert not torch.equal(want, zeros)

        def late_input():
            x = torch.zeros_like(real)
            torch.cuda.synchronize()
            torch.cuda._sleep(200_000_000)                       # ~0.1 s of spinning on the current stream ...
            x.copy_(real, non_blocking=True)                     # ... in front of the kernel that produces the input
            return x

        out = net.submit(1, late_input())                  # default: ordered behind the current stream; x dies here (record_stream)

        torch.cuda.synchronize()

        got = flat(out)
        assert torch.equal(got, want)
        ev = torch.cuda.Event()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):                            # a producer on some other stream, handed over as an event
            x = late_input()
            ev.record()
        out = net.submit(1, x, ready=ev)
        torch.cuda.synchronize()
        got = flat(out)
        assert torch.equal(got, want)
        x = late_input()
        out = net.submit(1, x, ready=False)                # the caller owns the ordering -- and here does not provide it
        torch.cuda.synchronize()
        got = flat(out)
        assert torch.equal(got, zeros)
        # in place: lane_input() written on the current stream behind a spin, submit(lane) without a copy
        torch.cuda.synchronize()
        torch.cuda._sleep(200_000_000)
        net.lane_input(0).copy_(real, non_blocking=True)
        out = net.submit(0)
        torch.cuda.synchronize()
        got = flat(out)
        assert torch.equal(got, want)


def test_submit_points_waits_for_the_producer_of_its_cloud():
    model = load_pkg("model")
    rng = np.random.default_rng(5)
    P = 20000
    az, el, r = rng.uniform(-np.pi, np.pi, (1, 2 * P)), np.deg2rad(rng.uniform(-24.8, 2.0, (1, 2 * P))), rng.uniform(2.0, 30.0, (1, 2 * P))
    cloud = torch.from_numpy(np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], -1).astype(np.float32)).to(DEV)
    net = model.PWCLONet(DEV, seed=3)
    net.capture(1, 64, 1800, lanes=1, num_points=P)
    flat = lambda out: torch.cat([out[0].reshape(-1), out[1].reshape(-1)]).clone()
    out = net.submit_points(0, cloud)
    torch.cuda.synchronize()
    want = flat(out)
    x = torch.zeros_like(cloud)
    torch.cuda.synchronize()
    torch.cuda._sleep(200_000_000)
    x.copy_(cloud, non_blocking=True)
    out = net.submit_points(0, x)
    torch.cuda.synchronize()
    got = flat(out)
    assert torch.equal(got, want)
