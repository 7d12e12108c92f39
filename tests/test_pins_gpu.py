"""GPU: pins and guards around the operator kernels -- KAT-2 through the HIP projection, the pose head + composition in
isolation against the oracle, the fp16-range check of the split products, and the full-pyramid parity on the
true-fp32-MFMA comparison build."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import load_pkg
from oracle import ops_np as O
from util_params import close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _point(r, az_deg, el_deg):
    az, el = math.radians(az_deg), math.radians(el_deg)
    return [r * math.cos(el) * math.cos(az), r * math.cos(el) * math.sin(az), r * math.sin(el)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_kat2_through_the_hip_projection(dtype):
    """The reference's projection demo (model_util.py:449-481, KAT-2) as GEOMETRY: five points placed so that
    ProjectPC2SphericalRing puts them in cells (1,1) (1,1) (1,1) (1,3) (4,4) of a 5x5 grid with ranges 9, 7, 7, 9, 8,
    features all ones -> feature grid: cell (1,1) = 2 (both range-7 points kept and SUMMED), (1,3) = 1, (4,4) = 1, 0
    elsewhere; the xyz grid holds the sum of the two identical range-7 points in (1,1).  Oracle and HIP kernel."""
    ops = load_pkg("_ops")
    # 5x5 grid: column c <- azimuth 180 - (c + 0.5) * 72 deg; row 1 <- elevation +5 deg, row 4 <- -15 deg (model_util.py:234-242)
    pts = np.array([[_point(9.0, 72.0, 5.0), _point(7.0, 72.0, 5.0), _point(7.0, 72.0, 5.0), _point(9.0, -72.0, 5.0),
                     _point(8.0, -144.0, -15.0)]], np.float32)
    feat = np.ones((1, 5, 4), np.float32)
    want_xyz, want_feat = O.ProjectPC2SphericalRing(pts, feat, 5, 5)
    grid = np.zeros((5, 5), np.float32)
    grid[1, 1], grid[1, 3], grid[4, 4] = 2.0, 1.0, 1.0
    assert np.array_equal(want_feat[0, :, :, 0], grid)                                   # the oracle reproduces KAT-2 ...
    assert np.allclose(want_xyz[0, 1, 1], 2 * pts[0, 1]) and np.allclose(want_xyz[0, 1, 3], pts[0, 3])
    _, got_xyz, got_feat = ops.warp_project(t(pts), t(feat).to(dtype), None, None, 5, 5)  # ... and so does the kernel
    assert got_feat.dtype == dtype
    assert np.array_equal(got_feat.float().cpu().numpy(), want_feat)
    assert np.allclose(got_xyz.cpu().numpy(), want_xyz, atol=1e-6)


@pytest.mark.parametrize("coarse", [True, False])
def test_pose_head_and_composition_match_the_oracle(coarse):
    """softmax_valid -> conv1d 64->256 -> q / t heads -> normalise -> composition with the coarse pose
    (pwclo_model.py:194-208, :262-280) as ONE call of the fused pose head, against the oracle's operator chain --
    plain, with the projection buffers cleared on the side (clear=), and with the next level's warp inside (warp=)."""
    ops = load_pkg("_ops")
    rng = np.random.default_rng(21 + coarse)
    B, N, C, level = 3, 228, 64, 3 if coarse else 1
    f, w = (rng.normal(0, 1, (B, N, C)).astype(np.float32) for _ in range(2))
    xyz = rng.normal(0, 5, (B, N, 3)).astype(np.float32)
    xyz[rng.random((B, N)) < 0.2] = 0
    xyz[2] = 0                                                       # a batch element without any valid point
    names = ("l%d_big", "l%d_q_coarse", "l%d_t_coarse") if coarse else ("l%d_big", "l%d_q_det", "l%d_t_det")
    shapes = ((C, 256), (256, 4), (256, 3))
    params = {}
    for n, s in zip(names, shapes):
        params[n % level + "/weights"] = rng.normal(0, 0.1, s).astype(np.float32)
        params[n % level + "/biases"] = rng.normal(0, 0.1, s[1:]).astype(np.float32)
    q_c = O.normalise_q(rng.normal(0, 1, (B, 1, 4)).astype(np.float32))
    t_c = rng.normal(0, 1, (B, 1, 3)).astype(np.float32)
    pooled = O.softmax_valid(f, w, ~np.all(xyz == 0, -1))
    q_det, t_det = O.pose_head(params, pooled, level, coarse)
    if coarse:
        want_q, want_t = q_det[:, 0], t_det[:, 0]
    else:
        want_q, want_t = O.compose(q_det, t_det, q_c, t_c)
    P = lambda n: t(params[n])
    wb, wq, wt = (n % level for n in names)
    args = (t(f), t(w), t(xyz), P(wb + "/weights"), P(wb + "/biases"), P(wq + "/weights"), P(wq + "/biases"),
            P(wt + "/weights"), P(wt + "/biases"))
    kw = {} if coarse else dict(q_coarse=t(q_c), t_coarse=t(t_c))
    pose7 = torch.zeros((B, 7), device=DEV)
    q, tt, qn = ops.pose_head(*args, pose7=pose7, **kw)
    close(q, want_q, atol=2e-5, rtol=1e-4)
    close(tt, want_t, atol=2e-5, rtol=1e-4)
    close(qn, O.normalise_q(want_q), atol=2e-5, rtol=1e-4)
    close(pose7, np.concatenate([O.normalise_q(want_q), want_t], -1), atol=2e-5, rtol=1e-4)
    # the side jobs do not change the pose, and the projection they prepare / run equals the stand-alone one
    H, W, Cf = 8, 113, 32
    pc = rng.normal(0, 8, (B, H * W, 3)).astype(np.float32)
    pf = rng.normal(0, 1, (B, H * W, Cf)).astype(np.float32)
    want_proj = ops.warp_project(t(pc), t(pf), q, tt, H, W)
    for warp in (None, (t(pc), t(pf))):
        buf = ops.ProjectionBuffers(B, H * W, H, W, Cf, DEV)
        buf.out_xyz.fill_(float("nan")); buf.out_feat.fill_(3.0); buf.scratch.fill_(-1)
        q2, t2, qn2 = ops.pose_head(*args, clear=buf, warp=warp, **kw)
        assert torch.equal(q2, q) and torch.equal(t2, tt) and torch.equal(qn2, qn)
        got = ops.warp_project(t(pc), t(pf), q2, t2, H, W, buffers=buf)
        assert torch.equal(got[0], want_proj[0]) and torch.equal(got[1], want_proj[1])
        assert torch.allclose(got[2], want_proj[2], atol=1e-4)
    # elo_mlp_args.clear_*: a row-wise MLP launch clears the projection buffers on the side (for a pose head that is handed the partial
    # sums of softmax_valid and has no partial-sums launch of its own left to do it: tests/test_sv_ride_gpu.py) -- same rows, buffers cleared
    fused = load_pkg("fused")
    for dt in (torch.float32, torch.float16):
        a16 = tuple(x.to(dt) for x in args[:2]) + args[2:]
        buf = ops.ProjectionBuffers(B, H * W, H, W, Cf, DEV, dt)
        buf.out_xyz.fill_(float("nan")); buf.out_feat.fill_(3.0); buf.scratch.fill_(-1)
        layer = fused.PackedDense(t(rng.normal(0, 0.1, (C, 32)).astype(np.float32)), t(np.zeros(32, np.float32)), True)
        plain_rows = fused.mlp([a16[0]], [layer])
        assert torch.equal(fused.mlp([a16[0]], [layer], clear=buf), plain_rows) and buf.cleared
        torch.cuda.synchronize()
        assert float(buf.out_xyz.abs().max()) == 0 and float(buf.out_feat.float().abs().max()) == 0


def test_range_check_flags_operands_beyond_fp16():
    """The fp16 hi/lo split saturates at |x| >= 65504 instead of raising: with the range check on
    (elo_range_check / ELO_RANGE_CHECK=1) the fused kernels COUNT such operands -- gathered inputs and layer outputs
    alike --, and count nothing on a healthy run."""
    fused, L = load_pkg("fused"), load_pkg("_lib")
    if fused.fp32_mfma():
        pytest.skip("the fp32-MFMA comparison build has no split operands")
    rng = np.random.default_rng(3)
    rows, K, N = 200, 64, 64
    W = rng.normal(0, 0.2, (K, N)).astype(np.float32)
    layers = [fused.PackedDense(t(W), t(np.zeros(N, np.float32)), relu=False),
              fused.PackedDense(t(np.eye(N, dtype=np.float32)), t(np.zeros(N, np.float32)), relu=False)]
    x = rng.normal(0, 1, (rows, K)).astype(np.float32)
    prev = L.range_check(True)
    try:
        L.range_violations(t(x))                                               # reset
        fused.mlp([t(x)], layers)
        assert L.range_violations(t(x)) == 0
        big = x.copy()
        big[7, 3] = 1e5                                                        # one input operand out of range
        fused.mlp([t(big)], layers)
        assert L.range_violations(t(x)) >= 1
        scaled = [fused.PackedDense(t(W * 4e4), t(np.zeros(N, np.float32)), relu=False), layers[1]]      # (|w| < 65504: packable)
        fused.mlp([t(x)], scaled)                                              # inputs fine, the first layer's OUTPUT is not
        assert L.range_violations(t(x)) > rows
        y = fused.mlp([t(x)], layers).cpu().numpy()                            # the checked instance computes the same numbers
    finally:
        L.range_check(bool(prev))
    assert np.array_equal(y, fused.mlp([t(x)], layers).cpu().numpy())
    assert L.range_violations(t(x)) == 0                                       # off: nothing is counted


def test_range_check_guards_a_captured_graph():
    """capture(..., check_every=N): every N-th REPLAY of a lane runs the graph recorded on the checked kernel instances, so a
    scan whose coordinates leave the fp16 range of the hi/lo split is caught in production (round 3 vetted the capture
    sample only).  A healthy stream counts nothing and gives the same poses as an unchecked capture bit for bit; an
    out-of-range pair makes collect() raise; with N = 2 only every second replay is a checked one."""
    model, synth, fused = load_pkg("model"), load_pkg("synth"), load_pkg("fused")
    if fused.fp32_mfma():
        pytest.skip("the fp32-MFMA comparison build has no split operands")
    B, H, W = 1, 64, 900
    pairs = []
    for seed in (5, 6, 7, 8):
        a, b = synth.frame_pair(B, H, W, seed=seed)
        pairs.append(torch.cat([t(a), t(b)], 0))
    bad = pairs[1].clone()
    bad[0, 24:40, 380:460] *= 1.0e4                                            # a patch of points at ~2e5 m (pyramid centres among them)
    plain = model.PWCLONet(DEV, seed=1)
    plain.capture(B, H, W, lanes=1, pose_ring=8)
    guarded = model.PWCLONet(DEV, seed=1)
    guarded.capture(B, H, W, lanes=1, pose_ring=8, check_every=1)
    for net in (plain, guarded):
        net.reset_poses(0)
        for p_ in pairs:
            net.submit(0, p_)
    want = plain.collect(0).clone()
    got = guarded.collect(0).clone()                                          # healthy stream: no violation, same bits
    assert torch.equal(got, want) and float(got.abs().max()) > 0
    guarded.reset_poses(0)
    guarded.submit(0, pairs[0])
    guarded.submit(0, bad)
    with pytest.raises(RuntimeError, match="beyond the fp16 range"):
        guarded.collect(0)
    assert guarded.range_violations() == 0                                     # the counter was read and reset
    sampled = model.PWCLONet(DEV, seed=1)
    sampled.capture(B, H, W, lanes=1, pose_ring=8, check_every=2)
    sampled.reset_poses(0)
    sampled.submit(0, bad)                                                     # replay 1: the unchecked graph -- goes unseen
    sampled.submit(0, pairs[0])                                                # replay 2: checked, healthy
    assert sampled.collect(0).shape[0] == 2
    sampled.reset_poses(0)
    sampled.submit(0, pairs[0])
    sampled.submit(0, bad)                                                     # replay 4: checked
    with pytest.raises(RuntimeError, match="beyond the fp16 range"):
        sampled.collect(0)
    # two lanes: each lane's checked graph counts into the lane's OWN device word (ABI 26: `range_counter` in the fused kernels'
    # argument blocks) -- the lane that met the saturated pair raises, the healthy lane's poses stay trusted (until round 6 one
    # process-wide word tainted every lane)
    two = model.PWCLONet(DEV, seed=1)
    two.capture(B, H, W, lanes=2, pose_ring=8, check_every=1)
    for lane in (0, 1):
        two.reset_poses(lane)
    two.submit(0, pairs[0])
    two.submit(1, bad)
    torch.cuda.synchronize()
    assert two.collect(0).shape[0] == 1                                        # lane 0 saw nothing
    with pytest.raises(RuntimeError, match="beyond the fp16 range"):
        two.collect(1)
    for lane in (0, 1):
        two.reset_poses(lane)
        two.submit(lane, pairs[2])
    assert two.collect(0).shape[0] == 1 and two.collect(1).shape[0] == 1       # healthy again
    # the eager vetting forward and elo_range_violations() still use the process-wide word (no lane: range_counter = NULL)
    assert two.range_violations() == 0


def test_full_pyramid_parity_on_the_fp32_mfma_build():
    """The same full-pyramid parity test on libelo_hip_f32.so (-DELO_DENSE_F32: every product on
    v_mfma_f32_16x16x4_f32), in a child process (the library is chosen at import time)."""
    env = dict(os.environ, ELO_DENSE_F32="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_model_gpu.py"), "-q", "-x", "-m", "gpu",
                          "-k", "test_full_pyramid_matches_oracle and fused"], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "2 passed" in out.stdout, out.stdout[-500:]


def test_capture_vets_the_operand_range_and_packing_refuses_huge_weights():
    """capture(sample=pair) runs one checked forward first (PWCLONet.check_range) and refuses inputs whose matrix-core
    operands leave the fp16 range of the hi/lo split; packing a layer whose folded weight is >= 65504 raises instead of
    writing inf halves."""
    model, synth, fused, tf_util = load_pkg("model"), load_pkg("synth"), load_pkg("fused"), load_pkg("tf_util")
    f1, f2 = synth.frame_pair(1, 64, 900, seed=8)
    pair = torch.from_numpy(np.concatenate([f1, f2], 0)).to(DEV)
    net = model.PWCLONet(DEV, seed=2)
    assert net.check_range(pair[:1], pair[1:]) == 0
    net.capture(1, 64, 900, sample=pair)                                       # a sane scan: captured
    with pytest.raises(RuntimeError, match="fp16 range"):
        net.capture(1, 64, 900, sample=pair * 3e4)                             # coordinates of ~1e6 m
    store = tf_util.VariableStore(DEV, seed=0)
    with tf_util.default_store(store), torch.no_grad():
        name, W, b, bn = tf_util.dense_variables("huge", 16, 16, (1, 1), False)
        W.mul_(1e6)
        store.invalidate()
        if not fused.fp32_mfma():
            with pytest.raises(ValueError, match="fp16 hi/lo split"):
                fused.packed_layer("huge", 16, 16, bn=False)


def test_bench_measures_the_roofline_traffic_itself():
    """bench.py's roofline.traffic: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) around the launch the leg times;
    the result is within a factor of the committed passes of the same kernel (profiles/r04_pmc) and far below the
    algorithmic bytes (the fused kernel never materialises the operator-boundary tensors)."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    got, why = bench._live_traffic(1, "f32")
    assert got is not None, why
    committed = bench._pmc_traffic("cv1_kernel", 1, "f32")
    assert 0.5 * committed <= got <= 2.0 * committed, (got, committed, why)
    assert got < 16588800 // 2
