"""CPU: KITTI I/O + metrics against the reference's own data
(tests/golden/kitti_seq04_gt.npz = ground_truth_pose/04.txt and kitti_T_diff/04_diff.npy) and closed-form cases."""
import os

import numpy as np

from conftest import GOLDEN, load_pkg


def _seq04():
    d = np.load(os.path.join(GOLDEN, "kitti_seq04_gt.npz"))
    return d["poses"], d["diff"]


def test_relative_transforms_match_the_reference_files():
    K, D = load_pkg("kitti"), load_pkg("distributed")
    poses, diff = _seq04()
    mine = K.relative_from_absolute(poses)
    assert mine.shape == diff.shape == (271, 12)
    assert np.allclose(mine, diff, atol=2e-5)                 # the .txt holds 7 significant digits
    # chaining the relative transforms (as main.py:557-572 chains network outputs) reproduces the trajectory
    T = np.eye(4)
    for i in range(1, 271):
        T = T @ K.to_4x4(diff[i])
        assert np.allclose(T[:3].reshape(12), poses[i], atol=5e-3)


def test_metrics_closed_form_cases():
    K = load_pkg("kitti")
    poses, _ = _seq04()
    t_rel, r_rel = K.overall(K.sequence_errors(poses, poses))
    assert abs(t_rel) < 1e-9 and abs(r_rel) < 1e-4            # identical trajectories
    # a 2 % scale error on every translation -> t_rel = 2 %, no rotation error
    scaled = poses.copy()
    scaled[:, [3, 7, 11]] *= 1.02
    t_rel, r_rel = K.overall(K.sequence_errors(poses, scaled))
    assert abs(t_rel - 2.0) < 0.05 and r_rel < 1e-3
    # straight 1 m/frame drive; prediction yaws an extra 0.001 rad per frame -> r_rel = 0.1 rad / 100 m
    n = 900
    gt = np.tile(np.eye(4)[:3].reshape(12), (n, 1))
    gt[:, 11] = np.arange(n)
    pred = gt.copy()
    for i in range(n):
        a = 0.001 * i
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        pred[i, [0, 1, 2, 4, 5, 6, 8, 9, 10]] = R.reshape(9)
    _, r_rel = K.overall(K.sequence_errors(gt, pred))
    # a segment of length L (first frame strictly beyond L metres -> L+1 frames) accumulates 0.001*(L+1) rad
    expect = np.mean([(L + 1) * 0.001 / L for L in K.SEGMENT_LENGTHS]) / np.pi * 180 * 100
    assert abs(r_rel - expect) / expect < 0.02


def test_io_round_trip(tmp_path):
    K = load_pkg("kitti")
    seq = tmp_path / "04"
    (seq / "velodyne").mkdir(parents=True)
    (seq / "calib.txt").write_text("P0: 1 0 0 0 0 1 0 0 0 0 1 0\nTr: 0 -1 0 0.1 0 0 -1 0.2 1 0 0 0.3\n")
    rng = np.random.default_rng(0)
    for i in range(2):
        rng.normal(0, 10, (1000 + i, 4)).astype(np.float32).tofile(seq / "velodyne" / ("%06d.bin" % i))
    _, diff = _seq04()
    pos2, pos1, n2, n1, T_gt = K.load_pair(str(tmp_path), "04", 1, diff, num_points=1500)
    assert pos1.shape == pos2.shape == (1500, 3) and (n1, n2) == (1000, 1001)
    assert not pos1[1000:].any() and pos1[:1000].any()
    Tr = K.to_4x4(K.read_calib(str(seq / "calib.txt"))["Tr"])
    assert np.allclose(T_gt, np.linalg.inv(Tr) @ K.to_4x4(diff[1]) @ Tr)
    _, same1, _, _, _ = K.load_pair(str(tmp_path), "04", 0, diff, num_points=1500)      # frame 0 pairs with itself
    out = tmp_path / "04_pred.txt"
    K.write_pred_txt(str(out), diff[:5])
    assert np.allclose(np.loadtxt(out), diff[:5], atol=1e-8)


def test_pose_rows_on_known_motion():
    """evaluate.pose_rows (main.py:537-572): constant forward motion in the LiDAR frame becomes constant motion along
    the camera's image of LiDAR x; one row per sample, the first row is the first sample's own transform."""
    ev, kitti = load_pkg("evaluate"), load_pkg("kitti")
    TR = np.array([4.276802385584e-04, -9.999672484946e-01, -8.084491683471e-03, -1.198459927713e-02,
                   -7.210626507497e-03, 8.081198471645e-03, -9.999413164504e-01, -5.403984729748e-02,
                   9.999738645903e-01, 4.859485810390e-04, -7.206933692422e-03, -2.921968648686e-01])
    n = 6
    q = np.tile([1.0, 0, 0, 0], (n, 1))
    t = np.tile([0.8, 0, 0], (n, 1))
    rows = ev.pose_rows(q, t, TR)
    Tr = kitti.to_4x4(TR)
    step = (Tr @ np.array([0.8, 0, 0, 0.0]))[:3]           # direction of LiDAR x in the camera frame
    for i in range(n):
        assert np.allclose(rows[i].reshape(3, 4)[:, :3], np.eye(3), atol=1e-9)
        assert np.allclose(rows[i].reshape(3, 4)[:, 3], (i + 1) * step, atol=1e-9)
    assert ev.pose_rows(np.zeros((0, 4)), np.zeros((0, 3)), TR).shape == (0, 12)


def test_metrics_equal_the_reference_evaluator():
    """kitti.sequence_errors / overall against numbers produced by the REFERENCE'S OWN evaluator methods
    (kitti_evaluation.py:103-195, run in the authoring container by tests/golden/make_kitti_metrics.py) on a committed
    (ground truth, perturbed prediction) pair of sequence-04 trajectories: every per-segment row
    [first_frame, r_err/len, t_err/len, len, speed], the cumulative distances and the overall averages -- including a
    prediction that ends before the ground truth does."""
    K = load_pkg("kitti")
    d = np.load(os.path.join(GOLDEN, "kitti_metrics.npz"))
    gt, pred = d["gt"], d["pred"]
    assert np.allclose(K.trajectory_distances([K.to_4x4(r) for r in gt]), d["dist"], rtol=1e-12, atol=1e-9)
    err = np.asarray(K.sequence_errors(gt, pred))
    assert err.shape == d["err"].shape == (43, 5)
    assert np.array_equal(err[:, [0, 3]], d["err"][:, [0, 3]])                       # same segments
    assert np.allclose(err, d["err"], rtol=1e-9, atol=1e-12)
    t_rel, r_rel = K.overall(err)
    ave_t, ave_r = d["overall"]                                                        # the reference's raw averages ...
    assert abs(t_rel - ave_t * 100.0) < 1e-9 and abs(r_rel - ave_r / np.pi * 180.0 * 100.0) < 1e-9   # ... in %, deg/100 m
    short = np.asarray(K.sequence_errors(gt, pred[:200]))
    assert short.shape == d["err_short"].shape and np.allclose(short, d["err_short"], rtol=1e-9, atol=1e-12)
