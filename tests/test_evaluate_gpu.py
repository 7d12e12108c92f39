"""GPU: scans on disk -> predicted trajectory file -> KITTI errors (evaluate.py, the main.py:459-600 loop)."""
import os

import numpy as np
import pytest
import torch

from conftest import load_pkg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TR = np.array([4.276802385584e-04, -9.999672484946e-01, -8.084491683471e-03, -1.198459927713e-02,
               -7.210626507497e-03, 8.081198471645e-03, -9.999413164504e-01, -5.403984729748e-02,
               9.999738645903e-01, 4.859485810390e-04, -7.206933692422e-03, -2.921968648686e-01])   # KITTI-style Tr


def _write_sequence(root, seq, n_frames, H, W):
    synth, kitti = load_pkg("synth"), load_pkg("kitti")
    d = os.path.join(root, seq, "velodyne")
    os.makedirs(d)
    with open(os.path.join(root, seq, "calib.txt"), "w") as f:
        f.write("P0: 1 0 0 0 0 1 0 0 0 0 1 0\nTr: " + " ".join("%.12e" % v for v in TR) + "\n")
    poses = []
    for i in range(n_frames):
        img = synth.range_image(H, W, seed=50 + i, yaw=0.01 * i, shift=(0.8 * i, 0.0, 0.0))
        pts = img.reshape(-1, 3)
        pts = pts[np.any(pts != 0, -1)]
        np.concatenate([pts, np.ones((len(pts), 1), np.float32)], 1).astype(np.float32).tofile(
            os.path.join(d, "%06d.bin" % i))
        P = np.eye(4)
        P[2, 3] = 0.8 * i                                   # camera z forward
        poses.append(P[:3].reshape(12))
    poses = np.stack(poses)
    return poses, kitti.relative_from_absolute(poses)


def test_sequence_pipeline_from_scans_to_errors(tmp_path):
    model, ev, kitti = load_pkg("model"), load_pkg("evaluate"), load_pkg("kitti")
    H, W, n = 64, 900, 5
    poses, T_diff = _write_sequence(str(tmp_path), "04", n, H, W)
    net = model.PWCLONet(DEV, seed=0)
    kw = dict(H_input=H, W_input=W, num_points=H * W)
    rows, score = ev.run_sequence(net, str(tmp_path), "04", T_diff, poses_gt=poses, out_dir=str(tmp_path / "out"),
                                  batch_size=2, **kw)
    assert rows.shape == (n, 12) and np.isfinite(rows).all()
    back = np.loadtxt(str(tmp_path / "out" / "04_pred.txt"))
    assert back.shape == (n, 12) and np.allclose(back, rows, atol=1e-7)       # '%.08f'
    assert score is not None and len(score) == 2                            # too short for a 100 m segment: nan
    # the batch of two equals one pair at a time (the padding row of the last batch is dropped)
    q2, t2 = ev.predict_sequence(net, str(tmp_path), "04", T_diff, batch_size=2, **kw)
    q1, t1 = ev.predict_sequence(net, str(tmp_path), "04", T_diff, batch_size=1, **kw)
    assert q2.shape == (n, 4) and np.allclose(q1, q2, atol=1e-5) and np.allclose(t1, t2, atol=1e-4)
    # through graph lanes recorded from the raw clouds on: same poses as the eager forward_points path
    net2 = model.PWCLONet(DEV, seed=0)
    q3, t3 = ev.predict_sequence(net2, str(tmp_path), "04", T_diff, batch_size=1, lanes=2, **kw)
    assert np.array_equal(q3, q1) and np.array_equal(t3, t1)
    # row 0 is sample 0's own prediction (scan 0 paired with itself), rows chain as T_i = T_{i-1} . Tr [R|t] Tr^-1
    dist = load_pkg("distributed")
    Tr = kitti.to_4x4(TR)
    T = np.eye(4)
    for i in range(n):
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = dist.quat2mat(q2[i].astype(np.float64)), t2[i]
        T = T @ (Tr @ M @ np.linalg.inv(Tr))
        assert np.allclose(T[:3].reshape(12), rows[i], atol=1e-6)
    # a self-pair has no motion to explain: identical inputs for both frames
    pos2, pos1, n2, n1, T_gt = kitti.load_pair(str(tmp_path), "04", 0, T_diff, H * W)
    assert n1 == n2 and np.array_equal(pos1, pos2) and np.allclose(T_gt, np.eye(4))
