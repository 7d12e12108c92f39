"""KITTI odometry I/O and the KITTI relative-error metrics: the components on either side of the hot path
(SURVEY.md section 8(f), ranks 2 and 3).  Plain numpy on the host: none of this is GPU work.

  * read_calib / load_pair        kitti_dataset.py:38-103, :109-126 (velodyne .bin -> two zero-padded clouds,
                                  T_gt = Tr^-1 . T_diff . Tr)
  * relative_from_absolute        how ground_truth_pose/kitti_T_diff/*.npy relates to ground_truth_pose/*.txt
  * write_pred_txt                main.py:574-583 (12 floats per line)
  * sequence_errors / overall     kitti_evaluation.py:103-195 (segment lengths 100..800 m, every 10th frame)
Pose chaining of the network output lives in evaluate.pose_rows (main.py:557-572; distributed.chain_poses is the same
function on the gathered (n,7) pose log).
"""
import os

import numpy as np

SEGMENT_LENGTHS = [100, 200, 300, 400, 500, 600, 700, 800]      # kitti_evaluation.py:28
STEP_SIZE = 10                                                  # kitti_evaluation.py:147 (10 Hz, every second)


def read_calib(path):
    """kitti_dataset.py:109-126 -> dict of float arrays; 'Tr' is the 3x4 velodyne->camera transform."""
    data = {}
    with open(path) as f:
        for line in f:
            if ":" not in line:
                continue
            key, value = line.split(":", 1)
            try:
                data[key] = np.array([float(v) for v in value.split()])
            except ValueError:
                data[key] = value.strip()
    return data


def to_4x4(row12):
    return np.vstack([np.asarray(row12, dtype=np.float64).reshape(3, 4), [0.0, 0.0, 0.0, 1.0]])


def relative_from_absolute(poses_n12):
    """(n,12) absolute camera poses -> (n,12) frame-to-previous-frame transforms inv(P[i-1]) @ P[i]
    (row 0 = inv(P[0]) @ P[0] = identity), the content of ground_truth_pose/kitti_T_diff/NN_diff.npy."""
    P = [to_4x4(r) for r in np.asarray(poses_n12)]
    out = [np.eye(4)[:3].reshape(12)]
    for i in range(1, len(P)):
        out.append((np.linalg.inv(P[i - 1]) @ P[i])[:3].reshape(12))
    return np.stack(out)


def ground_truth_transform(T_diff_row12, Tr):
    """kitti_dataset.py:92-101: T_gt = Tr^-1 . T_diff . Tr (camera-frame motion expressed in the LiDAR frame)."""
    Tr = np.asarray(Tr, dtype=np.float64)
    if Tr.shape == (12,):
        Tr = to_4x4(Tr)
    elif Tr.shape == (3, 4):
        Tr = np.vstack([Tr, [0, 0, 0, 1.0]])
    return np.linalg.inv(Tr) @ to_4x4(T_diff_row12) @ Tr


def pad_cloud(points_n4, num_points=150000):
    """kitti_dataset.py:78-89: xyz of an (n,4) velodyne scan, zero-padded to (num_points,3) float64."""
    out = np.zeros((num_points, 3))
    n = min(points_n4.shape[0], num_points)
    out[:n] = points_n4[:n, :3]
    return out, n


def load_pair(root, seq, idx, T_diff, num_points=150000):
    """kitti_dataset.py:38-103 for one sample of sequence `seq` ('00'..): returns (pos2, pos1, n2, n1, T_gt) in the
    reference's order (current frame first); frame 0 is paired with itself."""
    seq_dir = os.path.join(root, seq)
    Tr = read_calib(os.path.join(seq_dir, "calib.txt"))["Tr"]
    prev = idx - 1 if idx > 0 else 0
    scan = lambda i: np.fromfile(os.path.join(seq_dir, "velodyne", "%06d.bin" % i), dtype=np.float32).reshape(-1, 4)
    pos1, n1 = pad_cloud(scan(prev), num_points)
    pos2, n2 = pad_cloud(scan(idx), num_points)
    return pos2, pos1, n2, n1, ground_truth_transform(T_diff[idx], Tr)


def write_pred_txt(path, rows_n12):
    """main.py:574-583: one pose per line, 12 floats."""
    np.savetxt(path, np.asarray(rows_n12).reshape(-1, 12), fmt="%.08f")


# ------------------------------------------------------------------------------- metrics
def trajectory_distances(poses):
    """kitti_evaluation.py:103-120: cumulative path length along the ground truth."""
    xyz = np.stack([P[:3, 3] for P in poses])
    return np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(xyz, axis=0), axis=1))])


def rotation_error(E):
    """kitti_evaluation.py:122-127."""
    d = 0.5 * (E[0, 0] + E[1, 1] + E[2, 2] - 1.0)
    return float(np.arccos(max(min(d, 1.0), -1.0)))


def translation_error(E):
    """kitti_evaluation.py:129-133."""
    return float(np.linalg.norm(E[:3, 3]))


def sequence_errors(poses_gt_n12, poses_pred_n12):
    """kitti_evaluation.py:141-181 -> list of [first_frame, r_err/len, t_err/len, len, speed]."""
    gt = [to_4x4(r) for r in np.asarray(poses_gt_n12)]
    pr = [to_4x4(r) for r in np.asarray(poses_pred_n12)]
    dist = trajectory_distances(gt)
    err = []
    for first in range(0, len(gt), STEP_SIZE):
        for length in SEGMENT_LENGTHS:
            beyond = np.nonzero(dist[first:] > dist[first] + length)[0]        # :135-139
            if beyond.size == 0:
                continue
            last = first + int(beyond[0])
            if last >= len(pr) or first >= len(pr):
                continue
            delta_gt = np.linalg.inv(gt[first]) @ gt[last]
            delta_pr = np.linalg.inv(pr[first]) @ pr[last]
            E = np.linalg.inv(delta_pr) @ delta_gt
            frames = last - first + 1.0
            err.append([first, rotation_error(E) / length, translation_error(E) / length, length, length / (0.1 * frames)])
    return err


def overall(err):
    """kitti_evaluation.py:183-195 -> (t_rel in %, r_rel in deg per 100 m), the numbers of doc/result.png."""
    if len(err) == 0:
        return float("nan"), float("nan")
    e = np.asarray(err)
    return float(e[:, 2].mean() * 100.0), float(e[:, 1].mean() / np.pi * 180.0 * 100.0)
