"""Sequence evaluation: KITTI scans in, `NN_pred.txt` and the KITTI relative errors out -- what
main.py:459-600 `eval_one_epoch` does around the hot path, without the TF session and the subprocess call to
kitti_evaluation.py.

    rows, (t_rel, r_rel) = run_sequence(net, root, "04", T_diff, out_dir="results")

Per sample i of a sequence the dataset hands over (scan i, scan i-1) -- sample 0 pairs scan 0 with itself
(kitti_dataset.py:66-70) -- and the network predicts the motion of the pair in the LiDAR frame; the trajectory is
the running product of  Tr . [R(q)|t] . Tr^-1  (camera frame), ONE row per sample: row 0 is sample 0's own
(near-identity) prediction, not a prepended identity (main.py:557-572).
"""
import os

import numpy as np
import torch

from . import kitti
from .distributed import quat2mat


def pose_rows(q_n4, t_n3, Tr):
    """main.py:537-572: (n,4) quaternions + (n,3) translations (LiDAR frame) -> (n,12) chained camera-frame poses."""
    Tr = np.asarray(Tr, dtype=np.float64)
    if Tr.shape != (4, 4):
        Tr = kitti.to_4x4(Tr.reshape(12))
    Tr_inv = np.linalg.inv(Tr)
    T_final, rows = None, []
    for q, t in zip(np.asarray(q_n4, dtype=np.float64), np.asarray(t_n3, dtype=np.float64)):
        TT = np.eye(4)
        TT[:3, :3] = quat2mat(q.reshape(4))
        TT[:3, 3] = t.reshape(3)
        TT = Tr @ TT @ Tr_inv
        T_final = TT if T_final is None else T_final @ TT
        rows.append(T_final[:3, :].reshape(12).copy())
    return np.stack(rows) if rows else np.zeros((0, 12))


def predict_sequence(net, root, seq, T_diff, H_input=64, W_input=1800, batch_size=1, num_points=150000, frames=None,
                     lanes=0):
    """Run the network over samples `frames` (default: all scans found) of sequence `seq`; returns (q (n,4), t (n,3))
    = the l0 pose of every sample, in sample order.  Batches are padded by repeating the last sample
    (main.py:497-509 keeps stale rows instead; either way the padding rows are dropped).
    `lanes` > 0: through `lanes` hipGraphs recorded from the raw clouds on (`PWCLONet.capture(num_points=...)`: input
    stage + pyramid in one replay, several batches in flight while the host reads the next scans) instead of the eager
    `forward_points`; same results (no augmentation in evaluation: the identity T_trans of the eager call is a no-op)."""
    seq_dir = os.path.join(root, seq)
    if frames is None:
        frames = range(len([f for f in os.listdir(os.path.join(seq_dir, "velodyne")) if f.endswith(".bin")]))
    frames = list(frames)
    dev = net.device
    if lanes > 0:
        return _predict_sequence_lanes(net, root, seq, T_diff, H_input, W_input, batch_size, num_points, frames, lanes)
    eye = torch.eye(4, dtype=torch.float32, device=dev).repeat(batch_size, 1, 1)      # main.py:308-309: no augmentation
    qs, ts = [], []
    for start in range(0, len(frames), batch_size):
        chunk = frames[start:start + batch_size]
        cloud = np.zeros((batch_size, 2 * num_points, 3), np.float32)
        T_gt = np.zeros((batch_size, 4, 4), np.float32)
        for j in range(batch_size):
            pos2, pos1, _n2, _n1, T = kitti.load_pair(root, seq, chunk[min(j, len(chunk) - 1)], T_diff, num_points)
            cloud[j, :num_points], cloud[j, num_points:], T_gt[j] = pos2, pos1, T     # main.py:316-320
        out = net.forward_points(torch.from_numpy(cloud).to(dev), H_input, W_input, torch.from_numpy(T_gt).to(dev),
                                 eye, eye, is_training=False, aug_frame=np.ones(batch_size, np.int64))
        qs.append(out[0][:len(chunk)].reshape(-1, 4).cpu().numpy())
        ts.append(out[1][:len(chunk)].reshape(-1, 3).cpu().numpy())
    return np.concatenate(qs), np.concatenate(ts)


def _predict_sequence_lanes(net, root, seq, T_diff, H_input, W_input, batch_size, num_points, frames, lanes):
    dev = net.device
    net.capture(batch_size, H_input, W_input, lanes=lanes, num_points=num_points)
    chunks = [frames[s:s + batch_size] for s in range(0, len(frames), batch_size)]
    qs, ts = [None] * len(chunks), [None] * len(chunks)
    pinned = [torch.empty((batch_size, 2 * num_points, 3), dtype=torch.float32).pin_memory() for _ in range(lanes)]

    def collect(ci):                                         # the lane's stream has finished chunk ci
        lane = ci % lanes
        net.lane_stream(lane).synchronize()
        pose = net.lane_pose(lane)[:len(chunks[ci])].cpu().numpy()        # (b,7) = [q_norm | t] of l0
        qs[ci], ts[ci] = pose[:, :4].copy(), pose[:, 4:].copy()

    for ci, chunk in enumerate(chunks):
        lane = ci % lanes
        if ci >= lanes:
            collect(ci - lanes)                              # frees the lane (and its pinned staging buffer)
        cloud = pinned[lane].numpy()
        for j in range(batch_size):
            pos2, pos1, _n2, _n1, _T = kitti.load_pair(root, seq, chunk[min(j, len(chunk) - 1)], T_diff, num_points)
            cloud[j, :num_points], cloud[j, num_points:] = pos2, pos1                      # main.py:316-320
        net.submit_points(lane, pinned[lane])                # async H2D into the lane's cloud buffer + one replay
    for ci in range(max(0, len(chunks) - lanes), len(chunks)):
        collect(ci)
    return np.concatenate(qs), np.concatenate(ts)


def run_sequence(net, root, seq, T_diff, poses_gt=None, out_dir=None, **kw):
    """predict_sequence -> pose_rows -> `<out_dir>/<seq>_pred.txt` (main.py:574-583) -> KITTI errors against
    `poses_gt` ((n,12) absolute camera poses, e.g. ground_truth_pose/<seq>.txt) if given.
    Returns (rows (n,12), (t_rel %, r_rel deg/100m) or None)."""
    q, t = predict_sequence(net, root, seq, T_diff, **kw)
    Tr = kitti.read_calib(os.path.join(root, seq, "calib.txt"))["Tr"]
    rows = pose_rows(q, t, Tr)
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        kitti.write_pred_txt(os.path.join(out_dir, "%s_pred.txt" % seq), rows)
    score = None
    if poses_gt is not None:
        score = kitti.overall(kitti.sequence_errors(np.asarray(poses_gt)[:len(rows)], rows))
    return rows, score
