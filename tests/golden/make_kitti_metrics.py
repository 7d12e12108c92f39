"""Generate tests/golden/kitti_metrics.npz: KITTI relative-error numbers computed BY THE REFERENCE'S OWN evaluator code
(kitti_evaluation.py:103-195: trajectoryDistances, lastFrameFromSegmentLength, rotationError, translationError,
calcSequenceErrors, computeOverallErr) for a committed (ground truth, prediction) trajectory pair.

Run in the authoring container only (needs /root/reference):    python tests/golden/make_kitti_metrics.py

The evaluator module cannot be imported as shipped: besides matplotlib it imports `tools.transformations` and
`tools.pose_evaluation_utils`, a package that is not in the reference repository.  Neither is
used by the six methods above (they serve the plotting / quaternion-file paths), so this script registers EMPTY
placeholder modules under those names for the import and then calls the reference methods unchanged on an instance
made without __init__ (which only globs result directories).  No reference source is copied: the fixture holds data.

Trajectories: ground truth = ground_truth_pose/04.txt (271 poses, 394 m); prediction = the same relative motions with
a seeded perturbation (rotation noise 0.15 deg, translation noise 2 cm, 0.5 % scale drift per step), chained.
"""
import os
import sys
import types

import numpy as np

REF = os.environ.get("ELO_REFERENCE_DIR", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_evaluator():
    tools = types.ModuleType("tools")
    tr = types.ModuleType("tools.transformations")
    pe = types.ModuleType("tools.pose_evaluation_utils")
    pe.quat_pose_to_mat = None
    tools.transformations, tools.pose_evaluation_utils = tr, pe
    sys.modules.update({"tools": tools, "tools.transformations": tr, "tools.pose_evaluation_utils": pe})
    sys.path.insert(0, REF)
    import kitti_evaluation
    ev = object.__new__(kitti_evaluation.kittiOdomEval)
    ev.lengths = [100, 200, 300, 400, 500, 600, 700, 800]          # kitti_evaluation.py:28-29
    ev.num_lengths = len(ev.lengths)
    return ev


def rot(axis, angle):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def to44(r):
    T = np.eye(4)
    T[:3, :] = np.asarray(r, np.float64).reshape(3, 4)
    return T


def main():
    gt_rows = np.loadtxt(os.path.join(REF, "ground_truth_pose", "04.txt"))
    gt = [to44(r) for r in gt_rows]
    rng = np.random.default_rng(2024)
    pred = [gt[0].copy()]
    for i in range(1, len(gt)):
        step = np.linalg.inv(gt[i - 1]) @ gt[i]
        noise = np.eye(4)
        noise[:3, :3] = rot(rng.normal(size=3), np.deg2rad(0.15) * rng.normal())
        noise[:3, 3] = 0.02 * rng.normal(size=3)
        step[:3, 3] *= 1.005
        pred.append(pred[-1] @ step @ noise)
    ev = reference_evaluator()
    poses_gt, poses_pred = dict(enumerate(gt)), dict(enumerate(pred))
    err = ev.calcSequenceErrors(poses_gt, poses_pred)
    ave_t, ave_r = ev.computeOverallErr(err)
    # a SHORTER prediction (the evaluator skips segments that run past it, :162-163)
    short = {k: v for k, v in poses_pred.items() if k < 200}
    err_short = ev.calcSequenceErrors(poses_gt, short)
    out = os.path.join(HERE, "kitti_metrics.npz")
    np.savez_compressed(out, gt=gt_rows.astype(np.float64), pred=np.stack([p[:3].reshape(12) for p in pred]),
                        err=np.asarray(err, np.float64), overall=np.asarray([ave_t, ave_r]),
                        dist=np.asarray(ev.trajectoryDistances(poses_gt)), err_short=np.asarray(err_short, np.float64))
    print("wrote", out, "segments:", len(err), "short:", len(err_short), "ave_t_err %.6f ave_r_err %.8f" % (ave_t, ave_r))


if __name__ == "__main__":
    main()
