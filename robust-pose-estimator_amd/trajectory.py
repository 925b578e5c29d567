"""Trajectory bookkeeping around the tracker: the per-frame loop of
scripts/infer_trajectory.py:71-97, Freiburg I/O of core/utils/trajectory.py:17-23,38-61 and the ATE / RPE
evaluation of core/metrics/trajectory_metrics.py:7-105 + evaluation/evaluate_ate_freiburg.py:6-31 (incl. its
time-stamp ``offset``; infer_trajectory.py:106 uses -4).  Host-side numpy: this is bookkeeping over a few
hundred 7-vectors, not part of the GPU hot path."""
import os

import numpy as np
import torch


def pose_matrices(vec7):
    """(m,7) [t, q(xyzw)] -> (m,4,4) homogeneous matrices (numpy, float64)."""
    v = np.asarray(vec7, dtype=np.float64).reshape(-1, 7)
    x, y, z, w = v[:, 3], v[:, 4], v[:, 5], v[:, 6]
    M = np.zeros((v.shape[0], 4, 4))
    M[:, 0, 0] = 1 - 2 * (y * y + z * z); M[:, 0, 1] = 2 * (x * y - z * w); M[:, 0, 2] = 2 * (x * z + y * w)
    M[:, 1, 0] = 2 * (x * y + z * w); M[:, 1, 1] = 1 - 2 * (x * x + z * z); M[:, 1, 2] = 2 * (y * z - x * w)
    M[:, 2, 0] = 2 * (x * z - y * w); M[:, 2, 1] = 2 * (y * z + x * w); M[:, 2, 2] = 1 - 2 * (x * x + y * y)
    M[:, :3, 3] = v[:, :3]
    M[:, 3, 3] = 1.0
    return M


def track_sequence(estimator, frames, start_stamp=0, chunk=1):
    """The loop of infer_trajectory.py:70-91.  ``frames`` yields (limg, rimg, mask, stamp) already on the device.
    Returns [{'camera-pose': (7,) tensor (mm), 'timestamp': stamp}], starting with the initial pose.
    ``chunk`` > 1: frames are handed to ``estimator.forward_chunk`` in groups of ``chunk`` (one RAFT pass per group; the same
    trajectory bit for bit, about twice the frames per second at 16); the sequence's first frame always goes through ``forward``."""
    import torch
    traj = [{'camera-pose': estimator.last_pose.vec().reshape(7).detach().cpu(), 'timestamp': start_stamp}]
    pending = []

    def flush():
        if not pending:
            return
        if len(pending) == 1:
            limg, rimg, mask, stamp = pending[0]
            poses = estimator(limg, rimg, mask)[0].vec().reshape(1, 7)
        else:
            poses = estimator.forward_chunk(torch.cat([p[0] for p in pending]), torch.cat([p[1] for p in pending]),
                                            torch.cat([p[2] for p in pending]))[0]
        for row, (_, _, _, stamp) in zip(poses.detach().cpu(), pending):
            traj.append({'camera-pose': row.reshape(7), 'timestamp': stamp})
        pending.clear()
    for item in frames:
        pending.append(item)
        if chunk <= 1 or estimator.frame is None or len(pending) >= chunk:
            flush()
    flush()
    return traj


def save_trajectory(trajectory, path):
    """trajectory.py:17-23 -- ``stamp tx ty tz qx qy qz qw`` per line, translation mm -> m."""
    fn = os.path.join(path, 'trajectory.freiburg')
    with open(fn, 'w') as f:
        for tr in trajectory:
            v = np.asarray(tr['camera-pose'], dtype=np.float64).reshape(7)
            f.write(f"{tr['timestamp']} {v[0] / 1000.0} {v[1] / 1000.0} {v[2] / 1000.0} {v[3]} {v[4]} {v[5]} {v[6]}\n")
    return fn


def read_freiburg(path, ret_stamps=False, no_stamp=False):
    """trajectory.py:38-61 -- poses (m,7) with translation converted m -> mm (and integer time stamps)."""
    with open(path) as f:
        lines = f.read().replace(',', ' ').replace('\t', ' ').split('\n')
    rows = [[v.strip() for v in ln.split(' ') if v.strip() != ''] for ln in lines if len(ln) > 0 and ln[0] != '#']
    rows = [r for r in rows if len(r) > 0]
    if no_stamp:
        arr = np.asarray([r[0:7] for r in rows], dtype=float)
        arr[:, :3] *= 1000.0
        return arr
    stamps = [r[0] for r in rows]
    try:
        stamps = np.asarray([int(s.split('.')[0] + s.split('.')[1]) for s in stamps]) * 100
    except IndexError:
        stamps = np.asarray([int(s) for s in stamps])
    arr = np.asarray([r[1:8] for r in rows], dtype=float)
    arr[:, :3] *= 1000.0
    return (arr, stamps) if ret_stamps else arr


def _align(model, data):
    """Horn closed-form alignment (trajectory_metrics.py:7-35); model, data are 3xn."""
    mz = model - model.mean(1, keepdims=True)
    dz = data - data.mean(1, keepdims=True)
    Wm = np.zeros((3, 3))
    for c in range(model.shape[1]):
        Wm += np.outer(mz[:, c], dz[:, c])
    U, _, Vh = np.linalg.svd(Wm.T)
    S = np.identity(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1
    rot = U @ S @ Vh
    T = np.eye(4)
    T[:3, :3] = rot
    T[:3, 3] = (data.mean(1, keepdims=True) - rot @ model.mean(1, keepdims=True))[:, 0]
    return T


def absolute_trajectory_error(gt_poses, predicted_poses, prealign=True, ignore_failed_pos=False):
    """ATE-RMSE (trajectory_metrics.py:38-73) on (m,4,4) arrays; returns (rmse, per-pose translation error)."""
    gt, pred = np.asarray(gt_poses, dtype=np.float64), np.asarray(predicted_poses, dtype=np.float64)
    assert len(gt) == len(pred)
    valid = np.ones(len(pred), dtype=bool)
    if ignore_failed_pos:                     # identical consecutive predictions mark failed estimations
        for i in range(len(pred) - 1):
            valid[i + 1] = (pred[i] - pred[i + 1]).sum() != 0
    if prealign:
        pred = _align(pred[valid, :3, 3].T, gt[valid, :3, 3].T)[None] @ pred
    terr = np.sum((gt[valid, :3, 3] - pred[valid, :3, 3]) ** 2, axis=1)
    return float(np.sqrt(np.mean(terr))), np.sqrt(terr)


def relative_pose_error(gt_poses, predicted_poses, delta=1, ignore_failed_pos=False):
    """RPE (trajectory_metrics.py:76-105): per-pair translation norm and rotation angle of gt_rel^-1 pred_rel."""
    gt, pred = np.asarray(gt_poses, dtype=np.float64), np.asarray(predicted_poses, dtype=np.float64)
    te, re = [], []
    for i in range(len(gt) - delta):
        if ((pred[i] - pred[i + 1]).sum() != 0) or (not ignore_failed_pos):
            e = np.linalg.inv(np.linalg.inv(gt[i]) @ gt[i + delta]) @ (np.linalg.inv(pred[i]) @ pred[i + delta])
            te.append(np.sqrt(np.sum(e[:3, 3] ** 2)))
            re.append(np.arccos(max(min(0.5 * (np.trace(e[:3, :3]) - 1), 1.0), -1.0)))
    return np.asarray(te), np.asarray(re)


def evaluate(gt, pred, delta=1, offset=0, ignore_failed_pos=False):
    """evaluate_ate_freiburg.py:6-31.  gt / pred: freiburg file paths or {stamp: (7,) pose} dicts.
    A prediction with stamp k is compared with the ground truth at k + offset (kept if 0 < k+offset < max stamp)."""
    def load(x):
        if isinstance(x, dict):
            return x
        poses, stamps = read_freiburg(x, ret_stamps=True)
        return {int(k): p for k, p in zip(stamps, poses)}
    gt_d, pr_d = load(gt), load(pred)
    gmax = max(gt_d.keys())
    P, G = [], []
    for k in sorted(pr_d.keys()):
        if (k + offset > 0) and (k + offset < gmax):
            P.append(pr_d[k])
            G.append(gt_d[k + offset])
    Pm, Gm = pose_matrices(np.stack(P)), pose_matrices(np.stack(G))
    ate, terr = absolute_trajectory_error(Gm, Pm, ignore_failed_pos=ignore_failed_pos)
    rt, rr = relative_pose_error(Gm, Pm, delta=delta, ignore_failed_pos=ignore_failed_pos)
    return ate, float(np.mean(rt)), float(np.mean(rr)), terr, rt, rr
