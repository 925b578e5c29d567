"""Oracle (test infrastructure): ATE / RPE and Freiburg trajectory I/O, restating
core/metrics/trajectory_metrics.py:7-105 and core/utils/trajectory.py:17-23,38-61 of the reference
(numpy only; pinned by tests/golden/metrics.npz produced from the reference file itself)."""
import numpy as np


def align(model, data):
    """Horn closed-form alignment (trajectory_metrics.py:7-35); model, data are 3xn."""
    mz = model - model.mean(1, keepdims=True)
    dz = data - data.mean(1, keepdims=True)
    W = np.zeros((3, 3))
    for c in range(model.shape[1]):
        W += np.outer(mz[:, c], dz[:, c])
    U, d, Vh = np.linalg.svd(W.T)
    S = np.identity(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1
    rot = U @ S @ Vh
    trans = data.mean(1, keepdims=True) - rot @ model.mean(1, keepdims=True)
    T = np.eye(4)
    T[:3, :3] = rot
    T[:3, 3] = trans[:, 0]
    return T


def absolute_trajectory_error(gt, pred, prealign=True):
    gt, pred = np.asarray(gt), np.asarray(pred)
    if prealign:
        T = align(pred[:, :3, 3].T, gt[:, :3, 3].T)
        pred = T[None] @ pred
    terr = np.sum((gt[:, :3, 3] - pred[:, :3, 3]) ** 2, axis=1)
    return float(np.sqrt(np.mean(terr))), np.sqrt(terr)


def relative_pose_error(gt, pred, delta=1):
    te, re = [], []
    for i in range(len(gt) - delta):
        g = np.linalg.inv(gt[i]) @ gt[i + delta]
        p = np.linalg.inv(pred[i]) @ pred[i + delta]
        e = np.linalg.inv(g) @ p
        te.append(np.sqrt(np.sum(e[:3, 3] ** 2)))
        d = 0.5 * (np.trace(e[:3, :3]) - 1)
        re.append(np.arccos(max(min(d, 1.0), -1.0)))
    return np.asarray(te), np.asarray(re)


def save_freiburg(path, stamps, poses_mm):
    """trajectory.py:17-23 -- ``stamp tx ty tz qx qy qz qw`` with translation mm -> m."""
    with open(path, 'w') as f:
        for s, v in zip(stamps, np.asarray(poses_mm)):
            f.write(f"{s} {v[0] / 1000.0} {v[1] / 1000.0} {v[2] / 1000.0} {v[3]} {v[4]} {v[5]} {v[6]}\n")


def read_freiburg(path):
    """trajectory.py:38-61 (stamped variant) -- returns stamps (strings) and poses with translation in mm."""
    stamps, rows = [], []
    with open(path) as f:
        for line in f.read().replace(',', ' ').replace('\t', ' ').split('\n'):
            if len(line) == 0 or line[0] == '#':
                continue
            v = [x for x in line.split(' ') if x.strip() != '']
            stamps.append(v[0])
            rows.append([float(x) for x in v[1:8]])
    arr = np.asarray(rows)
    arr[:, :3] *= 1000.0
    return stamps, arr
