"""CPU: trajectory I/O and ATE / RPE of the product package against the golden values produced by the
reference's own core/metrics/trajectory_metrics.py (tests/golden/metrics.npz) and against the oracle."""
import numpy as np
import torch

from conftest import load_golden
from oracle import metrics as ometrics
from oracle import se3


def test_ate_rpe_match_reference_golden(rpe):
    from rpe_amd import trajectory as tj
    g = load_golden('metrics.npz')
    G, P = tj.pose_matrices(g['gt'].numpy()), tj.pose_matrices(g['pred'].numpy())
    assert np.allclose(G, se3.se3_matrix(g['gt']).numpy(), atol=1e-12)
    ate, terr = tj.absolute_trajectory_error(G, P)
    assert abs(ate - float(g['ate'])) < 1e-10 and np.allclose(terr, g['trans_err'].numpy(), atol=1e-10)
    ate_na, _ = tj.absolute_trajectory_error(G, P, prealign=False)
    assert abs(ate_na - float(g['ate_noalign'])) < 1e-10
    t, r = tj.relative_pose_error(G, P)
    assert np.allclose(t, g['rpe_trans'].numpy(), atol=1e-10) and np.allclose(r, g['rpe_rot'].numpy(), atol=1e-10)


def test_freiburg_round_trip_and_offset_convention(rpe, tmp_path):
    from rpe_amd import trajectory as tj
    g = load_golden('metrics.npz')
    gt, pred = g['gt'].numpy(), g['pred'].numpy()
    m = len(gt)
    for name, poses in (('gt', gt), ('pred', pred)):
        d = tmp_path / name
        d.mkdir()
        tj.save_trajectory([{'camera-pose': torch.from_numpy(p), 'timestamp': i} for i, p in enumerate(poses)], str(d))
    back, stamps = tj.read_freiburg(str(tmp_path / 'gt' / 'trajectory.freiburg'), ret_stamps=True)
    assert np.allclose(back, gt, rtol=1e-12, atol=1e-9) and stamps.tolist() == list(range(m))   # mm -> m -> mm
    ob = ometrics.read_freiburg(str(tmp_path / 'gt' / 'trajectory.freiburg'))[1]
    assert np.allclose(ob, back)
    # offset = 0: stamps 1..m-2 are compared (0 < k < max)
    ate, rt, rr, terr, _, _ = tj.evaluate(str(tmp_path / 'gt' / 'trajectory.freiburg'), str(tmp_path / 'pred' / 'trajectory.freiburg'))
    Gm, Pm = tj.pose_matrices(gt[1:m - 1]), tj.pose_matrices(pred[1:m - 1])
    assert abs(ate - tj.absolute_trajectory_error(Gm, Pm)[0]) < 1e-9 and len(terr) == m - 2
    # offset = -4 (infer_trajectory.py:106): prediction k is matched with ground truth k-4
    ate4, *_ = tj.evaluate({i: p for i, p in enumerate(gt)}, {i + 4: p for i, p in enumerate(pred)}, offset=-4)
    assert abs(ate4 - ate) < 1e-9


def test_failed_frames_can_be_ignored(rpe):
    from rpe_amd import trajectory as tj
    g = load_golden('metrics.npz')
    G, P = tj.pose_matrices(g['gt'].numpy()), tj.pose_matrices(g['pred'].numpy())
    P2 = P.copy()
    P2[10] = P2[9]                                            # a skipped frame repeats the previous pose
    a_all, _ = tj.absolute_trajectory_error(G, P2)
    a_ign, terr = tj.absolute_trajectory_error(G, P2, ignore_failed_pos=True)
    assert len(terr) == len(G) - 1 and a_ign != a_all
    t_ign, _ = tj.relative_pose_error(G, P2, ignore_failed_pos=True)
    assert len(t_ign) == len(G) - 2
