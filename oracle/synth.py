"""Oracle-side synthetic inputs (test infrastructure).

Seeded with ``numpy.random.default_rng`` as SURVEY.md section 8(d) prescribes.  Two families:

  * ``solver_case``  -- inputs of the pose layer (flow, pcl1, pcl2, w2D, w3D, masks, K, loss_weight)
                        built from a known relative pose, in the spirit of the reference's own
                        tests/unit_test_pose_head.py:13-36 (depth -> back-projection -> induced flow).
  * ``stereo_scene`` -- a textured stereo pair sequence rendered from a smooth depth map and a known
                        relative pose (used by the PoseNet-level tests and by bench.py through
                        the product package's own copy of the same recipe).
"""
import numpy as np
import torch

from . import se3 as _se3
from .pose_head import img_coords


def smooth_field(rng, h, w, cells=6):
    """Low-pass random field in [0,1]: bilinear up-sampling of a coarse uniform grid."""
    coarse = torch.from_numpy(rng.uniform(0.0, 1.0, size=(1, 1, cells, cells + 1)).astype(np.float32))
    f = torch.nn.functional.interpolate(coarse, size=(h, w), mode='bilinear', align_corners=True)
    return f[0, 0]


def intrinsics(h, w):
    K = np.array([[1.1 * w, 0.0, w / 2.0], [0.0, 1.1 * w, h / 2.0], [0.0, 0.0, 1.0]], dtype=np.float32)
    return torch.from_numpy(K)


def solver_case(seed, n, h, w, sigma_t=0.005, sigma_r=0.01, noise=1e-3, unit_weights=False,
                full_masks=False, outliers=True):
    """Pose-layer inputs with known ground-truth pose ``xi_gt`` (n,6) f64."""
    rng = np.random.default_rng(seed)
    K = intrinsics(h, w)[None].repeat(n, 1, 1)
    pix = img_coords(h, w, torch.float64)
    depth = torch.stack([0.2 + 0.7 * smooth_field(rng, h, w) for _ in range(n)]).double().reshape(n, 1, -1)
    Kinv = torch.linalg.inv(K.double())
    pcl1 = (depth * (Kinv @ pix[None])).reshape(n, 3, h, w)
    xi = np.concatenate((rng.normal(0, sigma_t, size=(n, 3)), rng.normal(0, sigma_r, size=(n, 3))), axis=1)
    xi = torch.from_numpy(xi)
    T = _se3.se3_exp(xi).reshape(n, 1, 7)
    X = _se3.se3_act(T, pcl1.reshape(n, 3, -1).permute(0, 2, 1))
    ip = torch.einsum('nij,npj->npi', K.double(), X)
    uv = ip[..., :2] / ip[..., 2:3]
    flow = (uv.permute(0, 2, 1) - pix[None, :2]).reshape(n, 2, h, w)
    flow = flow + noise * 50 * torch.from_numpy(rng.normal(size=flow.shape))
    pcl2 = X.permute(0, 2, 1).reshape(n, 3, h, w) + noise * torch.from_numpy(rng.normal(size=(n, 3, h, w)))
    if unit_weights:
        w1 = torch.ones(n, 1, h, w)
        w2 = torch.ones(n, 1, h, w)
    else:
        w1 = torch.from_numpy(rng.uniform(0.05, 1.0, size=(n, 1, h, w)).astype(np.float32))
        w2 = torch.from_numpy(rng.uniform(0.05, 1.0, size=(n, 1, h, w)).astype(np.float32))
    m1 = torch.ones(n, 1, h, w, dtype=torch.bool)
    m2 = torch.ones(n, 1, h, w, dtype=torch.bool)
    if not full_masks:
        for i in range(n):
            y0, x0 = int(rng.integers(0, h // 2)), int(rng.integers(0, w // 2))
            m1[i, 0, y0:y0 + h // 5, x0:x0 + w // 4] = False
            y0, x0 = int(rng.integers(0, h // 2)), int(rng.integers(0, w // 2))
            m2[i, 0, y0:y0 + h // 4, x0:x0 + w // 5] = False
    flow = flow.float()
    pcl1 = pcl1.float()
    pcl2 = pcl2.float()
    if outliers is True:
        outliers = 'flow'
    if outliers:
        for i in range(n):
            # flows that leave the image on every side (pose_head.py:24 strict bounds)
            flow[i, 0, 0, :3] = -5.0
            flow[i, 1, 1, :3] = -7.5
            flow[i, 0, 2, -3:] = 6.0
            flow[i, 1, -1, 4:7] = 3.25
            flow[i, 0, 3, 0] = -0.5            # lands exactly on x == 0 -> invalid (strict >)
    if outliers == 'clamp':
        for i in range(n):
            # a point behind / on the camera plane: z < 1e-12 triggers the clamp (pinhole_transforms.py:95)
            pcl1[i, 2, 5, 5] = -0.25
            pcl1[i, 2, 6, 6] = 0.0
            pcl1[i, :, 7, 7] = 0.0
    lw = torch.from_numpy(rng.uniform(0.5, 1.5, size=(n, 2)).astype(np.float32))
    return dict(flow=flow, pcl1=pcl1, pcl2=pcl2, w1=w1, w2=w2, mask1=m1, mask2=m2, K=K, loss_weight=lw,
                xi_gt=xi)


def solver_args(c):
    return (c['flow'], c['pcl1'], c['pcl2'], c['w1'], c['w2'], c['mask1'], c['mask2'], c['K'], c['loss_weight'])
