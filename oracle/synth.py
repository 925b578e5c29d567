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


# ----------------------------------------------------------------------------- module-level cases (tests/golden/{unet,posenet,tracker}.npz)
MODULE_HW = (352, 384)          # the smallest image the weight heads admit is 352x352 (1/8 grid 44x44, core/unet/unet.py)


def randomize_norms(model, seed):
    """Seeded non-trivial BatchNorm state (running mean / variance, scale, shift): freshly reset statistics (0, 1) would
    make the frozen-BN folding of the context encoder and the weight heads invisible to a parity test."""
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                c = m.num_features
                m.running_mean.copy_(torch.from_numpy(rng.normal(0, 0.2, c).astype(np.float32)))
                m.running_var.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)))
                m.weight.copy_(torch.from_numpy(rng.uniform(0.7, 1.3, c).astype(np.float32)))
                m.bias.copy_(torch.from_numpy(rng.normal(0, 0.1, c).astype(np.float32)))
    return model


def unet_case(cin, seed=71):
    """(x (2,cin,44,48) f32, state dict) of a TinyUNet(cin, MODULE_HW) with seeded weights and norm state."""
    from .unet import TinyUNet
    torch.manual_seed(seed + cin)
    net = TinyUNet(cin, MODULE_HW)
    for m in net.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            m.reset_parameters()
    randomize_norms(net, seed + cin)
    rng = np.random.default_rng(seed + cin + 1)
    x = torch.from_numpy(rng.normal(0, 1, size=(2, cin, MODULE_HW[0] // 8, MODULE_HW[1] // 8)).astype(np.float32))
    x[:, :8] *= 20.0                      # the eight geometry channels are not unit scale (disparities, 0..255 images)
    return x, {k: v.clone() for k, v in net.state_dict().items()}


def posenet_case(psynth, opn, seed=5):
    """(config, state dict, PoseNet.infer arguments) for one 352x384 frame pair.  ``psynth`` = the product package's
    synth module (the stereo renderer bench.py uses), ``opn`` = oracle.pose_net; weights = the seeded synthetic
    initialisation + randomize_norms."""
    H, W = MODULE_HW
    cfg = psynth.model_config(H, W, iters=12, lbgfs_iters=8)
    del cfg['solver'], cfg['mixed_precision']          # keys the reference's config does not have
    om = randomize_norms(psynth.init_synthetic_weights(opn.PoseNet(cfg)), seed + 100).eval()
    a = psynth.infer_args(psynth.stereo_frames(seed, 1, H, W))
    return cfg, {k: v.clone() for k, v in om.state_dict().items()}, a


def tracker_case(psynth, seed=9, n_frames=3):
    """[(left, right, mask)] * n_frames + K (3,3) + bf (pixels * mm): consecutive stereo frames for PoseEstimator.
    Frame i's left image is pair i's ``image1l``; right images come from the same renderer (pair i's ``image2r`` belongs to
    ``image2l``, so the frames are taken from independent pairs' second views -- the tracker does not care)."""
    H, W = MODULE_HW
    s = psynth.stereo_frames(seed, n_frames, H, W)
    frames = [(s['image2l'][i:i + 1], s['image2r'][i:i + 1], s['mask2'][i:i + 1]) for i in range(n_frames)]
    return frames, s['K'][0], float(s['baseline'][0]) * 250.0


def gate_case(seed=13):
    """(m,7) f32 relative poses for the failure gate (core/pose/pose_estimator.py:81-87): ordinary ones, log components
    just below / just above 0.1 in a translation and in a rotation entry, a NaN pose, and ordinary ones again."""
    rng = np.random.default_rng(seed)
    xi = np.concatenate((rng.normal(0, 0.01, size=(9, 3)), rng.normal(0, 0.02, size=(9, 3))), axis=1)
    xi[2, 1] = 0.0999
    xi[3, 1] = 0.1001
    xi[4, 4] = -0.0999
    xi[5, 4] = -0.1001
    rel = _se3.se3_exp(torch.from_numpy(xi)).float()
    rel[7] = float('nan')
    return rel
