"""Seeded synthetic stereo inputs and weights for tests and bench.py: there is no
network for datasets or checkpoints and the reference's trained weights are stripped blobs
(.MISSING_LARGE_BLOBS), so the benchmark runs random-init weights of the reference architecture on rendered
stereo pairs.  numpy ``default_rng`` seeds; everything is built on the CPU and moved by the caller."""
import numpy as np
import torch
import torch.nn.functional as F


def intrinsics(h, w):
    return torch.tensor([[1.1 * w, 0.0, w / 2.0], [0.0, 1.1 * w, h / 2.0], [0.0, 0.0, 1.0]], dtype=torch.float32)


def _smooth(rng, c, h, w, cells):
    coarse = torch.from_numpy(rng.uniform(0.0, 1.0, size=(1, c, cells, cells + 2)).astype(np.float32))
    return F.interpolate(coarse, size=(h, w), mode='bicubic', align_corners=True)[0].clamp(0, 1)


def _warp(img, dx, dy):
    n, _, h, w = img.shape
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    gx = 2 * (xs[None] + dx) / (w - 1) - 1
    gy = 2 * (ys[None] + dy) / (h - 1) - 1
    return F.grid_sample(img, torch.stack((gx, gy), dim=-1), align_corners=True, padding_mode='border')


def stereo_frames(seed, n, h, w, bf=7.2):
    """n independent frame pairs.  Returns dict of CPU tensors with the arguments of PoseNet.infer:
    image1l, image2l, image2r (n,3,h,w) 0..255; K (n,3,3); baseline (n,) normalised (bf/250);
    depth1 (n,1,h,w) in (0,1]; mask1, mask2 (n,1,h,w) bool; stereo_flow1 (n,2,h,w); xi_gt (n,6)."""
    rng = np.random.default_rng(seed)
    K = intrinsics(h, w)
    out = {k: [] for k in ('image1l', 'image2l', 'image2r', 'depth1', 'mask1', 'mask2', 'stereo_flow1', 'xi_gt')}
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5, indexing='ij')
    for _ in range(n):
        depth = 0.2 + 0.7 * _smooth(rng, 1, h, w, 5)[0]                                   # normalised units
        tex = 255.0 * (0.6 * _smooth(rng, 3, h, w, 24) + 0.4 * _smooth(rng, 3, h, w, 96))  # band-limited RGB
        xi = np.concatenate((rng.normal(0, 0.005, 3), rng.normal(0, 0.01, 3))).astype(np.float32)
        # small-motion flow of a rigid scene: X' = X + phi x X + tau
        X = torch.stack(((xs - K[0, 2]) / K[0, 0] * depth, (ys - K[1, 2]) / K[1, 1] * depth, depth))
        tau, phi = torch.from_numpy(xi[:3]), torch.from_numpy(xi[3:])
        Xp = X + torch.cross(phi[:, None, None].expand_as(X), X, dim=0) + tau[:, None, None]
        u = K[0, 0] * Xp[0] / Xp[2] + K[0, 2]
        v = K[1, 1] * Xp[1] / Xp[2] + K[1, 2]
        fx, fy = u - xs, v - ys
        disp = bf / depth
        img1 = tex[None]
        img2 = _warp(img1, -fx[None], -fy[None])
        img2r = _warp(img2, disp[None], torch.zeros_like(disp)[None])
        m1 = torch.ones(1, h, w, dtype=torch.bool)
        m2 = torch.ones(1, h, w, dtype=torch.bool)
        y0, x0 = int(rng.integers(0, h - h // 5)), int(rng.integers(0, w - w // 4))
        m1[:, y0:y0 + h // 5, x0:x0 + w // 4] = False                                     # 5 % rectangle cut out
        y0, x0 = int(rng.integers(0, h - h // 5)), int(rng.integers(0, w - w // 4))
        m2[:, y0:y0 + h // 5, x0:x0 + w // 4] = False
        out['image1l'].append(img1[0]); out['image2l'].append(img2[0]); out['image2r'].append(img2r[0])
        out['depth1'].append(depth[None]); out['mask1'].append(m1); out['mask2'].append(m2)
        out['stereo_flow1'].append(torch.stack((-disp, torch.zeros_like(disp))))
        out['xi_gt'].append(torch.from_numpy(xi))
    res = {k: torch.stack(v).contiguous() for k, v in out.items()}
    res['K'] = K[None].repeat(n, 1, 1)
    res['baseline'] = torch.full((n,), bf, dtype=torch.float32)
    return res


def ground_truth_flows(seed, n, h, w, bf=7.2, occluders=3):
    """What a TRAINED flow network converges to on a scene with depth edges (the lookup's realistic-coordinates roofline line in
    bench.py): a smooth background at normalised depth 0.45..0.9 with ``occluders`` foreground ellipses at depth 0.12..0.3, so
    the stereo disparities span 8..60 px with discontinuities, and the rigid temporal flow of a gate-sized camera motion
    (same distribution as stereo_frames).  Returns (time_flow, stereo_flow), each (n,2,h,w) f32 in pixels."""
    rng = np.random.default_rng(seed)
    K = intrinsics(h, w)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5, indexing='ij')
    tf, sf = [], []
    for _ in range(n):
        depth = 0.45 + 0.45 * _smooth(rng, 1, h, w, 5)[0]
        for _ in range(occluders):
            cx, cy = rng.uniform(0.15, 0.85) * w, rng.uniform(0.15, 0.85) * h
            ax, ay = rng.uniform(0.05, 0.2) * w, rng.uniform(0.05, 0.25) * h
            inside = ((xs - cx) / ax) ** 2 + ((ys - cy) / ay) ** 2 < 1.0
            depth = torch.where(inside, torch.full_like(depth, float(rng.uniform(0.12, 0.3))) + 0.02 * (xs - cx) / ax, depth)
        xi = np.concatenate((rng.normal(0, 0.005, 3), rng.normal(0, 0.01, 3))).astype(np.float32)
        X = torch.stack(((xs - K[0, 2]) / K[0, 0] * depth, (ys - K[1, 2]) / K[1, 1] * depth, depth))
        tau, phi = torch.from_numpy(xi[:3]), torch.from_numpy(xi[3:])
        Xp = X + torch.cross(phi[:, None, None].expand_as(X), X, dim=0) + tau[:, None, None]
        u = K[0, 0] * Xp[0] / Xp[2] + K[0, 2]
        v = K[1, 1] * Xp[1] / Xp[2] + K[1, 2]
        tf.append(torch.stack((u - xs, v - ys)))
        disp = bf / depth
        sf.append(torch.stack((-disp, torch.zeros_like(disp))))
    return torch.stack(tf).contiguous(), torch.stack(sf).contiguous()


def infer_args(s):
    return dict(image1l=s['image1l'], image2l=s['image2l'], intrinsics=s['K'], baseline=s['baseline'], depth1=s['depth1'],
                image2r=s['image2r'], mask1=s['mask1'], mask2=s['mask2'], stereo_flow1=s['stereo_flow1'])


def model_config(h, w, iters=12, lbgfs_iters=8, solver='lbfgs', use_weights=True, mixed_precision=False):
    """The ``model`` section of configuration/train.yaml:1-9 of the reference + image shape / solver settings.
    ``mixed_precision`` (upstream RAFT's flag): fp16 feature maps into the correlation (BASELINE config 5)."""
    return dict(small=False, dropout=0.0, iters=iters, pose_scale=1.0, lbgfs_iters=lbgfs_iters, use_weights=use_weights,
                image_shape=(h, w), solver=solver, mixed_precision=mixed_precision)


def init_synthetic_weights(model, seed=1234, flow_bias=-0.35):
    """Seeded re-initialisation of every parameter (module default initialisers, in module order) plus one
    deterministic adjustment: a negative x bias on the flow head.  Untrained RAFT drifts to positive x flow, which
    would make every stereo depth invalid (depth = bf / -flow.x, pose_net.py:73-75) and hand the solver an
    all-masked problem; the bias keeps disparities negative so all stages do real work on synthetic data."""
    torch.manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            m.reset_parameters()
        elif isinstance(m, torch.nn.BatchNorm2d):
            m.reset_parameters()
            m.reset_running_stats()
    with torch.no_grad():
        if hasattr(model, 'loss_weight'):
            model.loss_weight.fill_(1.0)
        raft = model.flow if hasattr(model, 'flow') else model
        raft.update_block.flow_head.conv2.bias.copy_(torch.tensor([flow_bias, 0.0]))
    return model
