"""GPU: the HIP path against goldens produced by the reference's OWN core/unet/unet.py, core/pose/pose_net.py and
core/pose/pose_estimator.py (oracle/gen_golden.py::gen_modules; tests/test_oracle_modules.py pins the oracle to the same
files at 0.0).  Seeded inputs and weights are regenerated here; the fixtures' input moments guard the generators.

Bars: discrete results bit-exact up to pixels whose decision value the two float32 RAFT implementations put on different
sides (counted, <= 50); flows 1e-3 px after 12 GRU iterations; weight maps 1e-4; pose 1e-5 (north star 1e-4)."""
import warnings

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import pose_net as opn
from oracle import synth as osynth

pytestmark = pytest.mark.gpu
H, W = osynth.MODULE_HW


def _sub(t):
    return t[..., ::4, ::4]


def _unpack(bits, shape):
    n = int(np.prod(shape))
    return torch.from_numpy(np.unpackbits(bits.numpy())[:n].astype(bool).reshape(shape))


@pytest.mark.parametrize('cin', [264, 272])
def test_tiny_unet_matches_reference(rpe, cin):
    from rpe_amd import ops, unet
    g = load_golden('unet.npz')
    x, sd = osynth.unet_case(cin)
    net = unet.TinyUNet(cin, (H, W))
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    scale = float(g[f'u{cin}_eval_sub'].abs().max())
    y = net(x[:1].cuda())                                          # module route: library convolutions + HIP epilogues
    assert float((_sub(y).cpu() - g[f'u{cin}_eval_sub']).abs().max()) <= 2e-5 * scale
    # the fused kernel chain PoseNet runs (rpe_unet_heads): this head's channels in the places PoseNet feeds them
    xc = x[:1].cuda()
    if cin == 264:
        inp1, inp2, hid, ctx = xc[:, :8], torch.zeros_like(xc[:, :8]), xc[:, 8:136], xc[:, 136:]
    else:
        inp1, inp2, hid, ctx = xc[:, :8], xc[:, 8:16], xc[:, 16:144], xc[:, 144:]
    other = unet.TinyUNet(536 - cin, (H, W)).cuda().eval()
    blobs = (unet.pack_params(net), unet.pack_params(other)) if cin == 264 else (unet.pack_params(other), unet.pack_params(net))
    w = ops.unet_heads(inp1.contiguous(), inp2.contiguous(), hid.contiguous(), ctx.contiguous(), *blobs, (H, W))[0 if cin == 264 else 1]
    assert float((_sub(w).cpu() - torch.sigmoid(g[f'u{cin}_eval_sub'])).abs().max()) <= 1e-5
    # train mode = batch statistics (what the reference trains): differentiable route, with and without gradients
    net.train()
    for grad in (True, False):
        net.load_state_dict(sd, strict=True)
        with torch.set_grad_enabled(grad):
            yt = net(x.cuda())
        assert yt.requires_grad == grad
        ts = float(g[f'u{cin}_train_sub'].abs().max())
        assert float((_sub(yt).detach().cpu() - g[f'u{cin}_train_sub']).abs().max()) <= 1e-4 * ts


@pytest.fixture(scope='module')
def case(rpe):
    from rpe_amd import pose_net, synth
    cfg, sd, a = osynth.posenet_case(synth, opn)
    model = pose_net.PoseNet(cfg)
    model.load_state_dict(sd, strict=True)
    return model.eval().cuda(), a, load_golden('posenet.npz'), synth


def test_posenet_stages_match_reference(case):
    model, a, g, _ = case
    shape = a['mask2'].shape
    s = model.stages(**{k: v.clone().cuda() for k, v in a.items()})
    for k, bar in (('time_flow', 1e-3), ('stereo_flow2', 1e-3), ('pcl1', 1e-6), ('w2d', 1e-4), ('w3d', 1e-4)):
        d = float((_sub(s[k]).cpu() - g[k + '_sub']).abs().max())
        print(f'{k}: {d:.2e}')
        assert d <= bar, k
    # discrete outputs
    m2 = _unpack(g['mask2_after'], shape)
    m2w = _unpack(g['mask2w'], shape)
    f2, fw = int((s['mask2'].cpu() != m2).sum()), int((s['mask2w'].cpu().bool() != m2w).sum())
    print(f'mask2: {f2} / mask2w: {fw} of {m2.numel()} pixels differ')
    assert f2 <= 50 and fw <= 50
    # continuous outputs away from the flipped pixels
    same = (s['mask2'].cpu() == m2) & (s['mask2w'].cpu().bool() == m2w)
    for k, bar in (('depth2', 1e-5), ('pcl2w', 2e-5)):
        d = ((_sub(s[k]).cpu() - g[k + '_sub']).abs() * _sub(same)).max()
        print(f'{k}: {float(d):.2e}')
        assert float(d) <= bar, k


def test_posenet_infer_matches_reference(case):
    model, a, g, _ = case
    m2 = a['mask2'].clone().cuda()
    pose, depth1, depth2, maps, tf, sf2 = model.infer(**{k: (m2 if k == 'mask2' else v.clone().cuda()) for k, v in a.items()},
                                                     ret_details=True)
    d = float((pose.data.cpu().reshape(1, 7) - g['pose']).abs().max())
    print(f'pose vs reference: {d:.2e}')
    assert d <= 1e-5
    assert int((m2.cpu() != _unpack(g['mask2_after'], m2.shape)).sum()) <= 50         # the caller's mask was mutated (pose_net.py:77)
    head = model.pose_head.problem
    try:
        head.lbgfs_iters = 20                                                          # configuration/infer_f2f.yaml:11
        p20 = model.infer(**{k: v.clone().cuda() for k, v in a.items()})
        assert float((p20.data.cpu().reshape(1, 7) - g['pose_k20']).abs().max()) <= 1e-5
        head.lbgfs_iters, model.use_weights = 8, False                                 # infer_f2f_nw.yaml:9
        pnw = model.infer(**{k: v.clone().cuda() for k, v in a.items()})
        assert float((pnw.data.cpu().reshape(1, 7) - g['pose_nw']).abs().max()) <= 1e-5
    finally:
        head.lbgfs_iters, model.use_weights = 8, True


def test_flow2depth_matches_reference(case):
    model, a, g, _ = case
    d, f, v = model.flow2depth(a['image2l'].cuda(), a['image2r'].cuda(), a['baseline'].cuda())
    vref = _unpack(g['f2d_valid'], v.shape)
    flips = int((v.cpu() != vref).sum())
    assert flips <= 50
    assert float(((_sub(d).cpu() - g['f2d_depth_sub']).abs() * _sub(v.cpu() == vref)).max()) <= 1e-5
    # upsample=False (pose_net.py:131-132) against the oracle restatement of the same three lines
    cfg = dict(model.config)
    om = opn.PoseNet(cfg)
    om.load_state_dict({k: t.cpu() for k, t in model.state_dict().items()})
    om.eval()
    dl, fl, vl = model.flow2depth(a['image2l'].cuda(), a['image2r'].cuda(), a['baseline'].cuda(), upsample=False)
    do, fo, vo = om.flow2depth(a['image2l'], a['image2r'], a['baseline'], upsample=False)
    assert dl.shape == do.shape == (1, 1, H // 8, W // 8) and fl.shape == (1, 2, H // 8, W // 8)
    assert float((fl.cpu() - fo).abs().max()) <= 1e-3
    same = vl.cpu() == vo
    assert int((~same).sum()) <= 5 and float(((dl.cpu() - do).abs() * same).max()) <= 1e-5


def test_tracker_matches_reference(case):
    model, _, _, synth = case
    from rpe_amd import pose_estimator
    g = load_golden('tracker.npz')
    frames, K, bf = osynth.tracker_case(synth)
    cfg = dict(frame2frame=True, depth_clipping=[1, 250], lbgfs_iters=8, conf_weighing=True)
    for reuse in (False, True):
        est = pose_estimator.PoseEstimator(dict(cfg, reuse_features=reuse), K, bf, model, (W, H)).cuda()
        for i, (l, r, m) in enumerate(frames):
            P, _, _, _ = est(l.cuda(), r.cuda(), m.clone().cuda())
            d = float((P.data.cpu().reshape(7) - g['abs_poses'][i]).abs().max())
            print(f'frame {i} (reuse {reuse}): abs pose diff {d:.2e} mm')
            assert d <= 2e-3                                  # millimetres after x250
            assert int((est.frame.mask.cpu() != _unpack(g['masks'][i], m.shape)).sum()) <= 50
        assert float((_sub(est.frame.depth).cpu() - g['depth_last_sub']).abs().median()) <= 1e-3


def test_chunked_tracker_matches_reference(case):
    """The same reference run (core/pose/pose_estimator.py on three frames, tests/golden/tracker.npz) through forward_chunk: frame 0 by
    forward, frames 1 and 2 as ONE pass of PoseNet.infer_chunk -- the reference's own poses, masks and depth, not only the bits of the
    frame-at-a-time route."""
    model, _, _, synth = case
    from rpe_amd import pose_estimator
    g = load_golden('tracker.npz')
    frames, K, bf = osynth.tracker_case(synth)
    cfg = dict(frame2frame=True, depth_clipping=[1, 250], lbgfs_iters=8, conf_weighing=True)
    est = pose_estimator.PoseEstimator(cfg, K, bf, model, (W, H)).cuda()
    l0, r0, m0 = frames[0]
    P0, _, _, _ = est(l0.cuda(), r0.cuda(), m0.clone().cuda())
    assert float((P0.data.cpu().reshape(7) - g['abs_poses'][0]).abs().max()) <= 2e-3
    ls, rs = torch.cat([f[0] for f in frames[1:]]).cuda(), torch.cat([f[1] for f in frames[1:]]).cuda()
    ms = torch.cat([f[2] for f in frames[1:]]).clone().cuda()
    poses, _, _, _ = est.forward_chunk(ls, rs, ms)
    for i in range(1, len(frames)):
        d = float((poses[i - 1].cpu() - g['abs_poses'][i]).abs().max())
        print(f'frame {i} (chunked): abs pose diff {d:.2e} mm')
        assert d <= 2e-3
        assert int((ms[i - 1:i].cpu() != _unpack(g['masks'][i], frames[i][2].shape)).sum()) <= 50
    assert bool(est.successes.all())
    assert float((_sub(est.frame.depth).cpu() - g['depth_last_sub']).abs().median()) <= 1e-3


class _Scripted(torch.nn.Module):
    """Prescribed relative poses in place of PoseNet (as in the generator), with this package's infer() signature."""

    def __init__(self, rel):
        super().__init__()
        self.rel, self.i = rel, 0

    def flow2depth(self, l, r, baseline, ret_cache=False):
        return torch.ones_like(l[:, :1]), torch.zeros_like(l[:, :2]), torch.ones_like(l[:, :1], dtype=torch.bool), None

    def infer(self, img1, *args, **kw):
        from rpe_amd.se3 import SE3
        p = SE3(self.rel[self.i:self.i + 1].clone().to(img1.device))[0]
        self.i += 1
        one = torch.ones_like(img1[:, :1])
        return p, one, one, (one, one), torch.zeros_like(img1[:, :2]), torch.zeros_like(img1[:, :2]), None


def test_tracker_gate_scale_and_chain_match_reference(case):
    model, _, _, _ = case
    from rpe_amd import pose_estimator
    g = load_golden('tracker.npz')
    rel = osynth.gate_case()
    cfg = dict(frame2frame=True, depth_clipping=[1, 250], lbgfs_iters=8, conf_weighing=True)
    est = pose_estimator.PoseEstimator(cfg, torch.eye(3), 1000.0, model, (W, H)).cuda()
    est.model = _Scripted(rel)
    tiny = torch.zeros(1, 3, 8, 8, device='cuda')
    out, ok = [], []
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter('always')
        for _ in range(rel.shape[0] + 1):
            P, *_ = est(tiny, tiny, torch.ones(1, 1, 8, 8, dtype=torch.bool, device='cuda'))
            out.append(P.data.cpu().reshape(7))
            ok.append(est.success)
    out = torch.stack(out)
    assert float((out - g['gate_abs']).abs().max()) <= 1e-5 * float(g['gate_abs'].abs().max())
    assert ok == [True, True, True, True, False, True, False, True, False, True]
    assert len([w for w in wl if issubclass(w.category, RuntimeWarning)]) == int(g['gate_warnings']) == 3
