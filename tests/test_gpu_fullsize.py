"""GPU: the BASELINE.json configurations at their REAL sizes against the CPU oracle (same seeded weights, same
seeded stereo pairs): config 2 geometry (640x512, PoseNet.infer end to end), config 3 (batch-32 RAFT, 640x512),
config 5 geometry (1280x1024: geometry pass, pose solve, one full infer) and config 4's collective under RCCL
(process group of one rank on the GPU box; the two-rank logic is covered by tests/test_sharding_gloo.py).

Tolerances (f32 network, different summation order only): flow <= 1e-3 px after 12 GRU iterations; masks equal except
at <= 50 pixels per batch whose decision value (stereo depth == 1, warp target == k + 0.5) the two float32 RAFT
implementations put on different sides (measured in round 3: 0 pixels at 640x512 n=2 and at 1280x1024); end-to-end pose
<= 5e-5 (north-star bar 1e-4) -- and, per row, <= 1e-6 x max(1, |pose|) once the pixels with a differing discrete
decision are excluded on both sides.  Round 2 attributed the 1.1e-5..1.7e-5 it measured at 640x512 to mask flips without
testing that; the measurement says otherwise: no pixel flips, and the whole difference sits in row 1 of seed 21, whose
8-iteration L-BFGS solve diverges on the random-init flow to a pose with |t| ~ 4 (the reference's gate rejects it) --
the difference there moves between 2.9e-6 and 7.2e-6 from build to build (a one-ulp change in an encoder epilogue is enough),
i.e. it is the conditioning of a diverging solve, not a discrete decision; row 0 agrees to 1e-10.  Bars per row: 1e-6 where the
pose stays in the unit ball (every solve the reference's gate could accept), 5e-5 for diverged rows.
"""
import os
import socket

import pytest
import torch

from oracle import pose_head as oph
from oracle import pose_net as opn

pytestmark = pytest.mark.gpu


def _models(h, w, iters=12, lbgfs_iters=8):
    from rpe_amd import pose_net, synth
    cfg = synth.model_config(h, w, iters=iters, lbgfs_iters=lbgfs_iters)
    model = synth.init_synthetic_weights(pose_net.PoseNet(cfg)).eval().cuda()
    om = opn.PoseNet(cfg)
    om.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    om.eval()
    return model, om, synth


@pytest.fixture(scope='module')
def models_640(rpe):
    return _models(512, 640)


def _check_infer(model, om, synth, h, w, n, seed):
    a = synth.infer_args(synth.stereo_frames(seed, n, h, w))
    g = model.stages(**{k: v.cuda() for k, v in a.items()})
    o = om.stages(**{k: v.clone() for k, v in a.items()})
    for k, tol in (('time_flow', 1e-3), ('stereo_flow2', 1e-3), ('pcl1', 1e-5), ('w2d', 1e-4), ('w3d', 1e-4)):
        d = float((g[k].cpu() - o[k]).abs().max())
        print(f'{w}x{h} {k}: {d:.2e}')
        assert d < tol, k
    f2, fw = int((g['mask2'].cpu() != o['mask2']).sum()), int((g['mask2w'].cpu() != o['mask2w']).sum())
    print(f'{w}x{h} n={n}: mask2 differs at {f2}, mask2w at {fw} of {o["mask2"].numel()} pixels')
    assert f2 <= 50 and fw <= 50
    m2 = a['mask2'].clone().cuda()
    pose = model.infer(**{k: (m2 if k == 'mask2' else v.cuda()) for k, v in a.items()})
    opose = om.infer(**{k: v.clone() for k, v in a.items()})
    d = float((pose.data.cpu().reshape(-1) - opose.reshape(-1)).abs().max())
    print(f'{w}x{h} n={n}: end-to-end pose diff {d:.2e}')
    # On IDENTICAL solver inputs the HIP solve matches the oracle to 1e-8 (test_gpu_pose / test_gpu_pipeline).  End to end
    # the flows differ by ~3e-5 px (summation order inside 40+ convolutions).  Bar: half the north-star tolerance
    # (1e-4 rad / 1e-4 translation-norm); what is measured is in the module docstring.
    assert d < 5e-5
    # Should a discrete decision ever flip (other hardware, other library versions): take every pixel at which a DISCRETE decision differs between the two runs -- the warped
    # mask (pose_net.py:107-108) or the 2-D term's in-image test on pix + flow (pose_head.py:24) -- out of mask1 (which
    # gates both residual terms) on BOTH sides and solve again: the poses then agree to 1e-6.
    def in_image(flow):
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64) + 0.5, torch.arange(w, dtype=torch.float64) + 0.5, indexing='ij')
        u, v = xs + flow[:, 0].double(), ys + flow[:, 1].double()
        return ((u > 0) & (v > 0) & (u < w) & (v < h))[:, None]
    differs = (g['mask2w'].cpu() != o['mask2w']) | (in_image(g['time_flow'].cpu()) != in_image(o['time_flow']))
    m1 = a['mask1'] & ~differs
    lw = model.loss_weight.detach()[None, :].repeat(n, 1)
    vec7, _ = model.pose_head(g['time_flow'], g['pcl1'], g['pcl2w'], g['w2d'], g['w3d'], m1.cuda(), g['mask2w'], g['intrinsics'], lw)
    To, _ = oph.lbfgs_solve(o['time_flow'], o['pcl1'], o['pcl2w'], o['w2d'], o['w3d'], m1, o['mask2w'], a['intrinsics'], lw.cpu(),
                            iters=model.pose_head.problem.lbgfs_iters, coupled=False)
    ref = oph.declarative_forward(To)[0][:, 0]
    ds = (vec7[:, 0].cpu() - ref).abs().amax(dim=1)
    de = (pose.data.cpu().reshape(n, 7) - opose.reshape(n, 7)).abs().amax(dim=1)
    scale = ref.abs().amax(dim=1).clamp_min(1.0)
    print(f'{w}x{h} n={n}: {int(differs.sum())} pixels with a differing discrete decision; per row: end-to-end diff {de.tolist()}, '
          f'without those pixels {ds.tolist()}, max |pose component| {scale.tolist()}')
    bar = torch.where(scale <= 1.0, torch.tensor(1e-6, dtype=ds.dtype), torch.tensor(5e-5, dtype=ds.dtype))
    assert bool((ds < bar).all())                             # (float32 output: 1 ulp at |t| ~ 4 is 4.8e-7)
    return a, o


def test_infer_640x512_matches_oracle(models_640):
    """BASELINE config 2 geometry: PoseNet.infer on n = 2 frame pairs of 640x512 (RAFT batch 4), stage by stage and end to end."""
    model, om, synth = models_640
    _check_infer(model, om, synth, 512, 640, 2, seed=21)


def test_raft_batch32_640x512_rows_match_oracle(models_640):
    """BASELINE config 3: one batch-32 RAFT pass (full correlation pyramid + 12 GRU iterations) at 640x512; rows 0, 13
    and 31 against the oracle run on those pairs alone (both encoders normalise per sample, so rows are independent)."""
    model, om, synth = models_640
    fr = synth.stereo_frames(31, 16, 512, 640)
    i1 = torch.cat((fr['image1l'], fr['image2l']))
    i2 = torch.cat((fr['image2l'], fr['image2r']))
    flows, hid, ctx = model.flow(i1.cuda(), i2.cuda())
    assert flows[-1].shape == (32, 2, 512, 640)
    for row in (0, 13, 31):
        with torch.no_grad():
            oflows, ohid, octx = om.flow(i1[row:row + 1], i2[row:row + 1])
        d = float((flows[-1][row:row + 1].cpu() - oflows[-1]).abs().max())
        print(f'batch-32 row {row}: flow diff {d:.2e} px')
        assert d < 1e-3
        assert float((hid[row:row + 1].cpu() - ohid).abs().max()) < 5e-3


def test_1280x1024_geometry_solve_and_infer(rpe):
    """BASELINE config 5 geometry in f32/f64: fused geometry pass and pose solve on the oracle's own inputs (tight), and one
    full PoseNet.infer (RAFT with 20 480 queries per pair) vs the oracle."""
    from rpe_amd import ops
    h, w = 1024, 1280
    model, om, synth = _models(h, w)
    a, o = _check_infer(model, om, synth, h, w, 1, seed=41)
    gg = ops.depth_backproject_warp(o['stereo_flow2'].cuda(), o['time_flow'].cuda(), a['baseline'].cuda(), a['intrinsics'].cuda(),
                                    a['depth1'].cuda(), a['image1l'].cuda(), a['image2l'].cuda(), a['stereo_flow1'].cuda(),
                                    a['mask2'].cuda())
    assert torch.equal(gg['mask2w'].cpu(), o['mask2w']) and torch.equal(gg['mask2'].cpu(), o['mask2'])
    assert torch.equal(gg['depth2'].cpu(), o['depth2'])                        # IEEE division, bit for bit
    assert float((gg['pcl2w'].cpu() - o['pcl2w']).abs().max()) < 1e-5
    assert float((gg['pcl1'].cpu() - o['pcl1']).abs().max()) < 1e-5
    lw = torch.ones(1, 2)
    args = (o['time_flow'], o['pcl1'], o['pcl2w'], o['w2d'], o['w3d'], a['mask1'], o['mask2w'], a['intrinsics'], lw)
    T, vec7, log6, info = ops.pose_solve(*[x.cuda() for x in args], iters=8)
    To, _ = oph.lbfgs_solve(*args, iters=8)
    assert float((T.cpu() - To).abs().max()) < 1e-8
    Tg, _, _, _ = ops.pose_solve(*[x.cuda() for x in args], iters=8, mode=ops.SOLVER_GN)
    Tgo, _ = oph.gn_solve(*args, iters=8)
    assert bool(torch.isfinite(Tg).all()) and float((Tg.cpu() - Tgo).abs().max()) < 1e-8


def test_rccl_all_gather_of_relative_poses(rpe):
    """BASELINE config 4's only collective, under RCCL on the GPU: a process group of ONE rank (this box has one GPU) must
    still execute the padded all-gather (sharding.gather_relative_poses has no world==1 shortcut) and give the serial result."""
    import torch.distributed as dist
    from rpe_amd import ops, sharding
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        assert float(ones) == 1.0
        calls = []
        real = dist.all_gather
        dist.all_gather = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            g = torch.Generator().manual_seed(3)
            xi = (torch.randn(9, 6, generator=g) * 0.02).cuda()
            rel_all = ops.se3_exp(xi)

            def run_block(s, e):
                return sharding.failure_gate(rel_all[s:e], ops.se3_log(rel_all[s:e]))
            poses, rel, ok = sharding.track_sharded(10, run_block, lambda r, sc: ops.se3_chain(r, scale=sc), 0, 1, scale=250.0)
        finally:
            dist.all_gather = real
        assert calls == [1]
        assert poses.shape == (10, 7) and bool(ok.all())
        assert torch.equal(rel, rel_all)                                       # f32 through the collective, unchanged
        assert float((poses[1:] - ops.se3_chain(rel_all, scale=250.0)).abs().max()) == 0.0
    finally:
        dist.destroy_process_group()


def test_config5_fp16_features_1280x1024(rpe):
    """BASELINE config 5: 1280x1024 stereo, fp16 features + f32 (f64) solve.  The HIP path (feature maps rounded to fp16, 16-bit
    MFMA correlation with f32 accumulation, everything else as the f32 path) against the oracle with the same rounding."""
    from rpe_amd import pose_net, synth
    h, w = 1024, 1280
    cfg = synth.model_config(h, w, iters=12, lbgfs_iters=8, mixed_precision=True)
    model = synth.init_synthetic_weights(pose_net.PoseNet(cfg)).eval().cuda()
    assert model.flow.mixed_precision
    om = opn.PoseNet(cfg)
    om.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    om.eval()
    assert om.flow.mixed_precision
    _check_infer(model, om, synth, h, w, 1, seed=43)
    # the rounding is visible: the f32 model gives a (slightly) different flow on the same frame
    cfg32 = synth.model_config(h, w, iters=12, lbgfs_iters=8)
    m32 = pose_net.PoseNet(cfg32)
    m32.load_state_dict(model.state_dict())
    m32.eval().cuda()
    a = synth.infer_args(synth.stereo_frames(43, 1, h, w))
    f16 = model.stages(**{k: v.cuda() for k, v in a.items()})['time_flow']
    f32 = m32.stages(**{k: v.cuda() for k, v in a.items()})['time_flow']
    d = float((f16 - f32).abs().max())
    print(f'fp16-feature vs f32 flow: {d:.2e} px')
    assert d > 1e-6
