"""The reference's REAL entry points: ``PoseEstimator(slam_config, K, bf, '<checkpoint>.pth', (W, H))`` as
scripts/infer_trajectory.py:50-51 builds it (core/pose/pose_estimator.py:26-37: the checkpoint's own ``config['model']``
with three overrides, ``module.``-prefixed DataParallel keys) and ``PoseNet.init_from_raft('<raft-things>.pth')``
(core/pose/pose_net.py:137-147).  The trained blobs are stripped from the checkout, so the checkpoints are written here
with the file layout those loaders read: every other test hands over a live PoseNet, these go through the files.

CPU part: construction, overrides, key mapping, the parsed ``slam`` section of configuration/infer_f2f.yaml key for key.
GPU part: the path-built estimator reproduces the reference-generated 3-frame golden (tests/golden/tracker.npz)."""
import numpy as np
import pytest
import torch
import yaml

from conftest import load_golden
from oracle import pose_net as opn
from oracle import synth as osynth

H, W = osynth.MODULE_HW

# configuration/infer_f2f.yaml:1-11 of the reference, verbatim keys (dist_thr / debug / average_pts belong to the frame-to-model
# branch and are ignored on the f2f path, exactly as core/pose/pose_estimator.py never reads them there)
INFER_F2F_YAML = """
slam:
  frame2frame: True
  checkpoint:
  dist_thr: 0.05
  depth_clipping:
    - 1
    - 250
  debug: False
  conf_weighing: True
  average_pts: False
  lbgfs_iters: 20
img_size:
  - 640
  - 512
rect_mode: conventional
"""


def _checkpoint(tmp_path, prefix='module.', train_shape=(256, 320)):
    """A poseNet_*.pth the way train_posenet.py saves it: {'config': <whole training config>, 'state_dict': DataParallel keys}.
    The stored model section deliberately disagrees with the inference settings in the three overridden keys."""
    from rpe_amd import pose_net, synth
    cfg, sd, _ = osynth.posenet_case(synth, opn)
    stored = dict(cfg, image_shape=train_shape, lbgfs_iters=100, use_weights=False)
    path = str(tmp_path / 'poseNet_synthetic.pth')
    torch.save({'config': {'model': stored, 'train': {'epochs': 1}}, 'state_dict': {prefix + k: v for k, v in sd.items()}}, path)
    return path, cfg, sd


def test_pose_estimator_from_checkpoint_path_applies_the_reference_overrides(tmp_path):
    from rpe_amd import pose_estimator, synth
    path, cfg, sd = _checkpoint(tmp_path)
    slam = yaml.safe_load(INFER_F2F_YAML)['slam']
    assert set(slam) == {'frame2frame', 'checkpoint', 'dist_thr', 'depth_clipping', 'debug', 'conf_weighing', 'average_pts', 'lbgfs_iters'}
    K = synth.intrinsics(H, W)
    est = pose_estimator.PoseEstimator(slam, K, 1800.0, path, (W, H))
    m = est.model
    # pose_estimator.py:28-30: image_shape = (img_shape[1], img_shape[0]); lbgfs_iters and use_weights from the slam section
    assert tuple(m.config['image_shape']) == (H, W) and m.img_coords.shape[-1] == H * W
    assert m.config['lbgfs_iters'] == 20 and m.pose_head.problem.lbgfs_iters == 20
    assert m.use_weights is True and m.config['use_weights'] is True
    assert m.config['iters'] == cfg['iters']                                # everything else comes from the checkpoint
    assert not m.training and not m.flow.training
    # pose_estimator.py:31-36: 'module.' stripped, strict load: every tensor of the file is in the model, bit for bit
    got = m.state_dict()
    assert set(got) == set(sd)
    for k, v in sd.items():
        assert torch.equal(got[k], v), k
    # :40-43 scale = 1 / depth_clipping[1], baseline and intrinsics buffers
    assert float(est.scale) == pytest.approx(1 / 250) and tuple(est.intrinsics.shape) == (1, 3, 3) and float(est.baseline[0]) == 1800.0
    # un-prefixed keys (a checkpoint saved without DataParallel) load the same way
    path2, _, _ = _checkpoint(tmp_path, prefix='')
    est2 = pose_estimator.PoseEstimator(slam, K, 1800.0, path2, (W, H))
    assert all(torch.equal(est2.model.state_dict()[k], v) for k, v in sd.items())
    # frame-to-model tracking is refused, not silently run as f2f
    with pytest.raises(NotImplementedError):
        pose_estimator.PoseEstimator(dict(slam, frame2frame=False), K, 1800.0, path, (W, H))
    # a checkpoint with a missing / unexpected tensor fails loudly like the reference's strict load_state_dict
    ck = torch.load(path, map_location='cpu')
    ck['state_dict'].pop(next(iter(ck['state_dict'])))
    with pytest.raises(RuntimeError):
        pose_estimator.PoseEstimator(slam, K, 1800.0, ck, (W, H))


def test_init_from_raft_loads_an_upstream_style_file(tmp_path):
    """raft-things.pth is a bare DataParallel state dict of RAFT: module.fnet.*, module.cnet.*, module.update_block.*
    (pose_net.py:137-147 strips the prefix and loads it into self.flow, strictly)."""
    from rpe_amd import pose_net, raft, synth
    cfg = synth.model_config(H, W)
    torch.manual_seed(7)
    src = raft.RAFT(cfg)
    with torch.no_grad():
        for p in src.parameters():
            p.add_(0.01 * torch.randn_like(p))
        for b_ in src.buffers():
            if b_.dtype.is_floating_point:
                b_.add_(0.1 * torch.rand_like(b_))
    keys = list(src.state_dict())
    for must in ('fnet.conv1.weight', 'fnet.layer2.0.downsample.0.weight', 'cnet.norm1.running_mean', 'cnet.layer3.1.norm2.weight',
                 'update_block.encoder.convc1.weight', 'update_block.gru.convq2.bias', 'update_block.flow_head.conv2.weight',
                 'update_block.mask.2.weight'):
        assert must in keys, must                                            # upstream princeton-vl/RAFT parameter names
    path = str(tmp_path / 'raft-things.pth')
    torch.save({'module.' + k: v for k, v in src.state_dict().items()}, path)
    model = pose_net.PoseNet(cfg)
    before = {k: v.clone() for k, v in model.state_dict().items() if not k.startswith('flow.')}
    assert model.init_from_raft(path) is model
    for k, v in src.state_dict().items():
        assert torch.equal(model.flow.state_dict()[k], v), k
    for k, v in before.items():                                              # heads and loss_weight are untouched
        assert torch.equal(model.state_dict()[k], v), k
    # as in the reference, loading does not change modes: batch norm is frozen from construction (pose_net.py:22), and train()
    # keeps the flow network in eval mode (:156-159)
    assert all(not m.training for m in model.flow.modules() if isinstance(m, torch.nn.BatchNorm2d))
    assert not model.train().flow.training and model.weight_head_2d.training
    bad = {k: v for k, v in src.state_dict().items() if 'convq2' not in k}
    torch.save(bad, path)
    with pytest.raises(RuntimeError):
        model.init_from_raft(path)


@pytest.mark.gpu
def test_path_built_estimator_reproduces_the_reference_tracker_golden(tmp_path, rpe):
    from rpe_amd import pose_estimator, synth

    def _unpack(bits, shape):
        n = int(np.prod(shape))
        return torch.from_numpy(np.unpackbits(bits.numpy())[:n].astype(bool).reshape(shape))

    path, _, _ = _checkpoint(tmp_path)
    g = load_golden('tracker.npz')
    frames, K, bf = osynth.tracker_case(synth)
    slam = dict(yaml.safe_load(INFER_F2F_YAML)['slam'], lbgfs_iters=8)      # the golden run's iteration count
    est = pose_estimator.PoseEstimator(slam, K, bf, path, (W, H)).cuda()
    for i, (l, r, m) in enumerate(frames):
        P, _, _, _ = est(l.cuda(), r.cuda(), m.clone().cuda())
        d = float((P.data.cpu().reshape(7) - g['abs_poses'][i]).abs().max())
        print(f'frame {i}: abs pose diff {d:.2e} mm')
        assert d <= 2e-3
        assert int((est.frame.mask.cpu() != _unpack(g['masks'][i], m.shape)).sum()) <= 50


@pytest.mark.gpu
def test_init_from_raft_then_flow_runs_on_the_hip_path(tmp_path, rpe):
    """The loaded file is what the kernels compute with: flow of the file-initialised PoseNet == flow of the source RAFT."""
    from rpe_amd import pose_net, raft, synth
    cfg = synth.model_config(H, W)
    src = synth.init_synthetic_weights(raft.RAFT(cfg), seed=77).eval()
    path = str(tmp_path / 'raft-things.pth')
    torch.save({'module.' + k: v for k, v in src.state_dict().items()}, path)
    model = pose_net.PoseNet(cfg).init_from_raft(path).eval().cuda()
    fr = synth.stereo_frames(3, 1, H, W)
    a, b_ = fr['image1l'].cuda(), fr['image2l'].cuda()
    f_model = model.flow(a, b_)[0][-1]
    f_src = src.cuda()(a, b_)[0][-1]
    assert torch.equal(f_model, f_src)
