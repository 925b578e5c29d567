#!/usr/bin/env python3
"""Sequential tracking throughput (the reference's real use: one stereo frame at a time, batch 1):
PoseEstimator over a synthetic sequence, with and without streaming encoder-feature reuse."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import pose_estimator, pose_net, synth

dev = torch.device('cuda:0')
H, W, F = 512, 640, 24
cfg = synth.model_config(H, W, lbgfs_iters=20)          # configuration/infer_f2f.yaml:11 lbgfs_iters: 20
model = synth.init_synthetic_weights(pose_net.PoseNet(cfg)).eval().to(dev)
fr = synth.stereo_frames(77, F, H, W)
frames = [(fr['image2l'][i:i + 1].to(dev), fr['image2r'][i:i + 1].to(dev), fr['mask2'][i:i + 1].to(dev)) for i in range(F)]
for reuse in (False, True):
    slam = dict(frame2frame=True, depth_clipping=[1, 250], lbgfs_iters=20, conf_weighing=True, reuse_features=reuse)
    for rep in range(2):
        est = pose_estimator.PoseEstimator(slam, fr['K'][0], 7.2 * 250.0, model, (W, H)).to(dev)
        torch.cuda.synchronize(); t = time.perf_counter()
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            for l, r, m in frames:
                est(l, r, m.clone())
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f'reuse_features={reuse}: {F / dt:.1f} frames/s ({1e3 * dt / F:.2f} ms/frame), 640x512, 12 GRU iters, L-BFGS 20')
