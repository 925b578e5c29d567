#!/usr/bin/env python3
"""One RAFT pass + PoseNet.infer at a map size the tuned kernels refuse (352x360: 1/8 map 44 x 45), for rocprofv3 --kernel-trace --stats:
the kernel table must hold nothing but this library's kernels (no miopen* / Cijk_*)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd  # noqa: F401
from rpe_amd import pose_net, synth

H, W = 352, 360
cfg = synth.model_config(H, W, iters=12, lbgfs_iters=8, use_weights=True)
model = synth.init_synthetic_weights(pose_net.PoseNet(cfg)).eval().cuda()
a = {k: v.cuda() for k, v in synth.infer_args(synth.stereo_frames(3, 2, H, W)).items()}
for _ in range(3):
    pose = model.infer(**{k: v.clone() for k, v in a.items()})
torch.cuda.synchronize()
print('pose', pose.data.cpu().tolist())
