#!/usr/bin/env python3
"""The pose solve alone: one persistent launch for all evaluations vs one launch per evaluation (RPE_SOLVE_LAUNCH_PER_EVALUATION), L-BFGS
and Gauss-Newton, 16 rows (the bench step) and 1 row x 20 iterations (the tracker), 640x512.  Median of 30 solves by HIP events."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops, synth, pose_net
dev = torch.device('cuda:0'); H, W = 512, 640
for B, iters in ((16, 8), (1, 20)):
    model = synth.init_synthetic_weights(pose_net.PoseNet(synth.model_config(H, W, lbgfs_iters=iters)), seed=1234).eval().to(dev)
    fr = synth.stereo_frames(1000, B, H, W)
    g = {k: v.to(dev) for k, v in synth.infer_args(fr).items()}
    s = model.stages(**g)
    lw = model.loss_weight.detach()[None].repeat(B, 1)
    args = (s['time_flow'], s['pcl1'], s['pcl2w'], s['w2d'], s['w3d'], g['mask1'].bool(), s['mask2w'], s['intrinsics'], lw)
    for mode in (0, 1):
        res = {}
        for persistent in (True, False, True, False):
            for _ in range(20):
                ops.pose_solve(*args, iters=iters, mode=mode, persistent=persistent)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
            for a, b in ev:
                a.record(); T, _, _, info = ops.pose_solve(*args, iters=iters, mode=mode, persistent=persistent); b.record()
            torch.cuda.synchronize()
            res.setdefault(persistent, []).append(statistics.median(a.elapsed_time(b) for a, b in ev) * 1e3)
        alg = B * H * W * 42 * iters
        print(f"B={B} iters={iters} {'GN' if mode else 'L-BFGS'}: one launch {res[True][0]:.0f} / {res[True][1]:.0f} us ({alg / res[True][1] / 8e6:.3f} of 8 TB/s), "
              f"launch per evaluation {res[False][0]:.0f} / {res[False][1]:.0f} us ({alg / res[False][1] / 8e6:.3f}); evals run {info[:, 1].tolist()[:4]}")
