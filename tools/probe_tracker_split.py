#!/usr/bin/env python3
"""Host-enqueue time vs GPU time of one frame of sequential tracking (VERDICT r5 #1a).  Per frame: perf_counter from the start of
PoseEstimator.forward to the return of the last enqueue (rpe_pose_gate_chain, just before the success-flag sync), HIP events around
the frame, and the wall time.  Run with the GPU box otherwise idle."""
import os, sys, time, warnings, statistics as st, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import pose_net, synth, pose_estimator, ops
dev = torch.device('cuda:0'); H, W = 512, 640
model = synth.init_synthetic_weights(pose_net.PoseNet(synth.model_config(H, W, lbgfs_iters=20)), seed=1234).eval().to(dev)
K = torch.tensor([[1.1 * W, 0, W / 2], [0, 1.1 * W, H / 2], [0, 0, 1.0]])
fr = synth.stereo_frames(7, 25, H, W)
L, R = fr['image2l'].to(dev), fr['image2r'].to(dev)
M = torch.ones(1, 1, H, W, dtype=torch.bool, device=dev)
est = pose_estimator.PoseEstimator(dict(frame2frame=True, depth_clipping=[1, 250], lbgfs_iters=20, conf_weighing=True), K, 4000.0, model, (W, H)).to(dev)
mark = {}
real = ops.pose_gate_chain
def gate(*a, **k):
    r = real(*a, **k); mark['t'] = time.perf_counter(); return r
ops.pose_gate_chain = gate
pose_estimator.ops.pose_gate_chain = gate
def run(n, rec=None):
    est.reset()
    for t in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        m = M.clone()
        t0 = time.perf_counter(); e0.record()
        est(L[t:t + 1], R[t:t + 1], m)
        e1.record(); t1 = time.perf_counter()
        if rec is not None and t > 0:
            rec.append((mark['t'] - t0, t1 - t0, e0, e1))
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    run(5); torch.cuda.synchronize()
    for rep in range(3):
        rec = []
        t0 = time.perf_counter(); run(25, rec); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        host = [a * 1e3 for a, _, _, _ in rec]; wall = [b * 1e3 for _, b, _, _ in rec]; gpu = [a.elapsed_time(b) for _, _, a, b in rec]
        print(f'rep {rep}: {24 / dt:.1f} frames/s incl. first frame | per frame median: host enqueue {st.median(host):.2f} ms, wall {st.median(wall):.2f} ms, '
              f'GPU (events) {st.median(gpu):.2f} ms | max host {max(host):.2f}')
