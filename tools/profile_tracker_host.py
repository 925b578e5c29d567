#!/usr/bin/env python3
"""Host-side cost of one frame of sequential tracking: cProfile over 20 frames (the GPU runs beside it; one synchronisation per frame)."""
import cProfile, os, pstats, sys, time, warnings, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import pose_net, synth, pose_estimator
dev = torch.device('cuda:0'); H, W = 512, 640
model = synth.init_synthetic_weights(pose_net.PoseNet(synth.model_config(H, W, lbgfs_iters=20)), seed=1234).eval().to(dev)
K = torch.tensor([[1.1 * W, 0, W / 2], [0, 1.1 * W, H / 2], [0, 0, 1.0]])
fr = synth.stereo_frames(7, 25, H, W)
L, R = fr['image1l'].to(dev), fr['image2r'].to(dev)
M = torch.ones(1, 1, H, W, dtype=torch.bool, device=dev)
est = pose_estimator.PoseEstimator(dict(frame2frame=True, depth_clipping=[1, 250], lbgfs_iters=20, conf_weighing=True), K, 4000.0, model, (W, H)).to(dev)
def run(n):
    est.reset()
    for t in range(n):
        est(L[t:t + 1], R[t:t + 1], M.clone())
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    run(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(21); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('%.2f ms per frame wall' % (dt / 20 * 1e3))
    pr = cProfile.Profile(); pr.enable(); run(21); torch.cuda.synchronize(); pr.disable()
    st = pstats.Stats(pr); st.sort_stats('tottime'); st.print_stats(22)
