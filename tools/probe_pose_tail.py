import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops, synth, pose_net, _lib
dev = torch.device('cuda:0'); H, W = 512, 640
for B, iters in ((16, 8), (1, 20)):
    model = synth.init_synthetic_weights(pose_net.PoseNet(synth.model_config(H, W, lbgfs_iters=iters)), seed=1234).eval().to(dev)
    fr = synth.stereo_frames(1000, B, H, W)
    g = {k: v.to(dev) for k, v in synth.infer_args(fr).items()}
    s = model.stages(**g)
    lw = model.loss_weight.detach()[None].repeat(B, 1)
    args = (s['time_flow'], s['pcl1'], s['pcl2w'], s['w2d'], s['w3d'], g['mask1'].bool(), s['mask2w'], s['intrinsics'], lw)
    L = _lib.lib(); L.rpe_pose_probe_read.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
    import hashlib
    for mode in (0, 1):
        it = iters
        for _ in range(30):
            ops.pose_solve(*args, iters=it, mode=mode)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); T, _, _, info = ops.pose_solve(*args, iters=it, mode=mode); e1.record(); torch.cuda.synchronize()
        out = (ctypes.c_longlong * 256)(); L.rpe_pose_probe_read(out)
        t = [[out[8 * e + k] * 0.01 for k in range(8)] for e in range(min(it, 32))]
        print(f'B={B} mode={mode} iters={it}: solve {e0.elapsed_time(e1) * 1e3:.0f} us; per evaluation of row 0 (us):')
        for e in range(1, min(it, 32) - 1):
            a = t[e]
            print(f'  eval {e}: wg0 pixel loop {a[7] - a[0]:.2f} | last wg leaves the loop +{a[5] - a[7]:.2f} | -> tail start {a[1] - a[5]:.2f} | loads+sum {a[2] - a[1]:.2f} | '
                  f'update {a[3] - a[2]:.2f} | write-back {a[4] - a[3]:.2f} | acks+epoch {a[6] - a[4]:.2f} | wg0 sees it {t[e + 1][0] - a[6]:.2f} | whole {t[e + 1][0] - a[0]:.2f}')
            if e >= 4: break
