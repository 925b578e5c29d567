"""The GRU's two launches of a half step as raft.py issues them -- channel slices of the 256-channel state buffers, the hidden state updated in
place -- on rpe_conv_wino1d and rpe_conv_wino1d_x3, alone and back to back (tools/bench_conv1d_x3.py times the same kernels on contiguous
tensors, where the variant is ~8 % better on z|r: with the slices both take 430-437 us)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
dev = torch.device('cuda:0'); torch.manual_seed(0)
N, c, H, W = 32, 128, 64, 80
with torch.no_grad():
    for scale in (1.0, 0.05):
        hx = torch.randn(N, 2 * c, H, W, device=dev) * scale; rhx = hx.clone()
        z_buf = torch.empty(N, c, H, W, device=dev)
        ctx_zr = torch.randn(N, 2 * c, H, W, device=dev) * scale; ctx_q = torch.randn(N, c, H, W, device=dev) * scale
        for vert in (False, True):
            k = (5, 1) if vert else (1, 5)
            wzr = torch.randn(2 * c, 2 * c, *k, device=dev) * 0.02; wq = torch.randn(c, 2 * c, *k, device=dev) * 0.02
            res = {}
            for name, P in (('f32', ops.PackedWino1d), ('x3', ops.PackedWino1dX3)):
                fzr = ops.conv_wino1d(hx, P(wzr), ops.CONV_GATE_ZR, z_buf, out2=rhx[:, :c], add=ctx_zr, hidden=hx[:, :c], gate_channels=c, prepare=True)
                fq = ops.conv_wino1d(rhx, P(wq), ops.CONV_GATE_H, hx[:, :c], add=ctx_q, hidden=hx[:, :c], zgate=z_buf, prepare=True)
                res[name] = (fzr, fq)
            acc = {}
            for rep in range(3):
                for name in (('f32', 'x3'), ('x3', 'f32'), ('f32', 'x3'))[rep]:
                    acc.setdefault(name, []).append((t(res[name][0]), t(res[name][1])))
            for name in ('f32', 'x3'):
                zs = sorted(a for a, _ in acc[name])[1]; qs = sorted(b for _, b in acc[name])[1]
                print('scale %.2f %s %s: z|r %.1f us, q %.1f us (slices of the 256-channel state buffers, in place)' % (scale, '5x1' if vert else '1x5', name, zs, qs), flush=True)
            # the GRU's own order: zr, q back to back, alternating kernels as in a step
            def seq(name):
                res[name][0](); res[name][1]()
            print('   pair back to back: f32 %.1f us, x3 %.1f us' % (t(lambda: seq('f32')), t(lambda: seq('x3'))), flush=True)
