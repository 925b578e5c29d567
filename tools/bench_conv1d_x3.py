#!/usr/bin/env python3
"""rpe_conv_wino1d_x3 (bf16x3 split on the 16-bit matrix cores) vs rpe_conv_wino1d (f32 matrix cores): the GRU's four convolutions with
their gate epilogues -- error against an f64 reference, then time at the bench shape."""
import os, sys
import torch
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


def run(kind, x, pw, add, hid, z, c, mode, out, out2):
    if mode == ops.CONV_GATE_ZR:
        return ops.conv_wino1d(x, pw, mode, out, out2=out2, add=add, hidden=hid, gate_channels=c, prepare=True)
    if mode == ops.CONV_GATE_H:
        return ops.conv_wino1d(x, pw, mode, out, add=add, hidden=hid, zgate=z, prepare=True)
    return ops.conv_wino1d(x, pw, mode, out, add=add, prepare=True)


def reference(x, w, pad, add, hid, z, c, mode):
    v = torch.nn.functional.conv2d(x.double(), w.double(), None, padding=pad) + add.double()
    if mode == ops.CONV_GATE_ZR:
        s = torch.sigmoid(v)
        return s[:, :c], s[:, c:] * hid.double()
    if mode == ops.CONV_GATE_H:
        return (1 - z.double()) * hid.double() + z.double() * torch.tanh(v), None
    return (v.relu() if mode == ops.CONV_RELU else v), None


dev = torch.device('cuda:0'); torch.manual_seed(0)
N = int(os.environ.get('CONV_N', 32))
with torch.no_grad():
    for (nb, ci, H, W) in ((2, 16, 16, 16), (1, 48, 20, 28), (2, 256, 64, 80), (N, 256, 64, 80)):
        c = 128
        for vert in (False, True):
            for mode, co in ((ops.CONV_LINEAR, 128), (ops.CONV_RELU, 256), (ops.CONV_GATE_ZR, 256), (ops.CONV_GATE_H, 128)):
                x = torch.randn(nb, ci, H, W, device=dev)
                w = torch.randn(co, ci, *((5, 1) if vert else (1, 5)), device=dev) * 0.05
                add = torch.randn(nb, co, H, W, device=dev); hid = torch.randn(nb, c, H, W, device=dev); z = torch.rand(nb, c, H, W, device=dev)
                outs = []
                fns = []
                for P in (ops.PackedWino1d, ops.PackedWino1dX3):
                    o = torch.full((nb, c if mode == ops.CONV_GATE_ZR else co, H, W), float('nan'), device=dev)
                    o2 = torch.full((nb, c, H, W), float('nan'), device=dev) if mode == ops.CONV_GATE_ZR else None
                    f = run(None, x, P(w), add, hid, z, c, mode, o, o2)
                    f(); fns.append(f); outs.append((o, o2))
                torch.cuda.synchronize()
                m = min(nb, 2)
                r1, r2 = reference(x[:m], w, (2, 0) if vert else (0, 2), add[:m], hid[:m], z[:m], c, mode)
                def err(o, r):
                    e = o[:m].double() - r
                    return e.abs().max().item(), e.pow(2).mean().sqrt().item()
                e1, e2 = err(outs[0][0], r1), err(outs[1][0], r1)
                line = '%s %3d->%3d %2dx%dx%d mode %d: f32 max %.2e rms %.2e | x3 max %.2e rms %.2e nan %d' % (
                    '5x1' if vert else '1x5', ci, co, nb, H, W, mode, *e1, *e2, int(torch.isnan(outs[1][0]).sum().item()))
                if r2 is not None:
                    line += ' | out2 f32 %.2e x3 %.2e nan %d' % (err(outs[0][1], r2)[0], err(outs[1][1], r2)[0], int(torch.isnan(outs[1][1]).sum().item()))
                if nb == N and N > 2:
                    acc = [[], []]
                    for rep in range(3):
                        for i in ((0, 1), (1, 0), (0, 1))[rep]:
                            acc[i].append(t(fns[i]))
                    t1, t2 = sorted(acc[0])[1], sorted(acc[1])[1]
                    line += '   || f32 %7.1f us | x3 %7.1f us (%.2fx; %5.1f TF direct-equivalent)' % (t1, t2, t1 / t2, 2.0 * nb * H * W * ci * co * 5 / t2 / 1e6)
                print(line, flush=True)
                if nb == N and N > 2 and hasattr(rpe_amd._lib.lib(), 'rpe_debug_y3_timing'):               # -DY3_TIMING variant builds only
                    import ctypes
                    buf = (ctypes.c_ulonglong * 16)()
                    rpe_amd._lib.lib().rpe_debug_y3_timing(buf)
                    n = max(buf[11], 1)
                    print('        cycles per step (wave 0, mid-grid workgroup, %d steps): stage 0 %d | stage 1 %d | barrier %d | stage 2 %d | stage 3 %d ;  prologue %d loop %d epilogue %d (first half: wait + fetch %d, exchange %d, barrier %d, first 4 quads %d, other 12 %d)' % (
                        n, *[buf[i] // n for i in range(1, 6)], buf[8], buf[9], buf[10], buf[12], buf[13], buf[14], buf[15], buf[7]))
