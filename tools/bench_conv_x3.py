#!/usr/bin/env python3
"""rpe_conv_wino_x3 (bf16x3 split on the 16-bit matrix cores) vs rpe_conv_wino (f32 matrix cores): time and error against an f64 reference."""
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


dev = torch.device('cuda:0'); torch.manual_seed(0)
N = int(os.environ.get('CONV_N', 32))
only = os.environ.get('CONV_ONLY')
with torch.no_grad():
    # small exact check first: one image, f64 reference
    for (ci, co, H, W) in ((16, 64, 16, 16), (32, 96, 32, 48), (64, 126, 24, 40), (256, 192, 64, 80)):
        x = torch.randn(2, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05; bias = torch.randn(co, device=dev)
        ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), padding=1).relu()
        o1 = torch.empty(2, co, H, W, device=dev); o2 = torch.full_like(o1, float('nan'))
        ops.conv_wino(x, ops.PackedWino(w, bias), ops.CONV_RELU, o1)
        ops.conv_wino(x, ops.PackedWinoX3(w, bias), ops.CONV_RELU, o2)
        torch.cuda.synchronize()
        e1, e2 = (o1.double() - ref), (o2.double() - ref)
        print('check %3d->%3d %dx%d: f32 max %.2e rms %.2e | x3 max %.2e rms %.2e | nan %d' % (ci, co, H, W, e1.abs().max().item(), e1.pow(2).mean().sqrt().item(),
              e2.abs().max().item(), e2.pow(2).mean().sqrt().item(), int(torch.isnan(o2).sum().item())), flush=True)
    H, W = 64, 80
    for name, ci, co in (('convc2', 256, 192), ('convf2', 128, 64), ('conv', 256, 126), ('fh1', 128, 256)):
        if only and name != only: continue
        x = torch.randn(N, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05; bias = torch.randn(co, device=dev)
        o1 = torch.empty(N, co, H, W, device=dev); o2 = torch.empty_like(o1)
        pw, px = ops.PackedWino(w, bias), ops.PackedWinoX3(w, bias)
        flop = 2.0 * N * H * W * ci * co * 9
        f1 = ops.conv_wino(x, pw, ops.CONV_RELU, o1, prepare=True); f2 = ops.conv_wino(x, px, ops.CONV_RELU, o2, prepare=True)
        acc = [[], []]
        for rep in range(3):
            for i in ((0, 1), (1, 0), (0, 1))[rep]:
                acc[i].append(t((f1, f2)[i]))
        t1, t2 = sorted(acc[0])[1], sorted(acc[1])[1]
        ref = torch.nn.functional.conv2d(x[:2].double(), w.double(), bias.double(), padding=1).relu()
        e1, e2 = (o1[:2].double() - ref), (o2[:2].double() - ref)
        print('%-7s %3d->%3d  f32 %7.1f us | x3 %7.1f us (%.2fx; %5.1f TF direct-equivalent)   err f32 max %.2e rms %.2e | x3 max %.2e rms %.2e' % (
            name, ci, co, t1, t2, t1 / t2, flop / t2 / 1e6, e1.abs().max().item(), e1.pow(2).mean().sqrt().item(), e2.abs().max().item(), e2.pow(2).mean().sqrt().item()), flush=True)
        if hasattr(rpe_amd._lib.lib(), 'rpe_debug_x3_timing'):               # -DX3_TIMING variant builds only
            import ctypes
            buf = (ctypes.c_ulonglong * 24)()
            rpe_amd._lib.lib().rpe_debug_x3_timing(buf)
            n = max(buf[15], 1)
            print('        cycles per step, wave 0 of a mid-grid workgroup (%d steps): ' % buf[15] + ' '.join('%d:%.0f' % (i, buf[i] / n) for i in range(12)))
            print('        prologue %d  loop %d  epilogue %d cycles' % (buf[12], buf[13], buf[14]))
            print('        prologue: set-up %d | requests + wait for the first patch %d | patch-up + barrier %d | first fragments %d ;  epilogue: Z to LDS %d | barrier %d | final pass %d' % (
                buf[18], buf[19], buf[20], buf[21], buf[16], buf[17], buf[14] - buf[16] - buf[17]))
    if not only:
        print('--- encoder layers (bias + instance-norm moments; 48 images)')
        for name, c, hh, ww in (('layer1', 64, 256, 320), ('layer2', 96, 128, 160), ('layer3', 128, 64, 80)):
            nb = 48
            x = torch.randn(nb, c, hh, ww, device=dev); w = torch.randn(c, c, 3, 3, device=dev) * 0.05; bias = torch.randn(c, device=dev)
            o1 = torch.empty(nb, c, hh, ww, device=dev); o2 = torch.empty_like(o1)
            pw, px = ops.PackedWino(w, None), ops.PackedWinoX3(w, None)
            s1, s2 = ops.conv_wino_stats_buffer(nb, c, hh, ww, dev), ops.conv_wino_stats_buffer(nb, c, hh, ww, dev)
            f1 = lambda: ops.conv_wino(x, pw, ops.CONV_LINEAR, o1, bias=bias, stats=s1)
            f2 = lambda: ops.conv_wino(x, px, ops.CONV_LINEAR, o2, bias=bias, stats=s2)
            acc = [[], []]
            for rep in range(3):
                for i in ((0, 1), (1, 0), (0, 1))[rep]:
                    acc[i].append(t((f1, f2)[i], reps=6))
            t1, t2 = sorted(acc[0])[1], sorted(acc[1])[1]
            m1 = ops.instnorm_finalize(s1, hh * ww); m2 = ops.instnorm_finalize(s2, hh * ww)
            print('%-7s %3d ch %3dx%3d  f32 %7.1f us | x3 %7.1f us (%.2fx)   maxdiff out %.1e  mean %.1e  inv-std rel %.1e' % (
                name, c, hh, ww, t1, t2, t1 / t2, (o1 - o2).abs().max().item(), (m1[..., 0] - m2[..., 0]).abs().max().item(),
                ((m1[..., 1] - m2[..., 1]).abs() / m1[..., 1]).max().item()), flush=True)
