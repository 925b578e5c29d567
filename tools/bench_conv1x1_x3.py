#!/usr/bin/env python3
"""rpe_conv1x1_x3 (bf16x3 split on the 16-bit matrix cores) vs rpe_conv1x1 (f32 matrix cores): error against an f64 reference on small and
ragged shapes, then time at the bench shape (convc1: 324 -> 256 on 32 maps of 64 x 80)."""
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
with torch.no_grad():
    for (nb, ci, co, H, W, relu) in ((2, 16, 128, 16, 16, True), (1, 40, 96, 20, 28, False), (3, 324, 256, 8, 12, True), (2, 336, 576, 64, 80, False), (N, 324, 256, 64, 80, True)):
        x = torch.randn(nb, ci, H, W, device=dev); w = torch.randn(co, ci, 1, 1, device=dev) * 0.05; bias = torch.randn(co, device=dev)
        mode = ops.CONV_RELU if relu else ops.CONV_LINEAR
        o1 = torch.full((nb, co, H, W), float('nan'), device=dev); o2 = torch.full((nb, co, H, W), float('nan'), device=dev)
        f1 = ops.conv1x1(x, ops.PackedConv1x1(w, bias), mode, o1, prepare=True); f2 = ops.conv1x1(x, ops.PackedConv1x1X3(w, bias), mode, o2, prepare=True)
        f1(); f2(); torch.cuda.synchronize()
        m = min(nb, 2)
        ref = torch.nn.functional.conv2d(x[:m].double(), w.double(), bias.double())
        ref = ref.relu() if relu else ref
        e1, e2 = o1[:m].double() - ref, o2[:m].double() - ref
        line = '%3d->%3d %2dx%dx%d relu %d: f32 max %.2e rms %.2e | x3 max %.2e rms %.2e nan %d' % (ci, co, nb, H, W, relu, e1.abs().max().item(), e1.pow(2).mean().sqrt().item(),
                                                                                                  e2.abs().max().item(), e2.pow(2).mean().sqrt().item(), int(torch.isnan(o2).sum().item()))
        if nb == N and N > 3:
            acc = [[], []]
            for rep in range(3):
                for i in ((0, 1), (1, 0), (0, 1))[rep]:
                    acc[i].append(t((f1, f2)[i]))
            t1, t2 = sorted(acc[0])[1], sorted(acc[1])[1]
            line += '   || f32 %7.1f us | x3 %7.1f us (%.2fx; %5.1f TF direct-equivalent)' % (t1, t2, t1 / t2, 2.0 * nb * H * W * ci * co / t2 / 1e6)
        print(line, flush=True)
        if nb == N and N > 3 and hasattr(rpe_amd._lib.lib(), 'rpe_debug_g3_timing'):
            import ctypes
            buf = (ctypes.c_ulonglong * 8)()
            rpe_amd._lib.lib().rpe_debug_g3_timing(buf)
            print('        wave 0 of a mid-grid workgroup: prologue %d, loop %d (%d per step), epilogue %d cycles' % (buf[0], buf[1], buf[1] // max(buf[3], 1), buf[2]))
