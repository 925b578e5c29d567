#!/usr/bin/env python3
"""Winograd F(2x2,3x3) (rpe_conv_wino) vs the direct implicit GEMM (rpe_conv_fused) on the update block's 3x3 shapes."""
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
N, H, W = int(os.environ.get('CONV_N', 32)), 64, 80
with torch.no_grad():
    only = os.environ.get('CONV_ONLY')
    for name, ci, co in (('convc2', 256, 192), ('convf2', 128, 64), ('conv', 256, 126), ('fh1', 128, 256)):
        if only and name != only: continue
        x = torch.randn(N, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05; bias = torch.randn(co, device=dev)
        o1 = torch.empty(N, co, H, W, device=dev); o2 = torch.empty_like(o1)
        pc, pw = ops.PackedConv(w, bias), ops.PackedWino(w, bias)
        flop = 2.0 * N * H * W * ci * co * 9
        td = t(lambda: ops.conv_fused(x, pc, ops.CONV_RELU, o1))
        tw = t(lambda: ops.conv_wino(x, pw, ops.CONV_RELU, o2))
        print('%-7s %3d->%3d  direct %7.1f us (%5.1f TF)   winograd %7.1f us (%5.1f TF executed, %5.1f effective)   maxdiff %.1e (|out| %.1f)' % (
            name, ci, co, td, flop / td / 1e6, tw, flop / 2.25 / tw / 1e6, flop / tw / 1e6, (o1 - o2).abs().max().item(), o1.abs().max().item()))
        if hasattr(rpe_amd._lib.lib(), 'rpe_debug_wino_timing'):               # -DWINO_TIMING variant builds only
            import ctypes
            buf = (ctypes.c_ulonglong * 8)()
            rpe_amd._lib.lib().rpe_debug_wino_timing(buf)
            n = max(buf[6], 1)
            print('        cycles per step, one wave: g0 %.0f  g1 %.0f  g2 %.0f  wait %.0f  barrier %.0f  g3 %.0f' % tuple(buf[i] / n for i in range(6)))
    print('--- GRU convolutions (256 varying input channels; direct implicit GEMM vs Winograd F(4,5) along the axis, gate epilogues)')
    c = 128
    for name, kh, kw, co, mode in (('zr 1x5', 1, 5, 256, ops.CONV_GATE_ZR), ('q 1x5', 1, 5, 128, ops.CONV_GATE_H), ('zr 5x1', 5, 1, 256, ops.CONV_GATE_ZR),
                                   ('q 5x1', 5, 1, 128, ops.CONV_GATE_H)):
        if only and only != 'gru': continue
        hx = torch.randn(N, 256, H, W, device=dev) * 0.5; rhx = hx.clone(); z = torch.rand(N, c, H, W, device=dev)
        w = torch.randn(co, 256, kh, kw, device=dev) * 0.03; add = torch.randn(N, co, H, W, device=dev) * 0.3
        pc, pw = ops.PackedConv(w), ops.PackedWino1d(w)
        flop = 2.0 * N * H * W * 256 * co * 5
        res = []
        for conv, pk in ((ops.conv_fused, pc), (ops.conv_wino1d, pw)):
            if mode == ops.CONV_GATE_ZR:
                zo, ro = torch.empty(N, c, H, W, device=dev), torch.empty(N, c, H, W, device=dev)
                fn = conv(hx, pk, mode, zo, out2=ro, add=add, hidden=hx[:, :c], gate_channels=c, prepare=True)
                out = (zo, ro)
            else:
                ho = torch.empty(N, c, H, W, device=dev)
                fn = conv(rhx, pk, mode, ho, add=add, hidden=hx[:, :c], zgate=z, prepare=True)
                out = (ho,)
            res.append((t(fn), out))
        (td, od), (tw, ow) = res
        print('%-7s 256->%3d  direct %7.1f us (%5.1f TF)   winograd %7.1f us (%5.1f TF executed, %5.1f effective)   maxdiff %.1e' % (
            name, co, td, flop / td / 1e6, tw, flop / 2.5 / tw / 1e6, flop / tw / 1e6, max((a - b).abs().max().item() for a, b in zip(od, ow))))
    if only: sys.exit(0)
    print('--- encoder layers (fnet: bias + instance-norm moments; 48 images)')
    for name, c, hh, ww in (('layer1', 64, 256, 320), ('layer2', 96, 128, 160), ('layer3', 128, 64, 80)):
        nb = 48
        x = torch.randn(nb, c, hh, ww, device=dev); w = torch.randn(c, c, 3, 3, device=dev) * 0.05; bias = torch.randn(c, device=dev)
        o1 = torch.empty(nb, c, hh, ww, device=dev); o2 = torch.empty_like(o1)
        pc, pw = ops.PackedConv(w, None), ops.PackedWino(w, None)
        sd, sw = ops.conv_stats_buffer(nb, c, hh, ww, dev), ops.conv_wino_stats_buffer(nb, c, hh, ww, dev)
        flop = 2.0 * nb * hh * ww * c * c * 9
        ones = torch.ones(c, device=dev)
        # (round-robin over the variants: a fixed order hands the last one a warmer, slower chip)
        fns = [lambda: ops.conv_fused(x, pc, ops.CONV_LINEAR, o1, bias=bias, stats=sd), lambda: ops.conv_wino(x, pw, ops.CONV_LINEAR, o2, bias=bias, stats=sw),
               lambda: ops.conv_wino(x, pw, ops.CONV_LINEAR, o2, bias=bias), lambda: ops.conv_wino(x, pw, ops.CONV_LINEAR, o2, bias=bias, scale=ones)]
        acc = [[] for _ in fns]
        for rep in range(3):
            for i in ([0, 1, 2, 3], [3, 2, 1, 0], [1, 3, 0, 2])[rep]:
                acc[i].append(t(fns[i], reps=6))
        td, tw, tn, te = (sorted(a)[1] for a in acc)
        fns[1]()                                          # (o2 = the moments variant's output for the comparison below)
        print('        (encoder instantiation without moments: %7.1f us)' % te)
        print('%-7s %3d ch %3dx%3d  direct %7.1f us (%5.1f TF)   winograd %7.1f us (%5.1f TF executed, %5.1f effective; without moments %7.1f us)   maxdiff %.1e' % (
            name, c, hh, ww, td, flop / td / 1e6, tw, flop / 2.25 / tw / 1e6, flop / tw / 1e6, tn, (o1 - o2).abs().max().item()))
