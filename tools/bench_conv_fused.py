#!/usr/bin/env python3
"""rpe_conv_fused vs the library convolution (+ the separate epilogue kernels) on the update block's shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops
F = torch.nn.functional

def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

dev = torch.device('cuda:0'); torch.manual_seed(0)
N, H, W = int(os.environ.get("CONV_N", 32)), 64, 80
cases = [('gru zr 1x5', 256, 256, 1, 5), ('gru q 1x5', 256, 128, 1, 5), ('gru zr 5x1', 256, 256, 5, 1), ('gru q 5x1', 256, 128, 5, 1),
         ('convc1 1x1', 324, 256, 1, 1), ('convc2 3x3', 256, 192, 3, 3), ('convf2 3x3', 128, 64, 3, 3), ('conv 3x3', 256, 126, 3, 3),
         ('fh1 3x3', 128, 256, 3, 3)]
only = sys.argv[1] if len(sys.argv) > 1 else None
with torch.no_grad():
    for name, ci, co, kh, kw in cases:
        if only and only not in name: continue
        x = torch.randn(N, ci, H, W, device=dev); w = torch.randn(co, ci, kh, kw, device=dev) * 0.05; bias = torch.randn(co, device=dev)
        out = torch.empty(N, co, H, W, device=dev)
        pc = ops.PackedConv(w, bias)
        flop = 2.0 * N * H * W * ci * co * kh * kw
        t_lib = t(lambda: ops.bias_act(F.conv2d(x, w, None, padding=(kh // 2, kw // 2)), bias))
        t_own = t(lambda: ops.conv_fused(x, pc, ops.CONV_RELU, out))
        ref = ops.bias_act(F.conv2d(x, w, None, padding=(kh // 2, kw // 2)), bias)
        print('%-12s %3d->%3d  library+bias_act %7.1f us   fused %7.1f us (%5.1f TF)   maxdiff %.1e' % (
            name, ci, co, t_lib, t_own, flop / t_own / 1e6, (ref - out).abs().max().item()))
