#!/usr/bin/env python3
"""Encoder-shaped stride-1 3x3 convolutions: library vs rpe_conv_fused (plain epilogue)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops
F = torch.nn.functional
def t(fn, reps=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
dev = torch.device('cuda:0'); torch.manual_seed(0)
with torch.no_grad():
    for name, N, c, H, W in (('layer1 64->64 @256x320 x48', 48, 64, 256, 320), ('layer1 x32', 32, 64, 256, 320),
                             ('layer2 96->96 @128x160 x48', 48, 96, 128, 160), ('layer3 128->128 @64x80 x48', 48, 128, 64, 80)):
        x = torch.randn(N, c, H, W, device=dev); w = torch.randn(c, c, 3, 3, device=dev) * 0.05
        out = torch.empty(N, c, H, W, device=dev); pc = ops.PackedConv(w, None)
        flop = 2.0 * N * H * W * c * c * 9
        tl = t(lambda: F.conv2d(x, w, None, padding=1)); to = t(lambda: ops.conv_fused(x, pc, ops.CONV_RELU, out))
        print('%-30s library %8.1f us (%5.1f TF)   fused %8.1f us (%5.1f TF)' % (name, tl, flop / tl / 1e6, to, flop / to / 1e6))
