#!/usr/bin/env python3
"""k_conv_igemm rate vs reduction length (cin) at fixed output size: separates per-step from per-workgroup costs."""
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
N, H, W = 32, 64, 80
with torch.no_grad():
    for kh, kw in ((1, 5), (5, 1), (1, 1)):
        for ci in (16, 64, 256, 1024):
            co = 256
            x = torch.randn(N, ci, H, W, device=dev); w = torch.randn(co, ci, kh, kw, device=dev) * 0.05
            out = torch.empty(N, co, H, W, device=dev); pc = ops.PackedConv(w, None)
            us = t(lambda: ops.conv_fused(x, pc, ops.CONV_LINEAR, out))
            print('%dx%d cin %4d -> 256: %8.1f us  %6.1f TF' % (kh, kw, ci, us, 2.0 * N * H * W * ci * co * kh * kw / us / 1e6))
