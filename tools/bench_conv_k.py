#!/usr/bin/env python3
"""k_conv_igemm rate vs reduction length (cin) at fixed output size: separates per-step from per-workgroup costs.
usage: bench_conv_k.py [cout]"""
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
co = int(sys.argv[1]) if len(sys.argv) > 1 else 256
with torch.no_grad():
    for kh, kw in ((1, 5), (5, 1), (3, 3), (1, 1)):
        prev = None
        for ci in (16, 64, 256, 1024):
            x = torch.randn(N, ci, H, W, device=dev); w = torch.randn(co, ci, kh, kw, device=dev) * 0.05
            out = torch.empty(N, co, H, W, device=dev); pc = ops.PackedConv(w, None)
            us = t(lambda: ops.conv_fused(x, pc, ops.CONV_LINEAR, out))
            steps = (ci // 16) * kh * kw
            slope = '' if prev is None else '  slope %.2f us/step (ideal %.2f)' % ((us - prev[0]) / (steps - prev[1]), 2.0 * N * H * W * 16 * co / 157.3e6)
            print('%dx%d cin %4d -> %d: %8.1f us  %6.1f TF%s' % (kh, kw, ci, co, us, 2.0 * N * H * W * ci * co * kh * kw / us / 1e6, slope))
            prev = (us, steps)
