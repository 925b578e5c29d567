#!/usr/bin/env python3
"""Cost of the encoder epilogue / loader options of rpe_conv_fused on fnet's layer1 shape (64 -> 64, 256x320, 48 images)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops
def t(fn, reps=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
dev = torch.device('cuda:0'); torch.manual_seed(0)
N, c, H, W = 48, 64, 256, 320
with torch.no_grad():
    x = torch.randn(N, c, H, W, device=dev); w = torch.randn(c, c, 3, 3, device=dev) * 0.05; bias = torch.randn(c, device=dev)
    out = torch.empty(N, c, H, W, device=dev); res = torch.randn(N, c, H, W, device=dev)
    pc = ops.PackedConv(w, bias); stats = ops.conv_stats_buffer(N, c, H, W, dev)
    scale = torch.rand(c, device=dev) + 0.5
    ops.conv_fused(x, pc, ops.CONV_LINEAR, out, stats=stats); mi = ops.instnorm_finalize(stats, H * W)
    flop = 2.0 * N * H * W * c * c * 9
    for name, kw in (('plain (update-block instantiation)', {}), ('scale (encoder instantiation)', dict(scale=scale)), ('scale + residual', dict(scale=scale, residual=res)),
                     ('stats', dict(stats=stats)), ('stats + pre_norm', dict(stats=stats, pre_norm=mi))):
        us = t(lambda: ops.conv_fused(x, pc, ops.CONV_RELU if 'stats' not in kw else ops.CONV_LINEAR, out, **kw))
        print('%-36s %8.1f us  %6.1f TF' % (name, us, flop / us / 1e6))
