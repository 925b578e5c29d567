#!/usr/bin/env python3
"""Does MIOpen run the update-block / encoder convolutions faster on NHWC tensors when PyTorch hands them over as
NHWC (PYTORCH_MIOPEN_SUGGEST_NHWC=1 + channels_last), i.e. without its own NCHW<->NHWC transposes?"""
import os, sys
os.environ.setdefault('PYTORCH_MIOPEN_SUGGEST_NHWC', sys.argv[1] if len(sys.argv) > 1 else '1')
import torch
F = torch.nn.functional

def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

dev = torch.device('cuda:0')
torch.manual_seed(0)
N = 32
cases = [  # name, cin, cout, kh, kw, H, W, stride, batch
    ('gru zr 1x5 256->256', 256, 256, 1, 5, 64, 80, 1, N), ('gru q 1x5 256->128', 256, 128, 1, 5, 64, 80, 1, N),
    ('gru zr 5x1 256->256', 256, 256, 5, 1, 64, 80, 1, N), ('gru q 5x1 256->128', 256, 128, 5, 1, 64, 80, 1, N),
    ('convc1 1x1 324->256', 324, 256, 1, 1, 64, 80, 1, N), ('convc2 3x3 256->192', 256, 192, 3, 3, 64, 80, 1, N),
    ('convf2 3x3 128->64', 128, 64, 3, 3, 64, 80, 1, N), ('conv 3x3 256->126', 256, 126, 3, 3, 64, 80, 1, N),
    ('fh1 3x3 128->256', 128, 256, 3, 3, 64, 80, 1, N),
    ('enc l1 3x3 64->64 @256x320 x48', 64, 64, 3, 3, 256, 320, 1, 48),
    ('enc l2 3x3 96->96 @128x160 x48', 96, 96, 3, 3, 128, 160, 1, 48),
    ('enc l3 3x3 128->128 @64x80 x48', 128, 128, 3, 3, 64, 80, 1, 48),
    ('enc l2a 3x3 64->96 s2 x48', 64, 96, 3, 3, 256, 320, 2, 48),
]
with torch.no_grad():
    for name, ci, co, kh, kw, H, W, s, b in cases:
        x = torch.randn(b, ci, H, W, device=dev); w = torch.randn(co, ci, kh, kw, device=dev) * 0.05
        pad = (kh // 2, kw // 2)
        flop = 2.0 * b * (H // s) * (W // s) * ci * co * kh * kw
        t0 = t(lambda: F.conv2d(x, w, None, stride=s, padding=pad))
        xc = x.contiguous(memory_format=torch.channels_last); wc = w.contiguous(memory_format=torch.channels_last)
        t1 = t(lambda: F.conv2d(xc, wc, None, stride=s, padding=pad))
        y0 = F.conv2d(x, w, None, stride=s, padding=pad); y1 = F.conv2d(xc, wc, None, stride=s, padding=pad)
        print('%-34s NCHW %8.1f us (%6.1f TF)   NHWC %8.1f us (%6.1f TF)  out_cl=%s  maxdiff %.2e' % (
            name, t0, flop / t0 / 1e6, t1, flop / t1 / 1e6, y1.is_contiguous(memory_format=torch.channels_last), (y0 - y1).abs().max().item()))
