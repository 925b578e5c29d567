#!/usr/bin/env python3
"""convc1 (324 -> 256, 1x1, batch 32 at 64x80) on rpe_conv_fused's implicit GEMM vs rpe_conv1x1's LDS-DMA GEMM."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpe_amd import ops
from bench_kernels import timeit
torch.manual_seed(0)
for (b, cin, cout, h, w) in ((32, 324, 256, 64, 80), (48, 128, 256, 64, 80), (2, 324, 256, 64, 80), (32, 256, 576, 64, 80)):
    x = torch.randn(b, cin, h, w, device='cuda'); wt = torch.randn(cout, cin, 1, 1, device='cuda') * 0.05; bias = torch.randn(cout, device='cuda')
    out = torch.empty(b, cout, h, w, device='cuda')
    fl = 2.0 * b * h * w * cin * cout
    for name, fn in (('conv_fused', ops.conv_fused(x, ops.PackedConv(wt, bias), ops.CONV_RELU, out, prepare=True)),
                     ('conv1x1   ', ops.conv1x1(x, ops.PackedConv1x1(wt, bias), ops.CONV_RELU, out, prepare=True))):
        med, mn = timeit(fn, 30)
        print(f'{cin:4d}->{cout:4d} b={b:2d}: {name} {med:8.1f} us (min {mn:.1f})  {fl / med / 1e6:6.1f} TFLOP/s')
