#!/usr/bin/env python3
"""convc1 on rpe_conv1x1 only (for PMC passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpe_amd import ops
torch.manual_seed(0)
b, cin, cout, h, w = 32, 324, 256, 64, 80
x = torch.randn(b, cin, h, w, device='cuda'); wt = torch.randn(cout, cin, 1, 1, device='cuda') * 0.05; bias = torch.randn(cout, device='cuda')
out = torch.empty(b, cout, h, w, device='cuda')
fn = ops.conv1x1(x, ops.PackedConv1x1(wt, bias), ops.CONV_RELU, out, prepare=True)
for _ in range(5):
    fn()
torch.cuda.synchronize()
