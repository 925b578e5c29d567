#!/usr/bin/env python3
"""The encoders' stride-2 layers at bench geometry (fnet: 48 images, moments): 3x3 stride 2 and the 1x1 stride-2 shortcut."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops
from bench_kernels import timeit
dev = torch.device('cuda:0'); torch.manual_seed(0)
for name, nb, cin, cout, k, h, w in (('layer2.conv1 3x3 s2', 48, 64, 96, 3, 256, 320), ('layer2.shortcut 1x1 s2', 48, 64, 96, 1, 256, 320),
                                     ('layer3.conv1 3x3 s2', 48, 96, 128, 3, 128, 160), ('layer3.shortcut 1x1 s2', 48, 96, 128, 1, 128, 160)):
    x = torch.randn(nb, cin, h, w, device=dev); wt = torch.randn(cout, cin, k, k, device=dev) * 0.05; bias = torch.randn(cout, device=dev)
    out = torch.empty(nb, cout, h // 2, w // 2, device=dev)
    pc = ops.PackedConv(wt, bias)
    st = ops.conv_stats_buffer(nb, cout, h, w, dev, stride=2)
    med, mn = timeit(lambda: ops.conv_fused(x, pc, ops.CONV_LINEAR, out, stats=st, stride=2), 12)
    fl = 2.0 * nb * (h // 2) * (w // 2) * cin * cout * k * k
    gb = (x.numel() * (0.5 if k == 1 else 1.0) + out.numel()) * 4 / 1e9
    print(f'{name:24s} x{nb}: {med:8.1f} us (min {mn:.1f})  {fl / med / 1e6:6.1f} TFLOP/s  {gb / med * 1e6 / 1e3:5.2f} TB/s of input rows + output')
