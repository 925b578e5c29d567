#!/usr/bin/env python3
"""Timing of rpe_stem_conv at bench geometry: the encoders' 7x7 stride-2 stems (fnet: 48 images, instance-norm moments; cnet: 16 images, folded
batch norm + ReLU) and the motion encoder's convf1 (7x7 stride 1 on the 2 flow channels of 32 pairs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops
from bench_kernels import timeit
dev = torch.device('cuda:0'); torch.manual_seed(0)
reps = int(os.environ.get('REPS', 20))
w3 = torch.randn(64, 3, 7, 7, device=dev) * 0.1; ps3 = ops.PackedStem(w3); b64 = torch.randn(64, device=dev); s64 = torch.rand(64, device=dev) + 0.5
for name, nb, kw in (('fnet stem (48 x 3 x 512 x 640, moments)', 48, dict(bias=b64, relu=False, stats=True)),
                     ('cnet stem (16 x 3 x 512 x 640, scale + bias + ReLU)', 16, dict(bias=b64, scale=s64, relu=True))):
    img = torch.rand(nb, 3, 512, 640, device=dev) * 255
    out = torch.empty(nb, 64, 256, 320, device=dev)
    if kw.get('stats'):
        kw['stats'] = torch.empty(nb, 64, rpe_amd._lib.lib().rpe_stem_tiles(512, 640, 2), 3, device=dev)
    med, mn = timeit(lambda: ops.stem_conv(img, ps3, out=out, **kw), reps)
    fl = 2.0 * nb * 256 * 320 * 147 * 64
    print(f'{name}: {med:8.1f} us (min {mn:.1f})  {fl / med / 1e6:6.1f} TFLOP/s useful, {fl * 160 / 147 / med / 1e6:6.1f} executed;  output {out.numel() * 4 / med / 1e3:6.0f} GB/s')
w2 = torch.randn(128, 2, 7, 7, device=dev) * 0.1; ps2 = ops.PackedStem(w2); b128 = torch.randn(128, device=dev)
for nb in (32, 2):
    flow = torch.randn(nb, 2, 64, 80, device=dev); out = torch.empty(nb, 128, 64, 80, device=dev)
    med, mn = timeit(lambda: ops.stem_conv(flow, ps2, bias=b128, relu=True, div=1.0, mul=1.0, sub=0.0, out=out), reps)
    fl = 2.0 * nb * 5120 * 98 * 128
    print(f'convf1 ({nb} x 2 x 64 x 80 -> 128): {med:8.1f} us (min {mn:.1f})  {fl / med / 1e6:6.1f} TFLOP/s useful, {fl * 112 / 98 / med / 1e6:6.1f} executed')
