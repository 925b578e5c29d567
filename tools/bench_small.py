#!/usr/bin/env python3
"""Timing of the step's small HBM-bound kernels at bench geometry: the flow head's output layer (rpe_conv3x3_to2_flow), instance-norm apply."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops
from bench_kernels import timeit
dev = torch.device('cuda:0'); torch.manual_seed(0)
b, h8, w8 = 32, 64, 80
x = torch.randn(b, 256, h8, w8, device=dev); w = torch.randn(2, 256, 3, 3, device=dev) * 0.05; bias = torch.randn(2, device=dev)
coords = torch.randn(b, 2, h8, w8, device=dev); flow = torch.empty_like(coords); hx = torch.empty(b, 256, h8, w8, device=dev); rhx = torch.empty_like(hx)
fn = ops.flow_update(x, w, bias, coords, coords, flow_out=flow, dst1=hx[:, 254:], dst2=rhx[:, 254:], prepare=True)
med, mn = timeit(fn, 30)
print(f'flow_update (256 -> 2, 3x3, batch 32): {med:7.1f} us (min {mn:.1f})  {x.numel() * 4 / med / 1e3:7.1f} GB/s of input')
nb, c, hh, ww = 48, 64, 256, 320
raw = torch.randn(nb, c, hh, ww, device=dev); res = torch.randn_like(raw); out = torch.empty_like(raw)
wt = torch.randn(c, c, 3, 3, device=dev) * 0.05
st = ops.conv_wino_stats_buffer(nb, c, hh, ww, dev)
ops.conv_wino(raw, ops.PackedWino(wt, None), ops.CONV_LINEAR, out, bias=torch.zeros(c, device=dev), stats=st)
med, mn = timeit(lambda: ops.instnorm_apply(raw, st, relu=True, residual=res, out=out), 10)
print(f'instnorm_apply (48 x 64 x 256 x 320, residual): {med:7.1f} us (min {mn:.1f})  {3 * raw.numel() * 4 / med / 1e3:7.1f} GB/s')
mi = ops.instnorm_finalize(st, hh * ww, channels=c)
med, mn = timeit(lambda: ops.instnorm_apply(raw, mi, relu=True, residual=res, out=out), 10)
print(f'instnorm_apply with the moments given (finalize first; 16 workgroups per plane): {med:7.1f} us (min {mn:.1f})  {3 * raw.numel() * 4 / med / 1e3:7.1f} GB/s')
med, mn = timeit(lambda: ops.instnorm_finalize(st, hh * ww, channels=c), 20)
print(f'instnorm_finalize (tile-major records of 48 x 64 x 256 x 320): {med:7.1f} us (min {mn:.1f})')
