#!/usr/bin/env python3
"""rpe_corr_lookup_conv1x1 against rpe_corr_lookup + convc1 (the route each batch takes), isolated: batch 32 (the bench step) and batch 2 (a
tracker frame), 640x512, median of 50 launches by HIP events after a warm-up."""
import os, sys, statistics, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops
dev = torch.device('cuda:0'); h8, w8 = 64, 80
def med(fn, n=50):
    for _ in range(30): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev) * 1e3
for b in (32, 16, 8, 4, 2, 1):
    g = torch.Generator(device='cpu').manual_seed(1)
    f1, f2 = torch.randn(b, 256, h8, w8, generator=g).to(dev), torch.randn(b, 256, h8, w8, generator=g).to(dev)
    pyr = ops.CorrPyramid(b, h8, w8, device=dev).build(f1, f2)
    ys, xs = torch.meshgrid(torch.arange(h8), torch.arange(w8), indexing='ij')
    co = (torch.stack((xs, ys)).float()[None].repeat(b, 1, 1, 1) + 0.37 + torch.randn(b, 2, h8, w8, generator=g) * 0.05).to(dev)
    wt, bias = (torch.randn(256, 324, 1, 1, generator=g) * 0.05).to(dev), torch.randn(256, generator=g).to(dev)
    corr, out = torch.empty(b, 324, h8, w8, device=dev), torch.empty(b, 256, h8, w8, device=dev)
    c1 = ops.Conv1x1(wt, bias)(corr, ops.CONV_RELU, out, prepare=True)
    lk = pyr.lookup(co, out=corr, prepare=True)
    fz = pyr.lookup_conv1x1(co, ops.PackedLookupConv(wt, bias), torch.empty_like(out), prepare=True)
    t_lk, t_c1 = med(lk), med(c1)
    t_two = med(lambda: (lk(), c1()))
    t_fz = med(fz)
    lk(); c1(); ref = out.clone(); got = fz()
    print(f'batch {b}: lookup {t_lk:.1f} us + convc1 {t_c1:.1f} us; back to back {t_two:.1f} us; fused {t_fz:.1f} us; equal {torch.equal(ref, got)}')
