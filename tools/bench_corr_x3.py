#!/usr/bin/env python3
"""Correlation build: f32 matrix pipe vs the six-product bf16 split (RPE_F32X3) -- time at the bench geometry and error of level 0
against an f64 evaluation (one pair, a block of queries)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd  # noqa: F401
from rpe_amd import ops

dev = torch.device('cuda:0'); torch.manual_seed(0)


def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for kind in ('gaussian', 'relu-like, heavy tails'):
    b, h8, w8 = 32, 64, 80
    f1 = torch.randn(b, 256, h8, w8, device=dev); f2 = torch.randn(b, 256, h8, w8, device=dev)
    if kind != 'gaussian':
        f1 = torch.relu(f1 + 1.0) * torch.exp(0.8 * torch.randn(b, 256, 1, 1, device=dev)); f2 = torch.relu(f2 + 1.0) * torch.exp(0.8 * torch.randn(b, 256, 1, 1, device=dev))
    pyr = ops.CorrPyramid(b, h8, w8, device=dev, bf16x3=True)
    ref = torch.einsum('cq,cp->qp', f1[3].double().reshape(256, -1), f2[3].double().reshape(256, -1)) / 16.0
    scale = float(ref.abs().max())
    for name, kw in (('f32', {}), ('bf16x3', dict(bf16x3=True))):
        t = timeit(lambda: pyr.build(f1, f2, **kw))
        pyr.build(f1, f2, **kw)
        nq = h8 * w8
        got = pyr.export_level(0)[3 * nq:4 * nq].double().reshape(nq, nq)
        err = (got - ref)
        lv3 = pyr.export_level(3)
        print(f'{kind:24s} {name:7s} {t:8.1f} us   level-0 error vs f64: rms {float(err.pow(2).mean().sqrt()):.3e} max {float(err.abs().max()):.3e}  (|corr| up to {scale:.1f})'
              f'   level 3 checksum {float(lv3.double().sum()):.6f}')
