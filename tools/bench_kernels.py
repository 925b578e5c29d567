#!/usr/bin/env python3
"""Micro-benchmarks of the hand-written kernels at BASELINE config-3 geometry (batch 32 pairs / 16 frames,
640x512): HIP-event time and algorithmic GB/s per launch.  Used for kernel iteration and for PMC runs
(rocprofv3 --pmc ... -- python3 tools/bench_kernels.py --only lookup --reps 5)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd  # noqa: E402
from rpe_amd import ops  # noqa: E402


def timeit(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='all')
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--pairs', type=int, default=32)
    ap.add_argument('--flow-std', type=float, default=3.0)
    ap.add_argument('--jitter', type=float, default=0.0, help='per-query white noise on the lookup coordinates (px at 1/8 resolution)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    H, W = 512, 640
    h8, w8 = H // 8, W // 8
    b = a.pairs
    want = lambda k: a.only in ('all', k)
    if want('lookup') or want('build'):
        f1 = torch.randn(b, 256, h8, w8, device=dev)
        f2 = torch.randn(b, 256, h8, w8, device=dev)
        pyr = ops.CorrPyramid(b, h8, w8, device=dev)
        if want('build'):
            med, mn = timeit(lambda: pyr.build(f1, f2), max(3, a.reps // 4))
            fl = 2.0 * b * (h8 * w8) ** 2 * 256
            print(f'corr_build      : {med:9.1f} us (min {mn:.1f})  {fl / med / 1e6:7.1f} TFLOP/s  (+ pyramid write)')
        pyr.build(f1, f2)
        ys, xs = torch.meshgrid(torch.arange(h8, device=dev), torch.arange(w8, device=dev), indexing='ij')
        base = torch.stack((xs, ys)).float()[None].repeat(b, 1, 1, 1)
        # smooth flow field (low-pass noise) + a global shift, like a RAFT iterate
        lo = torch.randn(b, 2, 6, 8, device=dev) * a.flow_std
        coords = base + torch.nn.functional.interpolate(lo, size=(h8, w8), mode='bilinear', align_corners=True) - 2.3
        if a.jitter > 0:
            coords = coords + torch.randn_like(coords) * a.jitter
        out = torch.empty(b, 324, h8, w8, device=dev)
        if want('lookup'):
            med, mn = timeit(lambda: pyr.lookup(coords, out=out), a.reps)
            alg = b * (h8 * w8 * 4 * (100 + 81) * 4 + h8 * w8 * 8)
            print(f'corr_lookup     : {med:9.1f} us (min {mn:.1f})  {alg / med / 1e3:7.1f} GB/s algorithmic ({alg / 1e6:.1f} MB)'
                  f'  = {alg / med / 1e3 / 8000:.3f} of 8 TB/s')
    if want('pose'):
        n = b // 2
        hw = H * W
        flow = torch.randn(n, 2, H, W, device=dev)
        pcl1 = torch.rand(n, 3, H, W, device=dev) + 0.2
        pcl2 = pcl1 + 0.01 * torch.randn(n, 3, H, W, device=dev)
        w1 = torch.rand(n, 1, H, W, device=dev); w2 = torch.rand(n, 1, H, W, device=dev)
        m1 = torch.rand(n, 1, H, W, device=dev) > 0.05; m2 = torch.rand(n, 1, H, W, device=dev) > 0.05
        K = torch.tensor([[704.0, 0, 320], [0, 704.0, 256], [0, 0, 1]], device=dev)[None].repeat(n, 1, 1)
        lw = torch.ones(n, 2, device=dev)
        T = torch.zeros(n, 7, dtype=torch.float64, device=dev); T[:, 6] = 1
        alg = n * hw * 42
        for hess in (False, True):
            med, mn = timeit(lambda: ops.pose_reduce(flow, pcl1, pcl2, w1, w2, m1, m2, K, lw, T, need_hessian=hess), a.reps)
            print(f'pose_reduce H={int(hess)} : {med:9.1f} us (min {mn:.1f})  {alg / med / 1e3:7.1f} GB/s algorithmic ({alg / 1e6:.1f} MB)'
                  f'  = {alg / med / 1e3 / 8000:.3f} of 8 TB/s   [incl. pack kernel + python]')
        for mode, name in ((ops.SOLVER_LBFGS, 'lbfgs'), (ops.SOLVER_GN, 'gn')):
            med, mn = timeit(lambda: ops.pose_solve(flow, pcl1, pcl2, w1, w2, m1, m2, K, lw, iters=8, mode=mode), a.reps)
            print(f'pose_solve {name:5s}: {med:9.1f} us for 8 iterations, n={n}  ({8 * alg / med / 1e3:7.1f} GB/s algorithmic)')
    if want('flowhead'):
        x = torch.randn(b, 256, h8, w8, device=dev); wt = torch.randn(2, 256, 3, 3, device=dev) * 0.05; bias = torch.randn(2, device=dev)
        co = torch.randn(b, 2, h8, w8, device=dev)
        med, mn = timeit(lambda: ops.conv3x3_to2(x, wt, bias, add=co), a.reps)
        alg = b * 256 * h8 * w8 * 4
        print(f'conv3x3_to2     : {med:9.1f} us (min {mn:.1f})  {alg / med / 1e3:7.1f} GB/s (input read once, {alg / 1e6:.1f} MB)')
        med, mn = timeit(lambda: torch.nn.functional.conv2d(x, wt, bias, padding=1), a.reps)
        print(f'  (MIOpen conv2d: {med:9.1f} us)')
    if want('geom'):
        n = b // 2
        sf2 = torch.randn(n, 2, H, W, device=dev); sf2[:, 0] = -20 - 5 * torch.rand(n, H, W, device=dev)
        tf = torch.randn(n, 2, H, W, device=dev) * 4
        base = torch.full((n,), 7.2, device=dev)
        K = torch.tensor([[704.0, 0, 320], [0, 704.0, 256], [0, 0, 1]], device=dev)[None].repeat(n, 1, 1)
        d1 = torch.rand(n, 1, H, W, device=dev); i1 = torch.rand(n, 3, H, W, device=dev); i2 = torch.rand(n, 3, H, W, device=dev)
        sf1 = torch.randn(n, 2, H, W, device=dev); m2 = torch.rand(n, 1, H, W, device=dev) > 0.05
        med, mn = timeit(lambda: ops.depth_backproject_warp(sf2, tf, base, K, d1, i1, i2, sf1, m2), a.reps)
        alg = n * H * W * 94
        print(f'geometry (2 krn): {med:9.1f} us (min {mn:.1f})  {alg / med / 1e3:7.1f} GB/s vs the reference path\'s 94 B/pixel')


if __name__ == '__main__':
    main()
