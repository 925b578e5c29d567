#!/usr/bin/env python3
"""Do two streams overlap at sequential-tracking sizes?  Main stream: the update block's convc2 at RAFT batch 2 (240 workgroups of 256 CUs'
512 slots); side stream: an encoder layer-1 convolution on 2 images (1280 workgroups).  Times: each alone, one after the other on one
stream, side by side on two streams."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops

dev = torch.device('cuda:0'); torch.manual_seed(0)
with torch.no_grad():
    xa = torch.randn(2, 256, 64, 80, device=dev); wa = torch.randn(192, 256, 3, 3, device=dev) * 0.05
    oa = torch.empty(2, 192, 64, 80, device=dev); pa = ops.PackedWino(wa, None)
    fa = ops.conv_wino(xa, pa, ops.CONV_RELU, oa, prepare=True)
    xb = torch.randn(2, 64, 256, 320, device=dev); wb = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    ob = torch.empty(2, 64, 256, 320, device=dev); pb = ops.PackedWino(wb, None)
    side = torch.cuda.Stream()
    NA, NB = 400, 40

    def run(a, b, two):
        torch.cuda.synchronize(); t = time.perf_counter()
        if b and two:
            with torch.cuda.stream(side):
                fb = ops.conv_wino(xb, pb, ops.CONV_RELU, ob, prepare=True)
                for _ in range(NB): fb()
        if a:
            for _ in range(NA): fa()
        if b and not two:
            fb = ops.conv_wino(xb, pb, ops.CONV_RELU, ob, prepare=True)
            for _ in range(NB): fb()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) * 1e3
    for _ in range(2):
        ta, tb, ts, tp = run(True, False, False), run(False, True, False), run(True, True, False), run(True, True, True)
    print('main alone %.2f ms | side alone %.2f ms | one stream %.2f ms | two streams %.2f ms' % (ta, tb, ts, tp))
