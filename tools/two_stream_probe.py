#!/usr/bin/env python3
"""Does running the GRU's convolution sequence for two half batches on two streams fill the partial last rounds of each launch?
(q convolutions are 1280 workgroups = 2.5 rounds of the chip's 512 slots at batch 32.)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops

dev = torch.device('cuda:0'); torch.manual_seed(0)
H, W, c = 64, 80, 128


def make(N):
    hx = torch.randn(N, 256, H, W, device=dev) * 0.5; rhx = hx.clone(); z = torch.rand(N, c, H, W, device=dev)
    seq = []
    for kh, kw in ((1, 5), (5, 1)):
        wzr = torch.randn(256, 256, kh, kw, device=dev) * 0.03; wq = torch.randn(128, 256, kh, kw, device=dev) * 0.03
        azr = torch.randn(N, 256, H, W, device=dev) * 0.3; aq = torch.randn(N, 128, H, W, device=dev) * 0.3
        seq.append(ops.conv_wino1d(hx, ops.PackedWino1d(wzr), ops.CONV_GATE_ZR, z, out2=rhx[:, :c], add=azr, hidden=hx[:, :c], gate_channels=c, prepare=True))
        seq.append(ops.conv_wino1d(rhx, ops.PackedWino1d(wq), ops.CONV_GATE_H, hx[:, :c], add=aq, hidden=hx[:, :c], zgate=z, prepare=True))
    w3 = torch.randn(192, 256, 3, 3, device=dev) * 0.05; o3 = torch.empty(N, 192, H, W, device=dev)
    seq.append(ops.conv_wino(hx, ops.PackedWino(w3, torch.randn(192, device=dev)), ops.CONV_RELU, o3, prepare=True))
    w4 = torch.randn(64, 128, 3, 3, device=dev) * 0.05; o4 = torch.empty(N, 64, H, W, device=dev)
    seq.append(ops.conv_wino(hx[:, :128], ops.PackedWino(w4, torch.randn(64, device=dev)), ops.CONV_RELU, o4, prepare=True))
    return seq


def run(seqs, streams, reps):
    for s in streams: s.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for seq, st in zip(seqs, streams):
            with torch.cuda.stream(st):
                for f in seq: f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


one = make(32)
halves = [make(16), make(16)]
s0 = torch.cuda.current_stream(); sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(2):
    print('one stream, batch 32          : %.3f ms per sequence' % run([one], [s0], 20))
    print('one stream, 2 x batch 16      : %.3f ms' % run(halves, [s0, s0], 20))
    print('two streams, 2 x batch 16     : %.3f ms' % run(halves, [sa, sb], 20))
