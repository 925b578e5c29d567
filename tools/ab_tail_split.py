#!/usr/bin/env python3
"""Experiment: launches that are a fractional number of rounds of the chip (q convolutions of the GRU: 1280 workgroups = 2.5 rounds of the
512 resident slots; convf2 2.5, convc2 7.5) -- does giving the LAST batch rows to 32-channel workgroups (the small-launch tile class, same bits)
on a second stream shorten the tail?  Times the whole launch against (rows [0, b1) on the main stream | rows [b1, b) on a side stream, fork /
join by events) for several b1."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops

dev = torch.device('cuda:0'); torch.manual_seed(0)
N, H, W, c = 32, 64, 80, 128
side = torch.cuda.Stream()


def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def forked(f1, f2):
    ev1, ev2 = torch.cuda.Event(), torch.cuda.Event()

    def run():
        ev1.record()
        with torch.cuda.stream(side):
            side.wait_event(ev1)
            f2()
            ev2.record()
        f1()
        torch.cuda.current_stream().wait_event(ev2)
    return run


with torch.no_grad():
    for name, kh, kw in (('q 1x5', 1, 5), ('q 5x1', 5, 1)):
        rhx = torch.randn(N, 256, H, W, device=dev) * 0.5; hx = rhx.clone(); z = torch.rand(N, c, H, W, device=dev)
        w = torch.randn(c, 256, kh, kw, device=dev) * 0.03; add = torch.randn(N, c, H, W, device=dev) * 0.3
        pw = ops.PackedWino1d(w)
        ho = torch.empty(N, c, H, W, device=dev)
        mk = lambda a, b: ops.conv_wino1d(rhx[a:b], pw, ops.CONV_GATE_H, ho[a:b], add=add[a:b], hidden=hx[a:b, :c], zgate=z[a:b], prepare=True)
        whole = mk(0, N)
        print('%s  whole launch %7.1f us' % (name, t(whole)))
        for b1 in (24, 25, 26, 27, 28):
            print('    rows [0,%d) | [%d,%d) on two streams: %7.1f us   (serial, one stream: %7.1f)' % (
                b1, b1, N, t(forked(mk(0, b1), mk(b1, N))), t(lambda f1=mk(0, b1), f2=mk(b1, N): (f1(), f2()))))
    for name, ci, co in (('convf2', 128, 64), ('convc2', 256, 192)):
        x = torch.randn(N, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05; bias = torch.randn(co, device=dev)
        o = torch.empty(N, co, H, W, device=dev)
        pw = ops.PackedWino(w, bias)
        mk = lambda a, b: ops.conv_wino(x[a:b], pw, ops.CONV_RELU, o[a:b], prepare=True)
        print('%s  whole launch %7.1f us' % (name, t(mk(0, N))))
        for b1 in (24, 26, 28, 30):
            print('    rows [0,%d) | [%d,%d) on two streams: %7.1f us   (serial: %7.1f)' % (
                b1, b1, N, t(forked(mk(0, b1), mk(b1, N))), t(lambda f1=mk(0, b1), f2=mk(b1, N): (f1(), f2()))))
