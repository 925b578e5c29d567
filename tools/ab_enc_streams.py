#!/usr/bin/env python3
"""fnet (48 images) and cnet (32 images) of a bench step: one after the other on one stream vs concurrently on two."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import pose_net, synth
dev = torch.device('cuda:0'); H, W, B = 512, 640, 16
model = synth.init_synthetic_weights(pose_net.PoseNet(synth.model_config(H, W))).eval().to(dev)
fr = synth.stereo_frames(1000, B, H, W)
i1, i2, i2r = fr['image1l'].to(dev), fr['image2l'].to(dev), fr['image2r'].to(dev)
side = torch.cuda.Stream(); e0, e1 = torch.cuda.Event(), torch.cuda.Event()

def seq():
    f = model.flow.encode_features((i1, i2, i2r)); c = model.flow.encode_context((i1, i2)); return f, c

def par():
    cur = torch.cuda.current_stream(); e0.record(cur)
    with torch.cuda.stream(side):
        side.wait_event(e0); c = model.flow.encode_context((i1, i2)); e1.record(side)
    f = model.flow.encode_features((i1, i2, i2r))
    cur.wait_event(e1); c.record_stream(cur)
    return f, c

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    a, b = seq(), par()
    print('equal:', torch.equal(a[0], b[0]), torch.equal(a[1], b[1]))
    for rep in range(3):
        print(f'sequential {t(seq):.2f} ms   two streams {t(par):.2f} ms')
