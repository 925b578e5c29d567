#!/usr/bin/env python3
"""Conv-library experiments: encoder and update-block convolutions in NCHW vs channels_last."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import raft as R

def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

dev = torch.device('cuda:0')
torch.manual_seed(0)
with torch.no_grad():
    enc = R.BasicEncoder(256, 'instance').eval().to(dev)
    encb = R.BasicEncoder(256, 'batch').eval().to(dev)
    x = torch.randn(32, 3, 512, 640, device=dev)
    print('fnet  NCHW 32 img: %.2f ms' % t(lambda: enc(x)))
    print('cnet  NCHW 32 img: %.2f ms' % t(lambda: encb(x)))
    enc_cl = enc.to(memory_format=torch.channels_last); xcl = x.contiguous(memory_format=torch.channels_last)
    print('fnet  CL   32 img: %.2f ms' % t(lambda: enc_cl(xcl)))
    encb_cl = encb.to(memory_format=torch.channels_last)
    print('cnet  CL   32 img: %.2f ms' % t(lambda: encb_cl(xcl)))
    ref = enc_cl(xcl); 
    ub = R.BasicUpdateBlock().eval().to(dev)
    N = 32
    corr = torch.randn(N, 324, 64, 80, device=dev); flow = torch.randn(N, 2, 64, 80, device=dev)
    hx = torch.randn(N, 384, 64, 80, device=dev); h = torch.randn(N, 128, 64, 80, device=dev)
    print('motion enc NCHW: %.2f ms' % t(lambda: ub.encoder(flow, corr)))
    print('flow head  NCHW: %.2f ms' % t(lambda: ub.flow_head(h)))
    g = ub.gru
    w1 = torch.cat((g.convz1.weight, g.convr1.weight), 0).detach(); b1 = torch.cat((g.convz1.bias, g.convr1.bias), 0).detach()
    w2 = torch.cat((g.convz2.weight, g.convr2.weight), 0).detach(); b2 = torch.cat((g.convz2.bias, g.convr2.bias), 0).detach()
    F = torch.nn.functional
    print('gru zr1 (1x5, 384->256) NCHW: %.2f ms' % t(lambda: F.conv2d(hx, w1, b1, padding=(0, 2))))
    print('gru q1  (1x5, 384->128) NCHW: %.2f ms' % t(lambda: ub.gru.convq1(hx)))
    print('gru zr2 (5x1, 384->256) NCHW: %.2f ms' % t(lambda: F.conv2d(hx, w2, b2, padding=(2, 0))))
    print('gru q2  (5x1, 384->128) NCHW: %.2f ms' % t(lambda: ub.gru.convq2(hx)))
    for name, conv, inp in (('convc1 1x1 324->256', ub.encoder.convc1, corr), ('convc2 3x3 256->192', ub.encoder.convc2, torch.randn(N,256,64,80,device=dev)),
                            ('convf1 7x7 2->128', ub.encoder.convf1, flow), ('convf2 3x3 128->64', ub.encoder.convf2, torch.randn(N,128,64,80,device=dev)),
                            ('conv 3x3 256->126', ub.encoder.conv, torch.randn(N,256,64,80,device=dev)), ('fh1 3x3 128->256', ub.flow_head.conv1, h),
                            ('fh2 3x3 256->2', ub.flow_head.conv2, torch.randn(N,256,64,80,device=dev))):
        print('  %-22s NCHW: %.3f ms' % (name, t(lambda: conv(inp))), end='')
        ccl = conv.to(memory_format=torch.channels_last); icl = inp.contiguous(memory_format=torch.channels_last)
        print('   CL: %.3f ms' % t(lambda: ccl(icl)))
    ubcl = ub.to(memory_format=torch.channels_last)
    hxcl = hx.contiguous(memory_format=torch.channels_last)
    w1c = w1.contiguous(memory_format=torch.channels_last); w2c = w2.contiguous(memory_format=torch.channels_last)
    print('gru zr1 CL: %.2f ms' % t(lambda: F.conv2d(hxcl, w1c, b1, padding=(0, 2))))
    print('gru zr2 CL: %.2f ms' % t(lambda: F.conv2d(hxcl, w2c, b2, padding=(2, 0))))
    print('gru q1  CL: %.2f ms' % t(lambda: ubcl.gru.convq1(hxcl)))
