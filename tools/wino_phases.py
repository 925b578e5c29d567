import os, sys, ctypes, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import rpe_amd
from rpe_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
def phases(tag):
    buf = (ctypes.c_ulonglong * 8)(); rpe_amd._lib.lib().rpe_debug_wino_timing(buf)
    total = buf[0] + buf[1] + buf[2]
    clock = ('  shader clock over the workgroup: %.0f MHz' % (total / buf[7] * 100.0)) if buf[7] else ''
    print('%-28s prologue %6d (set-up %5d, first loads %5d, first transform %5d)  loop %7d  epilogue %6d cycles (one wave of a mid-grid workgroup)%s' % (tag, buf[0], buf[3], buf[4], buf[5], buf[1], buf[2], clock))
for name, c, hh, ww, nb in (('layer1 64ch 256x320 x48', 64, 256, 320, 48), ('layer2 96ch 128x160 x48', 96, 128, 160, 48), ('layer3 128ch 64x80 x48', 128, 64, 80, 48)):
    x = torch.randn(nb, c, hh, ww, device=dev); w = torch.randn(c, c, 3, 3, device=dev) * 0.05; bias = torch.randn(c, device=dev)
    o = torch.empty(nb, c, hh, ww, device=dev); pw = ops.PackedWino(w, None)
    st = ops.conv_wino_stats_buffer(nb, c, hh, ww, dev)
    for _ in range(3): ops.conv_wino(x, pw, ops.CONV_LINEAR, o, bias=bias, stats=st)
    torch.cuda.synchronize(); phases(name + ' +moments')
    for _ in range(3): ops.conv_wino(x, pw, ops.CONV_RELU, o, bias=bias)
    torch.cuda.synchronize(); phases(name + ' plain')
    ones = torch.ones(c, device=dev)
    for _ in range(3): ops.conv_wino(x, pw, ops.CONV_RELU, o, bias=bias, scale=ones)
    torch.cuda.synchronize(); phases(name + ' scale')
    for _ in range(3): ops.conv_wino(x, pw, ops.CONV_RELU, o, bias=bias, scale=ones, residual=x)
    torch.cuda.synchronize(); phases(name + ' scale+res')
x = torch.randn(32, 256, 64, 80, device=dev); w = torch.randn(192, 256, 3, 3, device=dev) * 0.05
o = torch.empty(32, 192, 64, 80, device=dev); pw = ops.PackedWino(w, torch.randn(192, device=dev))
for _ in range(3): ops.conv_wino(x, pw, ops.CONV_RELU, o)
torch.cuda.synchronize(); phases('convc2 256->192 x32')
