"""The GRU's q launch (1x5, 128 output channels) at 2 pairs of 640x512 -- the small-launch shape of k_conv_wino1d (CB = 1), 60 launches: for
tools/pmc_kernel.sh (what does a lone workgroup per CU wait for?)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
N, c, H, W = int(os.environ.get('GRU_N', 2)), 128, 64, 80
with torch.no_grad():
    hx = torch.randn(N, 2 * c, H, W, device=dev) * 0.05; rhx = hx.clone()
    z_buf = torch.rand(N, c, H, W, device=dev); ctx_q = torch.randn(N, c, H, W, device=dev) * 0.05
    wq = torch.randn(c, 2 * c, 1, 5, device=dev) * 0.02
    fq = ops.conv_wino1d(rhx, ops.PackedWino1d(wq), ops.CONV_GATE_H, hx[:, :c], add=ctx_q, hidden=hx[:, :c], zgate=z_buf, prepare=True)
    for _ in range(60): fq()
    torch.cuda.synchronize()
print('done')
