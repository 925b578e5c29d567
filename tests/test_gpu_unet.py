"""GPU: the fused weight heads (rpe_unet_heads: both TinyUNets + bilinear resize + sigmoid as one kernel chain) against the
oracle's functional TinyUNet (torch CPU, restating core/unet/unet.py:7-82 + core/pose/pose_net.py:109-115) with the same
seeded parameters and non-trivial batch-norm statistics.  Bar: 1e-5 on the (0,1) weight maps."""
import pytest
import torch

from oracle import unet as ounet

pytestmark = pytest.mark.gpu


def _heads(h, w, seed):
    from rpe_amd import unet
    torch.manual_seed(seed)
    nets, onets = [], []
    for cin in (264, 272):
        n = unet.TinyUNet(cin, (h, w)).eval()
        for m in n.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
        o = ounet.TinyUNet(cin, (h, w)).eval()
        o.load_state_dict(n.state_dict())
        nets.append(n.cuda()); onets.append(o)
    return nets, onets


@pytest.mark.parametrize('h,w,b', [(512, 640, 2), (352, 384, 1), (1024, 1280, 1)])
def test_fused_heads_match_oracle(rpe, h, w, b):
    from rpe_amd import ops, unet
    nets, onets = _heads(h, w, seed=h)
    g = torch.Generator().manual_seed(b + w)
    inp1, inp2 = torch.randn(b, 8, h // 8, w // 8, generator=g), torch.randn(b, 8, h // 8, w // 8, generator=g)
    hc = torch.randn(b, 256, h // 8, w // 8, generator=g)                    # hidden | context as slices of one buffer
    hcd = hc.cuda()
    w2d, w3d = ops.unet_heads(inp1.cuda(), inp2.cuda(), hcd[:, :128], hcd[:, 128:], unet.pack_params(nets[0]), unet.pack_params(nets[1]), (h, w))
    with torch.no_grad():
        r2d = torch.sigmoid(onets[0](torch.cat((inp1, hc[:, :128], hc[:, 128:]), 1)))
        r3d = torch.sigmoid(onets[1](torch.cat((inp1, inp2, hc[:, :128], hc[:, 128:]), 1)))
    assert w2d.shape == r2d.shape == (b, 1, h, w)
    d2, d3 = float((w2d.cpu() - r2d).abs().max()), float((w3d.cpu() - r3d).abs().max())
    print(f'{w}x{h}: fused heads vs oracle {d2:.2e} / {d3:.2e}')
    assert d2 < 1e-5 and d3 < 1e-5
    assert float(w2d.min()) > 0.0 and float(w2d.max()) < 1.0
    # the PyTorch-ROCm route of the same module (library convolutions + HIP epilogues) agrees too
    with torch.no_grad():
        t2d = torch.sigmoid(nets[0](torch.cat((inp1.cuda(), hcd[:, :128], hcd[:, 128:]), 1)))
    assert float((w2d - t2d).abs().max()) < 1e-5


def test_fused_heads_reject_small_grids(rpe):
    from rpe_amd import ops, unet
    nets, _ = _heads(256, 320, seed=1)
    z = lambda c: torch.zeros(1, c, 32, 40, device='cuda')
    with pytest.raises(rpe.RpeError, match='too small'):
        ops.unet_heads(z(8), z(8), z(128), z(128), unet.pack_params(nets[0]), unet.pack_params(nets[1]), (256, 320))


def test_fused_heads_do_not_depend_on_the_batch(rpe):
    """At 640x512 a batch of 16 runs the valid 3x3 layers on 16-channel threads, a batch of 1 or 8 on 4-channel threads (launch fill):
    same explicit fused multiply-adds in the same order, so a frame's weight maps are the same bits in any batch (the chunked sequence
    tracker relies on it: 16-frame chunks against the reference's frame-at-a-time walk)."""
    from rpe_amd import ops, unet
    torch.manual_seed(0)
    H, W = 512, 640
    h8, w8, b = H // 8, W // 8, 16
    n2, n3 = unet.TinyUNet(264, (H, W)).cuda().eval(), unet.TinyUNet(272, (H, W)).cuda().eval()
    p2, p3 = unet.pack_params(n2), unet.pack_params(n3)
    i1, i2 = torch.randn(b, 8, h8, w8, device='cuda'), torch.randn(b, 8, h8, w8, device='cuda')
    hid, ctx = torch.randn(b, 128, h8, w8, device='cuda'), torch.randn(b, 128, h8, w8, device='cuda')
    full = ops.unet_heads(i1, i2, hid, ctx, p2, p3, (H, W))
    for sl in (slice(0, 1), slice(5, 13)):
        part = ops.unet_heads(i1[sl].contiguous(), i2[sl].contiguous(), hid[sl].contiguous(), ctx[sl].contiguous(), p2, p3, (H, W))
        assert torch.equal(part[0], full[0][sl]) and torch.equal(part[1], full[1][sl])
