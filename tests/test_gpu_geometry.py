"""GPU: fused depth / back-projection / warp / 1-8 stacks kernel against the oracle (the reference's own
torch calls) -- integer tap indices and masks bit-exact, floats to 1e-5 relative."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import synth, warp

pytestmark = pytest.mark.gpu


def scene(seed, n, h, w):
    rng = np.random.default_rng(seed)
    t = lambda a: torch.from_numpy(a.astype(np.float32))
    K = synth.intrinsics(h, w)[None].repeat(n, 1, 1)
    sflow2 = t(rng.normal(0, 1.0, size=(n, 2, h, w)))
    sflow2[:, 0] = -t(rng.uniform(-2.0, 40.0, size=(n, h, w)))            # some invalid (positive) disparities
    sflow2[:, 0, :4, :4] = 0.0                                              # division by zero -> inf -> invalid
    tflow = t(rng.normal(0, 6.0, size=(n, 2, h, w)))
    tflow[:, :, : h // 3] = torch.round(tflow[:, :, : h // 3] * 2) / 2      # many exact .0 / .5 positions
    tflow[0, :, 5, 5] = torch.tensor([1e9, -1e9])
    baseline = t(rng.uniform(4.0, 8.0, size=(n,)))
    depth1 = t(rng.uniform(0.1, 1.0, size=(n, 1, h, w)))
    img1, img2 = t(rng.uniform(0, 255, size=(n, 3, h, w))), t(rng.uniform(0, 255, size=(n, 3, h, w)))
    sflow1 = t(rng.normal(0, 5.0, size=(n, 2, h, w)))
    mask2 = torch.from_numpy(rng.uniform(size=(n, 1, h, w)) > 0.2)
    return sflow2, tflow, baseline, K, depth1, img1, img2, sflow1, mask2


@pytest.mark.parametrize('n,h,w', [(2, 64, 96), (1, 256, 320), (2, 512, 640)])
def test_fused_geometry_matches_oracle(rpe, n, h, w):
    from rpe_amd import ops
    sflow2, tflow, baseline, K, depth1, img1, img2, sflow1, mask2 = scene(n * 100 + h, n, h, w)
    out = ops.depth_backproject_warp(*[a.cuda() for a in (sflow2, tflow, baseline, K, depth1, img1, img2, sflow1, mask2)],
                                     want_pcl2=True)
    depth2, valid = warp.flow2depth(sflow2, baseline)
    m2 = mask2 & valid
    pcl1, pcl2 = warp.backproject(depth1, K), warp.backproject(depth2, K)
    pcl2w, mask2w, inp1, inp2 = warp.weight_inputs(pcl1, pcl2, img1, img2, m2, tflow, sflow1, sflow2)
    assert torch.equal(out['depth2'].cpu(), depth2)                         # IEEE division, bit-exact
    assert torch.equal(out['mask2'].cpu(), m2)
    assert torch.equal(out['mask2w'].cpu(), mask2w)                         # nearest-warp indices bit-exact
    close = lambda a, b, tol: float((a.cpu() - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
    assert close(out['pcl1'], pcl1, 1e-6)
    assert close(out['pcl2'], pcl2, 1e-6)
    assert close(out['pcl2w'], pcl2w, 1e-5)
    assert close(out['inp1'], inp1, 1e-6)
    assert close(out['inp2'], inp2, 1e-5)


def test_warp_taps_bit_exact(rpe):
    from rpe_amd import ops
    g = load_golden('warp.npz')
    for flow in (g['flow'], scene(5, 2, 128, 160)[1]):
        taps = ops.warp_taps(flow.cuda())
        ref = warp.sample_taps(flow)
        sane = np.abs(ref['ix']) < 1e6
        for k in ('x0', 'y0', 'xn', 'yn'):
            a = taps[k].cpu().numpy()
            ok = sane & (np.abs(ref['iy']) < 1e6)
            assert np.array_equal(a[ok], ref[k][ok].astype(np.int32)), k


def test_reference_golden_nearest_warp(rpe):
    """remap_from_flow_nearest output of the reference file itself (tests/golden/warp.npz)."""
    from rpe_amd import ops
    g = load_golden('warp.npz')
    flow, mask = g['flow'], g['mask']
    n, _, h, w = flow.shape
    # feed the mask through the fused kernel with an always-valid disparity
    sflow2 = torch.zeros(n, 2, h, w)
    sflow2[:, 0] = -10.0
    base = torch.full((n,), 5.0)
    K = synth.intrinsics(h, w)[None].repeat(n, 1, 1)
    z3, z2, z1 = torch.zeros(n, 3, h, w), torch.zeros(n, 2, h, w), torch.ones(n, 1, h, w)
    out = ops.depth_backproject_warp(*[a.cuda() for a in (sflow2, flow, base, K, z1, z3, z3, z2, mask)])
    expect = g['nearest_valid'] & g['nearest'].bool()
    assert torch.equal(out['mask2w'].cpu(), expect)


def test_flow2depth(rpe):
    from rpe_amd import ops
    sflow2, _, baseline, *_ = scene(3, 2, 64, 80)
    d, v = ops.flow2depth(sflow2.cuda(), baseline.cuda())
    dr, vr = warp.flow2depth(sflow2, baseline)
    assert torch.equal(d.cpu(), dr) and torch.equal(v.cpu(), vr)
