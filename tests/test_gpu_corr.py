"""GPU: correlation pyramid build + lookup, GRU gate kernels and convex up-sampling against the oracle's
RAFT restatement (torch CPU).  Float tolerance 2e-5 relative to the correlation scale (f32 dot products of
length 256 accumulated in a different order); integer taps bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import raft as oraft

pytestmark = pytest.mark.gpu


def fmaps(seed, b, h8, w8, c=256):
    rng = np.random.default_rng(seed)
    f1 = torch.from_numpy(rng.normal(size=(b, c, h8, w8)).astype(np.float32))
    f2 = torch.from_numpy(rng.normal(size=(b, c, h8, w8)).astype(np.float32))
    return f1, f2


def coords_for(seed, b, h8, w8, spread):
    rng = np.random.default_rng(seed)
    c0 = oraft.coords_grid(b, h8, w8)
    fl = torch.from_numpy(rng.normal(0, spread, size=(b, 2, h8, w8)).astype(np.float32))
    fl[:, :, : h8 // 3] = torch.round(fl[:, :, : h8 // 3] * 4) / 4          # exact quarter positions
    fl[:, :, -2:] = 0.0                                                      # exact integers (iteration 0 case)
    return c0 + fl


@pytest.mark.parametrize('b,h8,w8', [(2, 32, 40), (1, 64, 80), (2, 17, 23)])
def test_pyramid_and_lookup_match_oracle(rpe, b, h8, w8):
    from rpe_amd import ops
    f1, f2 = fmaps(b + h8, b, h8, w8)
    ref = oraft.CorrBlock(f1, f2, num_levels=4, radius=4)
    pyr = ops.CorrPyramid(b, h8, w8, device='cuda').build(f1.cuda(), f2.cuda())
    scale = float(ref.corr_pyramid[0].abs().max())
    for l in range(4):
        dense = pyr.export_level(l).cpu()
        r = ref.corr_pyramid[l][:, 0]
        assert dense.shape == r.shape, (l, dense.shape, r.shape)
        assert float((dense - r).abs().max()) <= 2e-5 * scale, l
    for spread in (0.0, 2.0, 30.0):                                          # 30: many windows leave the map
        coords = coords_for(7, b, h8, w8, spread)
        out = pyr.lookup(coords.cuda()).cpu()
        expect = ref(coords)
        assert out.shape == expect.shape
        assert float((out - expect).abs().max()) <= 4e-5 * scale, spread


def test_build_reads_fmap1_in_place_when_it_is_already_in_group_order(rpe):
    """64x80: a row is whole 8-pixel groups and the map is whole query tiles, so the build's A operand is fmap1 itself; a
    4-byte-offset copy of the same values cannot be (the LDS-DMA needs 16-byte rows) and goes through the permute pass.
    Same arithmetic either way -> identical pyramid bytes."""
    from rpe_amd import ops
    b, h8, w8 = 1, 64, 80
    f1, f2 = fmaps(5, b, h8, w8)
    g1, g2 = f1.cuda(), f2.cuda()
    shifted = torch.empty(g1.numel() + 1, device='cuda')[1:].view_as(g1)
    shifted.copy_(g1)
    assert g1.data_ptr() % 16 == 0 and shifted.data_ptr() % 16 == 4
    p0 = ops.CorrPyramid(b, h8, w8, device='cuda').build(g1, g2)
    p1 = ops.CorrPyramid(b, h8, w8, device='cuda').build(shifted, g2)
    for l in range(4):
        assert torch.equal(p0.export_level(l), p1.export_level(l)), l


def test_lookup_taps_bit_exact(rpe):
    """Floor indices of every window tap (read back through the side entry rpe_corr_lookup_taps) == floor of the position
    torch's grid_sample computes for upstream's normalised coordinates, RESTATED here in numpy f32 -- not torch.grid_sample itself,
    which returns no indices; the sampled VALUES are compared with torch's own grid_sample (oracle CorrBlock) in
    test_pyramid_and_lookup_match_oracle, incl. windows that leave the map and exact-integer / quarter positions."""
    from rpe_amd import ops
    b, h8, w8 = 2, 32, 40
    pyr = ops.CorrPyramid(b, h8, w8, device='cuda')
    coords = coords_for(11, b, h8, w8, 3.0)
    x0, y0 = pyr.taps(coords.cuda())
    x0, y0 = x0.cpu().numpy(), y0.cpu().numpy()
    c = coords.numpy().reshape(b, 2, -1)
    one, two = np.float32(1), np.float32(2)
    for l in range(4):
        wl, hl = w8 >> l, h8 >> l
        for i in range(9):
            d = np.float32(i - 4)
            for axis, size, got in ((0, wl, x0), (1, hl, y0)):
                v = c[:, axis] / np.float32(2 ** l) + d
                g = two * v / np.float32(size - 1) - one
                pos = ((g + one) / two) * np.float32(size - 1)
                assert np.array_equal(got[:, l, i], np.floor(pos).astype(np.int32)), (l, i, axis)


def test_lookup_handles_nonfinite_and_far_coords(rpe):
    from rpe_amd import ops
    b, h8, w8 = 1, 16, 24
    f1, f2 = fmaps(3, b, h8, w8)
    pyr = ops.CorrPyramid(b, h8, w8, device='cuda').build(f1.cuda(), f2.cuda())
    coords = oraft.coords_grid(b, h8, w8)
    coords[0, 0, 0, 0] = float('nan')
    coords[0, 1, 0, 1] = float('inf')
    coords[0, 0, 0, 2] = 1e12
    coords[0, 0, 0, 3] = -3000.0
    out = pyr.lookup(coords.cuda()).cpu()
    assert bool(torch.isfinite(out).all())
    assert float(out[0, :, 0, :4].abs().max()) == 0.0                       # all taps outside -> zero padding
    ref = oraft.CorrBlock(f1, f2)(oraft.coords_grid(b, h8, w8))
    assert float((out[0, :, 1:] - ref[0, :, 1:]).abs().max()) <= 4e-5 * float(ref.abs().max())


def test_gru_gates(rpe):
    from rpe_amd import ops
    torch.manual_seed(0)
    b, c, h, w = 2, 128, 16, 20
    zr = torch.randn(b, 2 * c, h, w) * 3
    hx = torch.randn(b, 384, h, w)
    q = torch.randn(b, c, h, w) * 3
    z_out = torch.empty(b, c, h, w, device='cuda')
    rhx = hx.clone().cuda()
    bzr = torch.randn(2 * c)
    ops.gru_gates_zr(zr.cuda(), hx.cuda(), c, z_out, rhx, bias=bzr.cuda())
    z_ref = torch.sigmoid(zr[:, :c] + bzr[:c, None, None])
    rh_ref = torch.sigmoid(zr[:, c:] + bzr[c:, None, None]) * hx[:, :c]
    assert torch.allclose(z_out.cpu(), z_ref, atol=1e-6)
    assert torch.allclose(rhx[:, :c].cpu(), rh_ref, atol=1e-6)
    assert torch.equal(rhx[:, c:].cpu(), hx[:, c:])                         # x part untouched
    hx_d = hx.clone().cuda()
    bq = torch.randn(c)
    ops.gru_gates_h(z_out, q.cuda(), hx_d, c, hx_d, bias=bq.cuda())         # in place
    h_ref = (1 - z_ref) * hx[:, :c] + z_ref * torch.tanh(q + bq[:, None, None])
    assert torch.allclose(hx_d[:, :c].cpu(), h_ref, atol=2e-6)
    assert torch.equal(hx_d[:, c:].cpu(), hx[:, c:])


def test_bias_act(rpe):
    from rpe_amd import ops
    torch.manual_seed(2)
    x = torch.randn(2, 126, 16, 20)
    x[0, 0, 0, 0] = float('nan')
    bias = torch.randn(126)
    hx = torch.zeros(2, 384, 16, 20, device='cuda')
    rhx = torch.ones(2, 384, 16, 20, device='cuda')
    ops.bias_act(x.cuda(), bias.cuda(), relu=True, out=hx, out_offset=256, out2=rhx, out2_offset=256)
    ref = torch.relu(x + bias[:, None, None])
    for buf, other in ((hx, 0.0), (rhx, 1.0)):
        got = buf[:, 256:382].cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(ref)) and torch.equal(torch.nan_to_num(got), torch.nan_to_num(ref))
        assert bool((buf[:, :256] == other).all()) and bool((buf[:, 382:] == other).all())
    y = x.clone().cuda()
    ops.bias_act(y, None, relu=False)                                        # identity, in place
    assert torch.equal(torch.nan_to_num(y.cpu()), torch.nan_to_num(x))
    odd = torch.randn(1, 3, 5, 7)                                            # hw % 4 != 0 -> scalar path
    assert torch.equal(ops.bias_act(odd.cuda(), bias[:3].cuda()).cpu(), torch.relu(odd + bias[:3, None, None]))


def test_convex_upsample(rpe):
    from rpe_amd import ops
    torch.manual_seed(1)
    b, h8, w8 = 2, 12, 20
    flow = torch.randn(b, 2, h8, w8) * 4
    mask = torch.randn(b, 576, h8, w8) * 2
    out = ops.upsample_convex(flow.cuda(), mask.cuda()).cpu()
    assert torch.allclose(out, oraft.upsample_flow(flow, mask), atol=2e-5)


def test_high_resolution_geometry_1280x1024(rpe):
    """BASELINE config 5 geometry (1/8 grid 128x160, 20 480 queries): pyramid + lookup vs the oracle."""
    from rpe_amd import ops
    b, h8, w8 = 1, 128, 160
    f1, f2 = fmaps(99, b, h8, w8)
    pyr = ops.CorrPyramid(b, h8, w8, device='cuda').build(f1.cuda(), f2.cuda())
    ref = oraft.CorrBlock(f1, f2, num_levels=4, radius=4)
    scale = float(ref.corr_pyramid[0].abs().max())
    coords = coords_for(13, b, h8, w8, 4.0)
    out = pyr.lookup(coords.cuda()).cpu()
    assert float((out - ref(coords)).abs().max()) <= 4e-5 * scale
    l3 = pyr.export_level(3).cpu()
    assert float((l3 - ref.corr_pyramid[3][:, 0]).abs().max()) <= 2e-5 * scale


def test_encoder_epilogues(rpe):
    """rpe_instnorm_act / rpe_affine_act vs torch's InstanceNorm2d / eval BatchNorm2d + ReLU + residual."""
    from rpe_amd import ops
    torch.manual_seed(3)
    for shape in ((2, 64, 32, 40), (1, 96, 17, 23)):                        # second: hw % 4 != 0
        x = torch.randn(*shape) * 3 + 1
        bias = torch.randn(shape[1])
        res = torch.randn(*shape)
        pre = x + bias[None, :, None, None]
        ref = torch.relu(F.instance_norm(pre, eps=1e-5))
        got = ops.instnorm_act(x.clone().cuda(), bias.cuda(), eps=1e-5, relu=True).cpu()
        assert torch.allclose(got, ref, atol=2e-5)
        ref2 = torch.relu(res + ref)
        got2 = ops.instnorm_act(x.clone().cuda(), bias.cuda(), relu=True, residual=res.cuda()).cpu()
        assert torch.allclose(got2, ref2, atol=2e-5)
        ref3 = F.instance_norm(pre, eps=1e-5)
        got3 = ops.instnorm_act(x.clone().cuda(), bias.cuda(), relu=False).cpu()
        assert torch.allclose(got3, ref3, atol=2e-5)
        bn = torch.nn.BatchNorm2d(shape[1]).eval()
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.data.normal_(); bn.bias.data.normal_()
        with torch.no_grad():
            refb = torch.relu(res + torch.relu(bn(pre)))
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            shift = (bias - bn.running_mean) * scale + bn.bias
        gotb = ops.affine_act(x.clone().cuda(), scale.cuda(), shift.cuda(), relu=True, residual=res.cuda()).cpu()
        assert torch.allclose(gotb, refb, atol=2e-5)


def test_encoders_match_oracle(rpe):
    from rpe_amd import raft as praft
    torch.manual_seed(4)
    img = torch.rand(2, 3, 96, 128) * 2 - 1
    for norm in ('instance', 'batch'):
        enc = praft.BasicEncoder(256, norm).eval()
        oenc = oraft.BasicEncoder(256, norm).eval()
        if norm == 'batch':
            for m in enc.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
        oenc.load_state_dict(enc.state_dict())
        with torch.no_grad():
            ref = oenc(img)
            got = enc.cuda()(img.cuda()).cpu()
        assert float((got - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max())), norm


def test_full_size_lookup_properties(rpe):
    """BASELINE geometry (64x80 queries), properties that need no CPU reference:
    (a) scaling fmap1 by 2 scales every lookup output by exactly 2 (bit for bit: GEMM, pooling and bilinear taps are
        all linear and a power-of-two factor is exact);
    (b) a permutation of the batch permutes the outputs bit for bit (pairs are independent);
    (c) at integer coordinates the centre tap of level 0 is the correlation of the query with its own target pixel."""
    from rpe_amd import ops
    b, h8, w8 = 4, 64, 80
    f1, f2 = fmaps(21, b, h8, w8)
    f1, f2 = f1.cuda(), f2.cuda()
    coords = coords_for(5, b, h8, w8, 3.0).cuda()
    pyr = ops.CorrPyramid(b, h8, w8, device='cuda')
    out = pyr.build(f1, f2).lookup(coords).clone()
    out2 = pyr.build(2.0 * f1, f2).lookup(coords).clone()
    assert torch.equal(out2, 2.0 * out)
    perm = [2, 0, 3, 1]
    outp = pyr.build(f1[perm].contiguous(), f2[perm].contiguous()).lookup(coords[perm].contiguous())
    assert torch.equal(outp, out[perm])
    c0 = oraft.coords_grid(b, h8, w8).cuda()
    centre = pyr.build(f1, f2).lookup(c0)[:, 40]                              # level 0, window index (4,4)
    diag = (f1 * f2).sum(1) / 16.0                                           # <f1[:,q], f2[:,q]> / sqrt(256)
    assert float((centre - diag).abs().max()) <= 4e-5 * float(diag.abs().max())


def test_flow_head_last_layer(rpe):
    from rpe_amd import ops
    torch.manual_seed(5)
    # widths that are a multiple of 4 take the four-pixels-per-thread kernel, the others the one-pixel kernel
    for (b, c, h, w) in ((2, 256, 64, 80), (1, 20, 13, 37), (3, 37, 9, 12), (1, 256, 44, 48), (2, 5, 3, 4)):
        x = torch.randn(b, c, h, w)
        wt = torch.randn(2, c, 3, 3) * 0.05
        bias = torch.randn(2)
        coords = torch.randn(b, 2, h, w) * 10
        ref = F.conv2d(x, wt, bias, padding=1)
        got = ops.conv3x3_to2(x.cuda(), wt.cuda(), bias.cuda()).cpu()
        assert torch.allclose(got, ref, atol=2e-4, rtol=1e-4)
        got2 = ops.conv3x3_to2(x.cuda(), wt.cuda(), bias.cuda(), add=coords.cuda()).cpu()
        assert torch.allclose(got2, coords + ref, atol=2e-4, rtol=1e-4)


def test_flow_head_with_fused_bookkeeping(rpe):
    """rpe_conv3x3_to2_flow: coords1 updated in place, flow = coords1 - coords0 (the integer grid) written to a plain buffer and
    behind the motion features of two 256-channel buffers -- bit-identical to the separate conv + subtract + copies."""
    from rpe_amd import ops
    torch.manual_seed(6)
    for (b, c, h, w) in ((32, 256, 64, 80), (1, 256, 64, 80), (2, 256, 44, 48), (1, 20, 13, 37)):
        x = torch.randn(b, c, h, w).cuda()
        wt, bias = (torch.randn(2, c, 3, 3) * 0.05).cuda(), torch.randn(2).cuda()
        coords0 = oraft.coords_grid(b, h, w).cuda()
        coords1 = (coords0 + torch.randn(b, 2, h, w).cuda() * 10).contiguous()
        want = ops.conv3x3_to2(x, wt, bias, add=coords1)
        hx, rhx = torch.full((b, 256, h, w), -7.0, device='cuda'), torch.full((b, 256, h, w), -7.0, device='cuda')
        flow = torch.empty(b, 2, h, w, device='cuda')
        got = ops.flow_update(x, wt, bias, coords1, coords1, flow_out=flow, dst1=hx[:, 254:], dst2=rhx[:, 254:])
        assert got is coords1 and torch.equal(coords1, want)
        assert torch.equal(flow, want - coords0)
        assert torch.equal(hx[:, 254:], flow) and torch.equal(rhx[:, 254:], flow)
        assert bool((hx[:, :254] == -7.0).all()) and bool((rhx[:, :254] == -7.0).all())
        again = ops.flow_update(x, wt, bias, want, torch.empty_like(want), prepare=True)()      # no extra destinations, prepared launcher
        assert torch.equal(again, ops.conv3x3_to2(x, wt, bias, add=want))


def test_flow_head_kernels_agree_bitwise(rpe):
    """A large launch takes the four-pixels-per-thread kernel (neighbours by DPP wave shifts), a small one the one-pixel kernel:
    same channel slices, tap order and explicit fused multiply-adds, so a map's flow update is the same bits in any batch."""
    from rpe_amd import ops
    torch.manual_seed(8)
    for (b, c, h, w) in ((32, 256, 64, 80), (40, 128, 44, 48)):
        x = torch.randn(b, c, h, w).cuda()
        wt, bias = (torch.randn(2, c, 3, 3) * 0.05).cuda(), torch.randn(2).cuda()
        add = (torch.randn(b, 2, h, w) * 10).cuda()
        big = ops.conv3x3_to2(x, wt, bias, add=add)
        ref = F.conv2d(x.double().cpu(), wt.double().cpu(), bias.double().cpu(), padding=1) + add.double().cpu()
        assert float((big.cpu().double() - ref).abs().max()) < 2e-4
        for sl in (slice(0, 1), slice(b - 2, b)):
            small = ops.conv3x3_to2(x[sl].contiguous(), wt, bias, add=add[sl].contiguous())
            assert torch.equal(small, big[sl])


def test_copy_planes(rpe):
    from rpe_amd import ops
    src = torch.randn(3, 10, 7, 12, device='cuda')
    dst = torch.full((3, 16, 7, 12), 5.0, device='cuda')
    ops.copy_planes(src[:, 2:7], dst[:, 9:14])
    assert torch.equal(dst[:, 9:14], src[:, 2:7]) and bool((dst[:, :9] == 5.0).all()) and bool((dst[:, 14:] == 5.0).all())
    odd = torch.randn(2, 3, 5, 7, device='cuda')                                                # plane size not a multiple of 4
    assert torch.equal(ops.copy_planes(odd[:, 1:], torch.empty(2, 2, 5, 7, device='cuda')), odd[:, 1:])
    with pytest.raises(rpe.RpeError):
        ops.copy_planes(src[:, :2], dst[:, :3])


def test_lookup_round_diagnostic(rpe):
    """rpe_corr_lookup_rounds: one round per group for smooth flow; a flow discontinuity inside a group costs extra rounds."""
    from rpe_amd import ops
    b, h8, w8 = 2, 32, 40
    pyr = ops.CorrPyramid(b, h8, w8, device='cuda')
    c0 = oraft.coords_grid(b, h8, w8).cuda()
    rounds, lines = pyr.rounds((c0 + 0.3).contiguous())
    assert rounds.shape == (b, 4, h8 * 5) and int(rounds.min()) == int(rounds.max()) == 1 and int(lines.min()) > 0
    jump = c0.clone()
    jump[:, 0, :, 20:] += 9.0                                  # queries 20.. of every row jump 9 px: groups 2 (x 16..23) straddle it
    r2, l2 = pyr.rounds(jump)
    r2 = r2.reshape(b, 4, h8, 5)
    assert int(r2[:, 0, :, 2].min()) == 2 and int(r2[:, 0, :, [0, 1, 3, 4]].max()) == 1
    l1g, l2g = lines.reshape(b, 4, h8, 5)[:, 0, 4:-4, 2], l2.reshape(b, 4, h8, 5)[:, 0, 4:-4, 2]
    assert bool((l2g > l1g).all())                              # the straddling groups fetch more lines (interior rows)
    far = (c0 + 1.0e4).contiguous()                             # every window outside its map: nothing to fetch
    r3, l3 = pyr.rounds(far)
    assert int(r3.max()) == 0 and int(l3.max()) == 0


@pytest.mark.parametrize('b,h8,w8', [(2, 32, 40), (1, 17, 23), (1, 128, 160)])
def test_fp16_feature_pyramid_matches_oracle(rpe, b, h8, w8):
    """BASELINE config 5 ("fp16 features"): feature maps rounded to fp16, 16-bit MFMA with f32 accumulation, f32 pyramid --
    against the oracle's f32 correlation of the SAME rounded maps (products of fp16 values are exact in f32, so only the
    summation order differs), at the 1280x1024 geometry too."""
    from rpe_amd import ops
    f1, f2 = fmaps(50 + h8, b, h8, w8)
    ref = oraft.CorrBlock(f1.half().float(), f2.half().float(), num_levels=4, radius=4)
    pyr = ops.CorrPyramid(b, h8, w8, device='cuda').build(f1.cuda(), f2.cuda(), fp16_features=True)
    scale = float(ref.corr_pyramid[0].abs().max())
    for l in (0, 3):
        dense = pyr.export_level(l).cpu()
        assert float((dense - ref.corr_pyramid[l][:, 0]).abs().max()) <= 2e-5 * scale, l
    coords = coords_for(17, b, h8, w8, 3.0)
    out = pyr.lookup(coords.cuda()).cpu()
    assert float((out - ref(coords)).abs().max()) <= 4e-5 * scale
    # and it really is a different (rounded) result than the f32 build
    f32 = ops.CorrPyramid(b, h8, w8, device='cuda').build(f1.cuda(), f2.cuda()).export_level(0).cpu()
    assert float((f32 - pyr.export_level(0).cpu()).abs().max()) > 1e-4 * scale


def test_bf16x3_build_is_f32_equivalent(rpe):
    """EXPERIMENT switch (RPE_F32X3 / RPE_CORR_BF16X3=1; off by default, never in bench.py's headline): every f32 product of the
    correlation as six bf16 products of an exact three-way split, f32 accumulation.  Its error against an f64 evaluation must stay
    within 1.25x the f32 matrix pipe's on the same data (measured 0.85x on Gaussian maps, 1.08x on post-ReLU maps with heavy-tailed
    channel scales), and the pooled levels / the lookup run on it unchanged."""
    from rpe_amd import ops
    b, h8, w8 = 2, 32, 40
    rng = np.random.default_rng(17)
    for heavy in (False, True):
        f1 = torch.from_numpy(rng.normal(size=(b, 256, h8, w8)).astype(np.float32))
        f2 = torch.from_numpy(rng.normal(size=(b, 256, h8, w8)).astype(np.float32))
        if heavy:
            f1 = torch.relu(f1 + 1) * torch.from_numpy(np.exp(0.8 * rng.normal(size=(b, 256, 1, 1))).astype(np.float32))
            f2 = torch.relu(f2 + 1) * torch.from_numpy(np.exp(0.8 * rng.normal(size=(b, 256, 1, 1))).astype(np.float32))
        ref = torch.einsum('bcq,bcp->bqp', f1.double().reshape(b, 256, -1), f2.double().reshape(b, 256, -1)) / 16.0
        errs = {}
        for name, kw in (('f32', {}), ('x3', dict(bf16x3=True))):
            pyr = ops.CorrPyramid(b, h8, w8, device='cuda', **kw).build(f1.cuda(), f2.cuda(), **kw)
            got = pyr.export_level(0).cpu().double().reshape(b, h8 * w8, h8 * w8)
            errs[name] = float((got - ref).pow(2).mean().sqrt())
            if name == 'x3':
                oref = oraft.CorrBlock(f1, f2, num_levels=4, radius=4)
                scale = float(oref.corr_pyramid[0].abs().max())
                for l in range(1, 4):
                    assert float((pyr.export_level(l).cpu() - oref.corr_pyramid[l][:, 0]).abs().max()) <= 2e-5 * scale, l
                coords = coords_for(7, b, h8, w8, 2.0)
                assert float((pyr.lookup(coords.cuda()).cpu() - oref(coords)).abs().max()) <= 4e-5 * scale
        print('rms error vs f64:', errs)
        assert errs['x3'] <= 1.25 * errs['f32'], errs
    # the default pyramid is sized for the f32 / fp16 builds only (the experiment's three bf16 planes need 1.5x the feature scratch)
    L = rpe.lib()
    assert L.rpe_corr_pyramid_bytes(b, h8, w8, 4) == L.rpe_corr_pyramid_bytes_ex(b, h8, w8, 4, 0) == L.rpe_corr_pyramid_bytes_ex(b, h8, w8, 4, 2)
    assert L.rpe_corr_pyramid_bytes_ex(b, h8, w8, 4, 3) > L.rpe_corr_pyramid_bytes(b, h8, w8, 4)
    with pytest.raises(rpe.RpeError):
        ops.CorrPyramid(b, h8, w8, device='cuda').build(f1.cuda(), f2.cuda(), bf16x3=True)


def test_build_launch_classes_agree_bitwise(rpe):
    """k_corr_build walks a band's patches in one workgroup (level 0 through the LDS carry) or, for launches of fewer than 4096 band
    workgroups, takes one workgroup per patch: the pyramid of a pair must not depend on how many pairs share the launch (the chunked
    sequence tracker builds 32 pairs at once, the frame-at-a-time one 2).  Every level, bit for bit, at bench geometry."""
    from rpe_amd import ops
    torch.manual_seed(3)
    b, c, h8, w8 = 16, 256, 64, 80                                     # 16 pairs: 5120 band workgroups -> the walk; 2 pairs -> per patch
    f1, f2 = torch.randn(b, c, h8, w8, device='cuda'), torch.randn(b, c, h8, w8, device='cuda')
    big = ops.CorrPyramid(b, h8, w8, device='cuda'); big.build(f1, f2)
    small = ops.CorrPyramid(2, h8, w8, device='cuda'); small.build(f1[5:7].contiguous(), f2[5:7].contiguous())
    for level in range(4):
        lb, ls = big.export_level(level), small.export_level(level)   # (pairs * queries, h_l, w_l)
        assert torch.equal(lb.reshape(b, h8 * w8, *lb.shape[1:])[5:7], ls.reshape(2, h8 * w8, *ls.shape[1:]))
    # and the lookups read the same values through both layouts' edges
    coords = torch.rand(2, 2, h8, w8, device='cuda') * torch.tensor([w8, h8], device='cuda').view(1, 2, 1, 1)
    cb = torch.zeros(b, 2, h8, w8, device='cuda'); cb[5:7] = coords
    assert torch.equal(big.lookup(cb)[5:7], small.lookup(coords))


@pytest.mark.parametrize('b,h8,w8,spread', [(2, 64, 80, 0.7), (1, 44, 48, 3.0), (3, 20, 24, 12.0), (2, 17, 16, 1.5), (7, 64, 80, 1.0), (9, 44, 48, 6.0),
                                            (40, 20, 24, 12.0)])
def test_lookup_fused_into_convc1_is_bit_identical_to_the_two_kernels(rpe, b, h8, w8, spread):
    """rpe_corr_lookup_conv1x1 (the 324-channel lookup result stays in LDS and is contracted there) against rpe_corr_lookup followed by
    convc1 on both of its routes (rpe_conv1x1's GEMM and rpe_conv_fused's implicit GEMM, which are bit-identical to each other): torch.equal,
    with smooth and rough flow (several staging rounds per group at spread 12), windows partly and wholly outside the maps, non-finite
    coordinates, with and without ReLU, into channel slices, with the second destination, through the prepared launcher; a group count that
    is not a multiple of the workgroup's eight (17 x 16).  The last three cases have more 64-query tiles than the chip has CUs (560, 297,
    320): they run the PERSISTENT pipelined kernel (lookup waves one tile ahead of the matrix waves through one LDS tile), with two or three
    tiles per workgroup and rough flow (several staging rounds inside a gated lookup)."""
    from rpe_amd import ops
    f1, f2 = fmaps(7 * b + h8, b, h8, w8)
    pyr = ops.CorrPyramid(b, h8, w8, device='cuda').build(f1.cuda(), f2.cuda())
    co = coords_for(b + w8, b, h8, w8, spread)
    co[0, 0, 1, 2] = float('nan'); co[0, 1, 2, 3] = float('inf'); co[0, :, 3, 4] = -1.0e9; co[0, :, 0, 0] = -7.5; co[-1, 0, -1, -1] = w8 + 9.25
    co = co.cuda()
    rng = np.random.default_rng(h8)
    wt = torch.from_numpy(rng.normal(0, 0.05, size=(256, 324, 1, 1)).astype(np.float32)).cuda()
    bias = torch.from_numpy(rng.normal(0, 0.5, size=(256,)).astype(np.float32)).cuda()
    corr = pyr.lookup(co)
    packed = ops.PackedLookupConv(wt, bias)
    for relu in (True, False):
        mode = ops.CONV_RELU if relu else ops.CONV_LINEAR
        ref_gemm = ops.conv1x1(corr, ops.PackedConv1x1(wt, bias), mode, torch.empty(b, 256, h8, w8, device='cuda'))
        ref_igemm = ops.conv_fused(corr, ops.PackedConv(wt, bias), mode, torch.empty(b, 256, h8, w8, device='cuda'))
        assert torch.equal(torch.nan_to_num(ref_gemm), torch.nan_to_num(ref_igemm)) and torch.equal(torch.isnan(ref_gemm), torch.isnan(ref_igemm))
        obuf = torch.full((b, 264, h8, w8), -7.0, device='cuda'); o2 = torch.full((b, 260, h8, w8), -7.0, device='cuda')
        pyr.lookup_conv1x1(co, packed, obuf[:, 4:260], out2=o2[:, 2:258], relu=relu)
        got = obuf[:, 4:260]
        assert torch.equal(torch.isnan(got), torch.isnan(ref_gemm)) and torch.equal(torch.nan_to_num(got), torch.nan_to_num(ref_gemm))
        assert torch.equal(torch.nan_to_num(o2[:, 2:258]), torch.nan_to_num(got))
        assert (obuf[:, :4] == -7.0).all() and (obuf[:, 260:] == -7.0).all() and (o2[:, :2] == -7.0).all() and (o2[:, 258:] == -7.0).all()
        again = torch.empty(b, 256, h8, w8, device='cuda')
        pyr.lookup_conv1x1(co, packed, again, relu=relu, prepare=True)()
        assert torch.equal(torch.nan_to_num(again), torch.nan_to_num(got))
    assert bool(torch.isfinite(ref_gemm).all())                       # non-finite coordinates sample zeros (rpe_corr_lookup), as in both routes above


def test_lookup_fused_into_convc1_refuses_what_it_cannot_do(rpe):
    from rpe_amd import ops
    with pytest.raises(rpe.RpeError):
        ops.PackedLookupConv(torch.zeros(256, 320, 1, 1, device='cuda'))
    with pytest.raises(rpe.RpeError):
        ops.PackedLookupConv(torch.zeros(128, 324, 1, 1, device='cuda'))
    f1, f2 = fmaps(3, 1, 16, 20)
    pyr = ops.CorrPyramid(1, 16, 20, device='cuda').build(f1.cuda(), f2.cuda())       # w8 = 20: rows are not whole groups of 8
    with pytest.raises(rpe.RpeError):
        pyr.lookup_conv1x1(coords_for(1, 1, 16, 20, 1.0).cuda(), ops.PackedLookupConv(torch.zeros(256, 324, 1, 1, device='cuda')),
                           torch.empty(1, 256, 16, 20, device='cuda'))
