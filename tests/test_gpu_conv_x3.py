"""GPU: rpe_conv_wino_x3 -- the LABELLED bf16x3 variant of the Winograd F(2x2,3x3) convolution (csrc/conv_wino_x3.hip): every f32 product of
the Winograd domain as six bf16 products of an exact three-way split on the 16-bit matrix cores, f32 accumulation.

What "f32-equivalent" means here and how it is held: the variant replaces rpe_conv_wino (csrc/conv_wino.hip, f32 matrix cores), with which
it shares the f32 Winograd transforms bit for bit; only the products differ.  So its error against the f64 convolution is compared with
THAT kernel's on the same data -- RMS at most 1.25x (measured 0.85-0.88x: the split drops terms below 2^-24 of a product and
accumulates in the matrix core's f32 adder tree instead of a serial fma chain) -- and, like rpe_conv_wino itself, with the direct f32
kernel's (RMS <= 3x, max <= 4x: the Winograd transforms' cost, test_gpu_conv.py).  The f64 bars of test_gpu_conv.py's Winograd tests are
kept unchanged for every epilogue.  The second half of the file holds rpe_conv_wino1d_x3 (the GRU's F(4,5) layers, csrc/conv_wino1d_x3.hip)
to the same bars against rpe_conv_wino1d."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_conv import _errs, _rand, _tol, _trained_like

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('cin,cout,h,w,b', [(256, 192, 64, 80, 2), (128, 64, 64, 80, 1), (256, 126, 44, 48, 2), (128, 256, 32, 40, 1),
                                            (16, 20, 6, 12, 3), (64, 96, 128, 160, 1), (32, 32, 16, 16, 2), (48, 160, 18, 20, 1)])
def test_x3_matches_f64(rpe, cin, cout, h, w, b):
    """The update block's shapes, ragged channel counts (126; 20), the trailing 32-channel tile (96, 160, 32), maps that are not whole
    16 x 16 patches (44 x 48, 6 x 12, 18 x 20), one K step (cin = 16).  Same bar as test_winograd_3x3_matches_f64.  Destinations are
    channel slices; out2 receives a copy; neighbours stay untouched; the prepared launcher gives the same bits."""
    from rpe_amd import ops
    rng = np.random.default_rng(cin + cout + h)
    x, wt, bias = _rand(rng, b, cin, h, w), _rand(rng, cout, cin, 3, 3, s=0.05), _rand(rng, cout, s=0.5)
    assert ops.PackedWinoX3.supported(wt, h, w)
    px = ops.PackedWinoX3(wt.cuda(), bias.cuda())
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    obuf = torch.full((b, cout + 8, h, w), -7.0, device='cuda')
    o2buf = torch.full((b, cout + 4, h, w), -7.0, device='cuda')
    xbuf = torch.zeros(b, cin + 4, h, w, device='cuda')                      # the input is a channel slice too (16-byte aligned planes)
    xbuf[:, 4:] = x.cuda()
    ops.conv_wino(xbuf[:, 4:], px, ops.CONV_RELU, obuf[:, 4:4 + cout], out2=o2buf[:, 4:])
    got = obuf[:, 4:4 + cout].cpu().double()
    assert (got - ref.clamp_min(0)).abs().max() < 3 * _tol(x, wt)
    assert torch.equal(obuf[:, 4:4 + cout], o2buf[:, 4:])
    assert (obuf[:, :4] == -7.0).all() and (obuf[:, 4 + cout:] == -7.0).all() and (o2buf[:, :4] == -7.0).all()
    lin = ops.conv_wino(x.cuda(), px, ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda'))
    assert (lin.cpu().double() - ref).abs().max() < 3 * _tol(x, wt)
    out3 = torch.empty(b, cout, h, w, device='cuda')
    ops.conv_wino(x.cuda(), px, ops.CONV_LINEAR, out3, prepare=True)()
    assert torch.equal(out3, lin)
    # deterministic: fixed summation order, no atomics
    assert torch.equal(ops.conv_wino(x.cuda(), px, ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda')), lin)


def test_x3_rejects_what_it_cannot_do(rpe):
    from rpe_amd import ops
    with pytest.raises(rpe.RpeError):
        ops.PackedWinoX3(torch.zeros(8, 24, 3, 3, device='cuda'))            # cin % 16
    with pytest.raises(rpe.RpeError):
        ops.PackedWinoX3(torch.zeros(8, 16, 1, 5, device='cuda'))
    px = ops.PackedWinoX3(torch.zeros(8, 16, 3, 3, device='cuda'))
    for hh, ww in ((7, 12), (8, 10)):                                         # odd height; rows that are not whole 16-byte quads
        with pytest.raises(rpe.RpeError, match='UNSUPPORTED'):
            ops.conv_wino(torch.zeros(1, 16, hh, ww, device='cuda'), px, ops.CONV_RELU, torch.empty(1, 8, hh, ww, device='cuda'))
    shifted = torch.zeros(16 * 6 * 12 + 1, device='cuda')[1:].view(1, 16, 6, 12)        # an input that is not 16-byte aligned
    with pytest.raises(rpe.RpeError, match='UNSUPPORTED'):
        ops.conv_wino(shifted, px, ops.CONV_RELU, torch.empty(1, 8, 6, 12, device='cuda'))


@pytest.mark.parametrize('c,h,w,b', [(64, 64, 80, 3), (96, 44, 48, 2), (128, 32, 40, 2)])
def test_x3_encoder_epilogues_match_f64(rpe, c, h, w, b):
    """The encoders' epilogues on the variant, same references and bars as test_winograd_encoder_epilogues_match_f64: folded batch norm +
    ReLU + residual + ReLU; instance-norm moments (the SAME record regions and layout as rpe_conv_wino: the consumers do not know which
    kernel ran -- counts equal exactly, means / M2 to rounding); the input normalised + ReLU'd on the way in (pre_norm)."""
    from rpe_amd import ops
    rng = np.random.default_rng(c + h + 1)
    x, wt, bias = _rand(rng, b, c, h, w), _rand(rng, c, c, 3, 3, s=0.05), _rand(rng, c, s=0.5)
    res = _rand(rng, b, c, h, w).abs()
    scale, shift = _rand(rng, c).abs() + 0.5, _rand(rng, c, s=0.3)
    px, pw = ops.PackedWinoX3(wt.cuda(), None), ops.PackedWino(wt.cuda(), None)
    conv = F.conv2d(x.double(), wt.double(), None, padding=1)
    ref = (res.double() + (conv * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]).clamp_min(0)).clamp_min(0)
    got = ops.conv_wino(x.cuda(), px, ops.CONV_RELU, torch.empty(b, c, h, w, device='cuda'), scale=scale.cuda(), bias=shift.cuda(), residual=res.cuda())
    assert (got.cpu().double() - ref).abs().max() < 3 * _tol(x, wt) * 2.5
    pre = conv + bias.double()[None, :, None, None]
    mean, var = pre.mean((2, 3)), pre.var((2, 3), unbiased=False)
    stats, stats32 = ops.conv_wino_stats_buffer(b, c, h, w, 'cuda'), ops.conv_wino_stats_buffer(b, c, h, w, 'cuda')
    stats.tensor.fill_(float('nan'))
    raw = ops.conv_wino(x.cuda(), px, ops.CONV_LINEAR, torch.empty(b, c, h, w, device='cuda'), bias=bias.cuda(), stats=stats)
    raw32 = ops.conv_wino(x.cuda(), pw, ops.CONV_LINEAR, torch.empty(b, c, h, w, device='cuda'), bias=bias.cuda(), stats=stats32)
    assert (raw.cpu().double() - pre).abs().max() < 3 * _tol(x, wt)
    st, st32 = stats.cpu().double(), stats32.cpu().double()                  # tile-major: (b, records, c, 3)
    assert not torch.isnan(st).any()                                          # every record of the f32 kernel's layout is written
    assert torch.equal(st[..., 0], st32[..., 0])                              # the same pixel counts, record by record
    assert float(st[..., 0].sum(1).min()) == float(st[..., 0].sum(1).max()) == h * w
    live = st[..., 0] > 0
    assert float((st[..., 1] - st32[..., 1])[live].abs().max()) < 1e-4 and float(((st[..., 2] - st32[..., 2])[live].abs() / (st32[..., 2][live] + 1e-3)).max()) < 1e-3
    mi = ops.instnorm_finalize(stats, h * w, eps=1e-5).cpu().double()
    assert float((mi[..., 0] - mean).abs().max()) < 1e-5 and float((mi[..., 1] * torch.sqrt(var + 1e-5) - 1).abs().max()) < 2e-5
    ref2 = (res.double() + ((pre - mean[:, :, None, None]) / torch.sqrt(var + 1e-5)[:, :, None, None]).clamp_min(0)).clamp_min(0)
    got2 = ops.instnorm_apply(raw, stats, eps=1e-5, relu=True, residual=res.cuda())
    inv = float((1 / torch.sqrt(var + 1e-5)).max())
    assert (got2.cpu().double() - ref2).abs().max() < (3 * _tol(x, wt) + 2e-6) * inv * 2
    m_i = torch.stack((_rand(rng, b, c, s=0.3), _rand(rng, b, c).abs() + 0.5), dim=-1).contiguous()
    xin = ((x.double() - m_i[..., 0].double()[:, :, None, None]) * m_i[..., 1].double()[:, :, None, None]).clamp_min(0)
    ref3 = F.conv2d(xin, wt.double(), bias.double(), padding=1)
    got3 = ops.conv_wino(x.cuda(), px, ops.CONV_LINEAR, torch.empty(b, c, h, w, device='cuda'), bias=bias.cuda(), pre_norm=m_i.cuda())
    assert (got3.cpu().double() - ref3).abs().max() < 3 * _tol(xin.float(), wt) + 1e-5
    # moments + pre_norm together (fnet's second convolution of a block), against the f32 kernel
    s_a, s_b = ops.conv_wino_stats_buffer(b, c, h, w, 'cuda'), ops.conv_wino_stats_buffer(b, c, h, w, 'cuda')
    g_a = ops.conv_wino(x.cuda(), px, ops.CONV_LINEAR, torch.empty(b, c, h, w, device='cuda'), bias=bias.cuda(), stats=s_a, pre_norm=m_i.cuda())
    g_b = ops.conv_wino(x.cuda(), pw, ops.CONV_LINEAR, torch.empty(b, c, h, w, device='cuda'), bias=bias.cuda(), stats=s_b, pre_norm=m_i.cuda())
    assert (g_a - g_b).abs().max() < 3 * _tol(xin.float(), wt) + 1e-5
    ma, mb = ops.instnorm_finalize(s_a, h * w).cpu(), ops.instnorm_finalize(s_b, h * w).cpu()
    assert float((ma[..., 0] - mb[..., 0]).abs().max()) < 1e-5 and float(((ma[..., 1] - mb[..., 1]).abs() / mb[..., 1]).max()) < 2e-5


@pytest.mark.parametrize('cin,cout,h,w', [(256, 192, 64, 80), (128, 256, 64, 80), (64, 64, 128, 160), (256, 126, 44, 48)])
@pytest.mark.parametrize('flow_channels', [False, True])
def test_x3_error_relative_to_the_f32_kernels_on_trained_like_statistics(rpe, cin, cout, h, w, flow_channels):
    """Trained-like statistics (post-ReLU activations with a positive mean, a saturated hidden state, +-50 px flow channels, heavy-tailed
    weights with outliers: test_gpu_conv._trained_like), error against the f64 convolution:
      * RMS at most 1.25x the f32 Winograd kernel's (rpe_conv_wino: the kernel the variant replaces) -- the f32-equivalence bar (measured
        0.85-0.88x);
      * max at most 1.25x the larger of the two f32 kernels' maxima (the maximum over ~10^6 outputs is an extreme-value statistic: against
        rpe_conv_wino alone it reads 0.8-1.3x from one seed to the next while staying at 0.25-0.55x the direct kernel's);
      * at most 3x (RMS) / 4x (max) the direct f32 kernel's (rpe_conv_fused), the bar rpe_conv_wino itself is held to."""
    from rpe_amd import ops
    rng = np.random.default_rng(cin * 1000 + cout * 7 + h + int(flow_channels))
    b = 2
    x, wt, bias = _trained_like(rng, b, cin, cout, 3, 3, h, w, flow_channels)
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    direct = ops.conv_fused(x.cuda(), ops.PackedConv(wt.cuda(), bias.cuda()), ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda'))
    wino = ops.conv_wino(x.cuda(), ops.PackedWino(wt.cuda(), bias.cuda()), ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda'))
    x3 = ops.conv_wino(x.cuda(), ops.PackedWinoX3(wt.cuda(), bias.cuda()), ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda'))
    dmax, drms = _errs(direct, ref)
    wmax, wrms = _errs(wino, ref)
    xmax, xrms = _errs(x3, ref)
    print(f'3x3 {cin}->{cout} flow_channels={flow_channels}: |out| {float(ref.abs().max()):.1f}; direct max {dmax:.2e} rms {drms:.2e}; f32 winograd max {wmax:.2e} '
          f'rms {wrms:.2e}; bf16x3 winograd max {xmax:.2e} rms {xrms:.2e}; x3 / f32-winograd {xmax / wmax:.2f} / {xrms / wrms:.2f}; x3 / direct {xmax / dmax:.2f} / {xrms / drms:.2f}')
    assert xrms <= 1.25 * wrms and xmax <= 1.25 * max(wmax, dmax)
    assert xrms <= 3.0 * drms and xmax <= 4.0 * dmax


def test_x3_special_values(rpe):
    """Inf / NaN inputs poison exactly the outputs they reach (no zero-times-Inf NaNs elsewhere: out-of-map patch elements are zeros, never
    clamped copies), and subnormal-scale data comes through: the truncation split is exact down to the f32 subnormal range."""
    from rpe_amd import ops
    rng = np.random.default_rng(5)
    x, wt = _rand(rng, 1, 16, 16, 16), _rand(rng, 32, 16, 3, 3, s=0.05)
    x[0, 3, 7, 9] = float('inf'); x[0, 5, 0, 0] = float('nan')
    px = ops.PackedWinoX3(wt.cuda(), None)
    got = ops.conv_wino(x.cuda(), px, ops.CONV_LINEAR, torch.empty(1, 32, 16, 16, device='cuda')).cpu()
    bad = ~torch.isfinite(got)
    reach = torch.zeros(16, 16, dtype=torch.bool)
    reach[5:11, 7:13] = True                    # the 2x2 tiles whose 4x4 patches contain (7, 9): rows 6..9 / cols 8..11 as outputs, +-1 tile
    reach[0:2, 0:2] = True
    assert bad[0].any(0)[7, 9] and bad[0].any(0)[0, 0]
    assert not bad[0].any(0)[~reach].any()
    xs = (_rand(rng, 1, 16, 16, 16) * 1e-30).cuda()
    ref = F.conv2d(xs.cpu().double(), wt.double(), None, padding=1)
    gs = ops.conv_wino(xs, px, ops.CONV_LINEAR, torch.empty(1, 32, 16, 16, device='cuda')).cpu().double()
    assert (gs - ref).abs().max() < 1e-5 * ref.abs().max()


# ---- rpe_conv_wino1d_x3: the GRU's 1x5 / 5x1 convolutions (csrc/conv_wino1d_x3.hip), the variant of rpe_conv_wino1d ---------------------

from test_gpu_conv import _ref_conv


@pytest.mark.parametrize('kh,kw,h,w', [(1, 5, 64, 80), (5, 1, 64, 80), (1, 5, 44, 48), (5, 1, 44, 48), (1, 5, 30, 40), (5, 1, 30, 40)])
def test_x3_gru_half_step_matches_f64(rpe, kh, kw, h, w):
    """test_gpu_conv.test_gru_half_step_matches_f64 on the variant, same reference and bars: z, r*h and the blended hidden state of one
    SepConvGRU half fused into the two convolutions (the branch-free gate passes: addend, no bias), in place on the hidden state."""
    from rpe_amd import ops
    c, b = 128, 2
    rng = np.random.default_rng(kh * 10 + kw + h)
    hx = _rand(rng, b, 2 * c, h, w, s=0.5)
    wzr, azr = _rand(rng, 2 * c, 2 * c, kh, kw, s=0.03), _rand(rng, b, 2 * c, h, w, s=0.3)
    wq, aq = _rand(rng, c, 2 * c, kh, kw, s=0.03), _rand(rng, b, c, h, w, s=0.3)
    hid = hx[:, :c].double()
    zr = torch.sigmoid(_ref_conv(hx, wzr, None, azr))
    z, r = zr[:, :c], zr[:, c:]
    rhx = torch.cat((r * hid, hx[:, c:].double()), 1)
    q = torch.tanh(_ref_conv(rhx.float(), wq, None, aq))
    hnew = (1 - z) * hid + z * q
    g_hx, g_rhx = hx.cuda(), hx.cuda().clone()
    g_z = torch.empty(b, c, h, w, device='cuda')
    pzr, pq = ops.PackedWino1dX3(wzr.cuda()), ops.PackedWino1dX3(wq.cuda())
    ops.conv_wino1d(g_hx, pzr, ops.CONV_GATE_ZR, g_z, out2=g_rhx[:, :c], add=azr.cuda(), hidden=g_hx[:, :c], gate_channels=c)
    tz = _tol(hx, wzr) * 0.25 + 2e-7
    assert (g_z.cpu().double() - z).abs().max() < tz
    assert (g_rhx[:, :c].cpu().double() - r * hid).abs().max() < tz * float(hx.abs().max())
    assert torch.equal(g_rhx[:, c:], g_hx[:, c:])
    ops.conv_wino1d(g_rhx, pq, ops.CONV_GATE_H, g_hx[:, :c], add=aq.cuda(), hidden=g_hx[:, :c], zgate=g_z)   # in place on h
    assert (g_hx[:, :c].cpu().double() - hnew).abs().max() < _tol(hx, wq) + tz * 2
    assert torch.equal(g_hx[:, c:].cpu(), hx[:, c:])
    # with a bias (the gate passes with runtime operands) and through the prepared launcher: same values
    bzr = _rand(rng, 2 * c, s=0.1)
    z2 = torch.sigmoid(_ref_conv(hx, wzr, bzr, azr))[:, :c]
    g_z2, g_r2 = torch.empty(b, c, h, w, device='cuda'), torch.empty(b, c, h, w, device='cuda')
    ops.conv_wino1d(hx.cuda(), ops.PackedWino1dX3(wzr.cuda(), bzr.cuda()), ops.CONV_GATE_ZR, g_z2, out2=g_r2, add=azr.cuda(), hidden=hx.cuda()[:, :c],
                    gate_channels=c, prepare=True)()
    assert (g_z2.cpu().double() - z2).abs().max() < tz


@pytest.mark.parametrize('cin,cout,kh,kw,h,w,b', [
    (256, 256, 1, 5, 64, 80, 2),      # convz1|convr1 at bench geometry
    (256, 128, 5, 1, 64, 80, 2),      # convq2
    (64, 96, 1, 5, 20, 24, 1),        # ragged output channels, map smaller than two patches
    (64, 200, 5, 1, 7, 36, 2),        # 7 rows: the last 4-pixel tile is cut, 36 columns: the last patch is cut; a second, ragged channel tile
    (16, 16, 1, 5, 33, 4, 1),         # one quad wide, one K step
    (16, 16, 5, 1, 3, 4, 1),          # shorter than the filter
    (48, 70, 1, 5, 18, 20, 3),        # three K steps; 70 channels: the upper 64-channel half of the tile holds six
])
@pytest.mark.parametrize('relu', [False, True])
def test_x3_winograd_1d_matches_f64(rpe, cin, cout, kh, kw, h, w, b, relu):
    """test_gpu_conv.test_winograd_1d_matches_f64 on the variant: the plain epilogues with bias, addend and a second output, on channel
    slices of wider buffers; same bar."""
    from rpe_amd import ops
    rng = np.random.default_rng(cin + cout + kh * 7 + kw + h)
    x, wt, bias, add = _rand(rng, b, cin, h, w), _rand(rng, cout, cin, kh, kw, s=0.05), _rand(rng, cout, s=0.1), _rand(rng, b, cout, h, w, s=0.3)
    ref = _ref_conv(x, wt, bias, add)
    if relu:
        ref = ref.clamp_min(0)
    xbuf = torch.full((b, cin + 8, h, w), float('nan'), device='cuda'); xbuf[:, 4:4 + cin] = x.cuda()
    obuf = torch.full((b, cout + 8, h, w), -7.0, device='cuda'); o2buf = torch.full((b, cout + 4, h, w), -7.0, device='cuda')
    assert ops.PackedWino1dX3.supported(wt, w)
    pw = ops.PackedWino1dX3(wt.cuda(), bias.cuda())
    ops.conv_wino1d(xbuf[:, 4:4 + cin], pw, ops.CONV_RELU if relu else ops.CONV_LINEAR, obuf[:, 4:4 + cout], out2=o2buf[:, 4:], add=add.cuda())
    got = obuf[:, 4:4 + cout].cpu().double()
    assert (got - ref).abs().max() < _tol(x, wt) * 2
    assert torch.equal(obuf[:, 4:4 + cout], o2buf[:, 4:])
    assert (obuf[:, :4] == -7.0).all() and (obuf[:, 4 + cout:] == -7.0).all() and (o2buf[:, :4] == -7.0).all()
    # no addend, no bias; deterministic
    plain = ops.conv_wino1d(x.cuda(), ops.PackedWino1dX3(wt.cuda()), ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda'))
    assert (plain.cpu().double() - _ref_conv(x, wt, None, None)).abs().max() < _tol(x, wt) * 2
    assert torch.equal(ops.conv_wino1d(x.cuda(), ops.PackedWino1dX3(wt.cuda()), ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda')), plain)


def test_x3_winograd_1d_rejects_what_it_cannot_do(rpe):
    from rpe_amd import ops
    with pytest.raises(rpe.RpeError):
        ops.PackedWino1dX3(torch.zeros(8, 24, 1, 5, device='cuda'))            # cin % 16
    with pytest.raises(rpe.RpeError):
        ops.PackedWino1dX3(torch.zeros(8, 16, 3, 3, device='cuda'))
    pw = ops.PackedWino1dX3(torch.zeros(8, 16, 1, 5, device='cuda'))
    with pytest.raises(rpe.RpeError, match='UNSUPPORTED'):                      # width not a multiple of 4
        ops.conv_wino1d(torch.zeros(1, 16, 8, 10, device='cuda'), pw, ops.CONV_LINEAR, torch.empty(1, 8, 8, 10, device='cuda'))
    shifted = torch.zeros(16 * 8 * 12 + 1, device='cuda')[1:].view(1, 16, 8, 12)         # an input that is not 16-byte aligned
    with pytest.raises(rpe.RpeError, match='UNSUPPORTED'):
        ops.conv_wino1d(shifted, pw, ops.CONV_LINEAR, torch.empty(1, 8, 8, 12, device='cuda'))
    with pytest.raises(rpe.RpeError, match='UNSUPPORTED'):                      # plain and gate epilogues only
        ops.conv_wino1d(torch.zeros(1, 16, 8, 12, device='cuda'), pw, ops.CONV_LINEAR, torch.empty(1, 8, 8, 12, device='cuda'), scale=torch.ones(8, device='cuda'))
    assert not ops.PackedWino1dX3.supported(torch.zeros(8, 16, 1, 5), 10) and not ops.PackedWino1dX3.supported(torch.zeros(8, 24, 5, 1), 12)
    assert ops.PackedWino1dX3.supported(torch.zeros(8, 16, 5, 1), 12)


@pytest.mark.parametrize('kind,cin,cout,h,w', [('1x5', 256, 256, 64, 80), ('5x1', 256, 128, 64, 80), ('1x5', 256, 128, 44, 48)])
@pytest.mark.parametrize('flow_channels', [False, True])
def test_x3_winograd_1d_error_relative_to_the_f32_kernels_on_trained_like_statistics(rpe, kind, cin, cout, h, w, flow_channels):
    """The f32-equivalence bar of the 3x3 variant, for F(4,5): on trained-like statistics (a saturated hidden state, +-50 px flow
    channels, heavy-tailed weights) the error against the f64 convolution is at most 1.25x (RMS) the f32 Winograd kernel's
    (rpe_conv_wino1d, measured 0.85x), its maximum at most 1.25x the larger of the two f32 kernels' maxima, and within the bars
    rpe_conv_wino1d itself is held to against the direct kernel (3x RMS, 4x max)."""
    from rpe_amd import ops
    rng = np.random.default_rng(cin * 1000 + cout * 7 + h + 3 * len(kind) + ord(kind[0]) + int(flow_channels))
    kh, kw = {'1x5': (1, 5), '5x1': (5, 1)}[kind]
    b = 2
    x, wt, bias = _trained_like(rng, b, cin, cout, kh, kw, h, w, flow_channels)
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=(kh // 2, kw // 2))
    direct = ops.conv_fused(x.cuda(), ops.PackedConv(wt.cuda(), bias.cuda()), ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda'))
    wino = ops.conv_wino1d(x.cuda(), ops.PackedWino1d(wt.cuda(), bias.cuda()), ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda'))
    x3 = ops.conv_wino1d(x.cuda(), ops.PackedWino1dX3(wt.cuda(), bias.cuda()), ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda'))
    dmax, drms = _errs(direct, ref)
    wmax, wrms = _errs(wino, ref)
    xmax, xrms = _errs(x3, ref)
    print(f'{kind} {cin}->{cout} flow_channels={flow_channels}: |out| {float(ref.abs().max()):.1f}; direct max {dmax:.2e} rms {drms:.2e}; f32 winograd max {wmax:.2e} '
          f'rms {wrms:.2e}; bf16x3 winograd max {xmax:.2e} rms {xrms:.2e}; x3 / f32-winograd {xmax / wmax:.2f} / {xrms / wrms:.2f}; x3 / direct {xmax / dmax:.2f} / {xrms / drms:.2f}')
    assert xrms <= 1.25 * wrms and xmax <= 1.25 * max(wmax, dmax)
    assert xrms <= 3.0 * drms and xmax <= 4.0 * dmax


def test_x3_gru_gates_on_saturated_state_and_large_flow(rpe):
    """test_gpu_conv.test_gru_gates_on_saturated_state_and_large_flow on the variant (winograd bars): hidden state at +-1, flow channels at
    +-50, pre-activations over +-30 -- saturated gates are 0 or 1 to rounding, nothing is NaN, the blend stays inside [-1, 1]."""
    from rpe_amd import ops
    c, b, h, w = 128, 2, 64, 80
    rng = np.random.default_rng(77)
    hx, wzr, _ = _trained_like(rng, b, 2 * c, 2 * c, 1, 5, h, w)
    hx[:, :c] = torch.sign(hx[:, :c] - 1.0) * (1 - 1e-4 * torch.rand(b, c, h, w))
    _, wq, _ = _trained_like(rng, b, 2 * c, c, 1, 5, h, w)
    azr = torch.from_numpy(rng.normal(0, 10.0, size=(b, 2 * c, h, w)).astype(np.float32))
    aq = torch.from_numpy(rng.normal(0, 10.0, size=(b, c, h, w)).astype(np.float32))
    hid = hx[:, :c].double()
    pre = _ref_conv(hx, wzr, None, azr)
    z = torch.sigmoid(pre)[:, :c]
    g_hx, g_rhx = hx.cuda(), hx.cuda().clone()
    g_z = torch.empty(b, c, h, w, device='cuda')
    ops.conv_wino1d(g_hx, ops.PackedWino1dX3(wzr.cuda()), ops.CONV_GATE_ZR, g_z, out2=g_rhx[:, :c], add=azr.cuda(), hidden=g_hx[:, :c], gate_channels=c)
    assert bool(torch.isfinite(g_z).all()) and float(g_z.min()) >= 0.0 and float(g_z.max()) <= 1.0
    assert float((g_z.cpu().double() - z).abs().max()) < 0.25 * 2 * _tol(hx, wzr) + 2e-6
    assert float(pre.abs().max()) > 25 and float((z < 1e-9).float().mean()) > 0.001 and float((z > 1 - 1e-9).float().mean()) > 0.001
    rh = g_rhx[:, :c].cpu()
    q = torch.tanh(_ref_conv(torch.cat((rh, hx[:, c:]), 1), wq, None, aq))
    ops.conv_wino1d(g_rhx, ops.PackedWino1dX3(wq.cuda()), ops.CONV_GATE_H, g_hx[:, :c], add=aq.cuda(), hidden=g_hx[:, :c], zgate=g_z)
    hnew = (1 - g_z.cpu().double()) * hid + g_z.cpu().double() * q
    assert bool(torch.isfinite(g_hx).all()) and float(g_hx[:, :c].abs().max()) <= 1.0 + 1e-6
    assert float((g_hx[:, :c].cpu().double() - hnew).abs().max()) < _tol(hx, wq) * 2 + 4e-6


@pytest.mark.parametrize('kh,kw', [(1, 5), (5, 1)])
def test_x3_gates_with_the_gate_boundary_inside_a_channel_half(rpe, kh, kw):
    """48 hidden channels: z | r = 96 output channels, the boundary between z and r * h falls inside the kernel's first 64-channel half
    (the hidden state must be fetched for a half that holds both kinds) and the tile's second half holds 32 channels."""
    from rpe_amd import ops
    c, b, h, w = 48, 2, 20, 24
    rng = np.random.default_rng(kh + 3)
    hx = _rand(rng, b, 2 * c, h, w, s=0.5)
    wzr, azr = _rand(rng, 2 * c, 2 * c, kh, kw, s=0.05), _rand(rng, b, 2 * c, h, w, s=0.3)
    wq, aq = _rand(rng, c, 2 * c, kh, kw, s=0.05), _rand(rng, b, c, h, w, s=0.3)
    hid = hx[:, :c].double()
    zr = torch.sigmoid(_ref_conv(hx, wzr, None, azr))
    z, r = zr[:, :c], zr[:, c:]
    g_hx, g_rhx = hx.cuda(), hx.cuda().clone()
    g_z = torch.empty(b, c, h, w, device='cuda')
    ops.conv_wino1d(g_hx, ops.PackedWino1dX3(wzr.cuda()), ops.CONV_GATE_ZR, g_z, out2=g_rhx[:, :c], add=azr.cuda(), hidden=g_hx[:, :c], gate_channels=c)
    tz = _tol(hx, wzr) * 0.25 + 2e-7
    assert (g_z.cpu().double() - z).abs().max() < tz
    assert (g_rhx[:, :c].cpu().double() - r * hid).abs().max() < tz * float(hx.abs().max())
    q = torch.tanh(_ref_conv(torch.cat((g_rhx[:, :c].cpu(), hx[:, c:]), 1), wq, None, aq))
    ops.conv_wino1d(g_rhx, ops.PackedWino1dX3(wq.cuda()), ops.CONV_GATE_H, g_hx[:, :c], add=aq.cuda(), hidden=g_hx[:, :c], zgate=g_z)
    hnew = (1 - g_z.cpu().double()) * hid + g_z.cpu().double() * q
    assert (g_hx[:, :c].cpu().double() - hnew).abs().max() < _tol(hx, wq) + tz * 2
    assert torch.equal(g_hx[:, c:].cpu(), hx[:, c:])


def test_x3_kernels_agree_with_the_f32_kernels_on_random_shapes(rpe):
    """Sixty random shapes per kernel (channel counts around the 32 / 64 / 128 tile edges, maps from one tile to several ragged patches,
    every epilogue mode of the 1-D kernel): the variant against the f32 Winograd kernel on the same data, tolerance = twice the f64 bar of
    either (both are within it of the exact result).  Destinations are pre-filled: nothing outside them may change."""
    from rpe_amd import ops
    rng = np.random.default_rng(2025)
    for it in range(60):
        cin = 16 * int(rng.integers(1, 11)); cout = int(rng.choice([1, 7, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 130, 191, 256]))
        h, w = int(rng.integers(1, 41)), 4 * int(rng.integers(1, 14)); b = int(rng.integers(1, 4))
        kh, kw = ((1, 5), (5, 1))[it & 1]
        mode = (ops.CONV_LINEAR, ops.CONV_RELU, ops.CONV_GATE_ZR, ops.CONV_GATE_H)[(it >> 1) & 3]
        if mode == ops.CONV_GATE_ZR:
            cout = 2 * max(1, cout // 2)
        c = cout // 2 if mode == ops.CONV_GATE_ZR else cout
        x, wt = _rand(rng, b, cin, h, w).cuda(), _rand(rng, cout, cin, kh, kw, s=0.05).cuda()
        add, hid, z = _rand(rng, b, cout, h, w, s=0.3).cuda(), _rand(rng, b, c, h, w, s=0.5).cuda(), torch.rand(b, c, h, w).cuda()
        outs = []
        for P in (ops.PackedWino1d, ops.PackedWino1dX3):
            o = torch.full((b, c + 2, h, w), -7.0, device='cuda'); o2 = torch.full((b, c + 2, h, w), -7.0, device='cuda')
            kw_ = dict(add=add)
            if mode == ops.CONV_GATE_ZR:
                kw_.update(out2=o2[:, 1:1 + c], hidden=hid, gate_channels=c)
            elif mode == ops.CONV_GATE_H:
                kw_.update(hidden=hid, zgate=z)
            ops.conv_wino1d(x, P(wt), mode, o[:, 1:1 + c], **kw_)
            assert bool((o[:, 0] == -7.0).all()) and bool((o[:, 1 + c] == -7.0).all()), (it, cin, cout, h, w, kh, mode)
            outs.append((o, o2))
        tol = 4 * _tol(x.cpu(), wt.cpu())
        assert float((outs[0][0] - outs[1][0]).abs().max()) < tol, (it, cin, cout, h, w, kh, mode)
        if mode == ops.CONV_GATE_ZR:
            assert float((outs[0][1] - outs[1][1]).abs().max()) < tol, (it, cin, cout, h, w, kh, mode)
    for it in range(60):
        cin = 16 * int(rng.integers(1, 11)); cout = int(rng.choice([1, 7, 31, 32, 33, 63, 64, 65, 96, 126, 128, 129, 192, 256]))
        h, w = 2 * int(rng.integers(1, 21)), 4 * int(rng.integers(1, 14)); b = int(rng.integers(1, 4))
        x, wt, bias = _rand(rng, b, cin, h, w).cuda(), _rand(rng, cout, cin, 3, 3, s=0.05).cuda(), _rand(rng, cout, s=0.5).cuda()
        outs = []
        for P in (ops.PackedWino, ops.PackedWinoX3):
            o = torch.full((b, cout + 2, h, w), -7.0, device='cuda')
            ops.conv_wino(x, P(wt, bias), ops.CONV_RELU if it & 1 else ops.CONV_LINEAR, o[:, 1:1 + cout])
            assert bool((o[:, 0] == -7.0).all()) and bool((o[:, 1 + cout] == -7.0).all()), (it, cin, cout, h, w)
            outs.append(o)
        assert float((outs[0] - outs[1]).abs().max()) < 6 * _tol(x.cpu(), wt.cpu()), (it, cin, cout, h, w)


# ---- rpe_conv1x1_x3: 1x1 layers as bf16x3 GEMMs (csrc/conv1x1_x3.hip), the variant of rpe_conv1x1 ------------------------------------------

@pytest.mark.parametrize('cin,cout,h,w,b,relu', [(324, 256, 64, 80, 2, True), (128, 256, 64, 80, 3, False), (256, 576, 32, 40, 1, False),
                                                 (16, 128, 16, 16, 2, True), (40, 96, 20, 28, 1, False), (324, 130, 6, 10, 3, True), (8, 5, 2, 2, 1, False)])
def test_x3_conv1x1_matches_f64(rpe, cin, cout, h, w, b, relu):
    """convc1's shape (324 input channels: the last step is ragged), the encoders' and the mask head's output layers, channel counts that are
    not multiples of the 128-channel tile, maps that are not whole 256-pixel runs (down to one quad): against f64 at the f32 kernel's bar,
    on channel slices, with the second destination, through the prepared launcher; neighbours untouched."""
    from rpe_amd import ops
    rng = np.random.default_rng(cin + cout + h)
    x, wt, bias = _rand(rng, b, cin, h, w), _rand(rng, cout, cin, 1, 1, s=0.05), _rand(rng, cout, s=0.5)
    ref = F.conv2d(x.double(), wt.double(), bias.double())
    ref = ref.clamp_min(0) if relu else ref
    mode = ops.CONV_RELU if relu else ops.CONV_LINEAR
    xbuf = torch.full((b, cin + 8, h, w), float('nan'), device='cuda'); xbuf[:, 4:4 + cin] = x.cuda()
    obuf = torch.full((b, cout + 8, h, w), -7.0, device='cuda'); o2buf = torch.full((b, cout + 4, h, w), -7.0, device='cuda')
    px = ops.PackedConv1x1X3(wt.cuda(), bias.cuda())
    ops.conv1x1(xbuf[:, 4:4 + cin], px, mode, obuf[:, 4:4 + cout], out2=o2buf[:, 4:])
    assert (obuf[:, 4:4 + cout].cpu().double() - ref).abs().max() < _tol(x, wt)
    assert torch.equal(obuf[:, 4:4 + cout], o2buf[:, 4:])
    assert (obuf[:, :4] == -7.0).all() and (obuf[:, 4 + cout:] == -7.0).all() and (o2buf[:, :4] == -7.0).all()
    again = torch.empty(b, cout, h, w, device='cuda')
    ops.conv1x1(x.cuda(), px, mode, again, prepare=True)()
    assert torch.equal(again, obuf[:, 4:4 + cout])
    # against the f32 kernel on the same data
    f32 = ops.conv1x1(x.cuda(), ops.PackedConv1x1(wt.cuda(), bias.cuda()), mode, torch.empty(b, cout, h, w, device='cuda'))
    assert (f32 - again).abs().max() < 2 * _tol(x, wt)


def test_x3_conv1x1_rejects_what_it_cannot_do_and_error_bars(rpe):
    from rpe_amd import ops
    px = ops.PackedConv1x1X3(torch.zeros(8, 16, 1, 1, device='cuda'))
    with pytest.raises(rpe.RpeError, match='UNSUPPORTED'):                      # tanh stays on rpe_conv1x1
        ops.conv1x1(torch.zeros(1, 16, 4, 4, device='cuda'), px, ops.CONV_TANH, torch.empty(1, 8, 4, 4, device='cuda'))
    with pytest.raises(rpe.RpeError, match='UNSUPPORTED'):                      # planes that are not whole 16-byte quads
        ops.conv1x1(torch.zeros(1, 16, 3, 5, device='cuda'), px, ops.CONV_LINEAR, torch.empty(1, 8, 3, 5, device='cuda'))
    with pytest.raises(rpe.RpeError):
        ops.PackedConv1x1X3(torch.zeros(8, 16, 3, 3, device='cuda'))
    # trained-like statistics (heavy-tailed weights, post-ReLU activations): error against f64 at most 1.25x the f32 kernel's (RMS and max)
    rng = np.random.default_rng(11)
    for cin, cout in ((324, 256), (128, 256)):
        x, wt, bias = _trained_like(rng, 2, cin, cout, 1, 1, 64, 80)
        ref = F.conv2d(x.double(), wt.double(), bias.double())
        f32 = ops.conv1x1(x.cuda(), ops.PackedConv1x1(wt.cuda(), bias.cuda()), ops.CONV_LINEAR, torch.empty(2, cout, 64, 80, device='cuda'))
        x3 = ops.conv1x1(x.cuda(), ops.PackedConv1x1X3(wt.cuda(), bias.cuda()), ops.CONV_LINEAR, torch.empty(2, cout, 64, 80, device='cuda'))
        fmax, frms = _errs(f32, ref)
        xmax, xrms = _errs(x3, ref)
        print(f'1x1 {cin}->{cout}: f32 max {fmax:.2e} rms {frms:.2e}; bf16x3 max {xmax:.2e} rms {xrms:.2e}; ratios {xmax / fmax:.2f} / {xrms / frms:.2f}')
        assert xrms <= 1.25 * frms and xmax <= 1.25 * fmax


def test_x3_conv1x1_special_values_as_documented(rpe):
    """include/rpe.h (rpe_conv1x1_x3, SPECIAL VALUES) at convc1's ragged shape (cin = 324: the padded last K step reads channel 323 a second
    time against zero weights): a NaN poisons exactly the pixel it sits in -- in channel 323 too --, an Inf gives a non-finite value in exactly
    its own pixel (NaN where the f32 kernel gives Inf: the documented deviation of the split), every other output equals the clean run bit
    for bit, and subnormal-scale / large finite data comes through at the f32 kernel's accuracy."""
    from rpe_amd import ops
    rng = np.random.default_rng(17)
    cin, cout, h, w = 324, 256, 16, 20
    x, wt, bias = _rand(rng, 2, cin, h, w), _rand(rng, cout, cin, 1, 1, s=0.05), _rand(rng, cout, s=0.5)
    px, pf = ops.PackedConv1x1X3(wt.cuda(), bias.cuda()), ops.PackedConv1x1(wt.cuda(), bias.cuda())
    run = lambda t, p: ops.conv1x1(t.cuda(), p, ops.CONV_LINEAR, torch.empty(2, cout, h, w, device='cuda')).cpu()
    clean = run(x, px)
    y = x.clone()
    y[0, 323, 3, 4] = float('nan'); y[0, 7, 9, 9] = float('nan'); y[1, 323, 5, 6] = float('inf'); y[1, 100, 0, 0] = float('-inf')
    got, f32 = run(y, px), run(y, pf)
    hit = torch.zeros(2, h, w, dtype=torch.bool)
    hit[0, 3, 4] = hit[0, 9, 9] = hit[1, 5, 6] = hit[1, 0, 0] = True
    bad = ~torch.isfinite(got)
    assert bad.all(1)[hit].all() and not bad.any(1)[~hit].any()                          # exactly the four pixels, every output channel of them
    assert torch.equal(got.permute(0, 2, 3, 1)[~hit], clean.permute(0, 2, 3, 1)[~hit])   # everything else untouched, bit for bit
    assert torch.isnan(got[0, :, 3, 4]).all() and torch.isnan(f32[0, :, 3, 4]).all()     # NaN in, NaN out: both kernels
    assert torch.isinf(f32[1, :, 5, 6]).all() and torch.isnan(got[1, :, 5, 6]).all()     # the documented deviation: Inf -> NaN under the split
    for scale in (1e-30, 1e30):
        xs = x * scale
        ref = F.conv2d(xs.double(), wt.double(), bias.double() * (0.0 if scale < 1 else 1.0))
        g = ops.conv1x1(xs.cuda(), ops.PackedConv1x1X3(wt.cuda(), (bias * (0.0 if scale < 1 else 1.0)).cuda()), ops.CONV_LINEAR,
                        torch.empty(2, cout, h, w, device='cuda')).cpu().double()
        assert (g - ref).abs().max() < 1e-5 * ref.abs().max(), scale
