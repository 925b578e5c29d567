"""GPU: the fused implicit-GEMM convolutions (rpe_conv_fused) against a float64 CPU evaluation of the same
convolution + epilogue (torch.nn.functional.conv2d in f64 is the reference arithmetic: the f32 result of any
summation order must sit within a few f32 roundings of it).  Tolerance: 3e-6 * sqrt(K) * max|x| * max|w| per
accumulated term, K = cin*kh*kw -- about ten times the error of an f32 dot product of that length."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(rng, *shape, s=1.0):
    return torch.from_numpy((rng.normal(size=shape) * s).astype(np.float32))


def _ref_conv(x, w, bias, add):
    v = F.conv2d(x.double(), w.double(), None, padding=(w.shape[2] // 2, w.shape[3] // 2))
    if add is not None:
        v = v + add.double()
    if bias is not None:
        v = v + bias.double()[None, :, None, None]
    return v


def _tol(x, w):
    k = w.shape[1] * w.shape[2] * w.shape[3]
    return 3e-6 * np.sqrt(k) * float(x.abs().max()) * float(w.abs().max()) + 1e-6


@pytest.mark.parametrize('cin,cout,kh,kw,h,w,b', [
    (256, 192, 3, 3, 64, 80, 2),      # convc2: 64-row tiles, three of them
    (256, 126, 3, 3, 32, 40, 2),      # conv: ragged output channels
    (128, 64, 3, 3, 44, 48, 1),       # convf2 on a map that is not a multiple of the pixel tile
    (324, 256, 1, 1, 32, 40, 2),      # convc1: input channels not a multiple of 16
    (128, 256, 3, 3, 44, 48, 2),      # flow head conv1: 128-row tiles, ragged pixel tile
    (32, 128, 1, 5, 8, 12, 1),        # tiny map: one partial tile, rows shorter than the halo logic's stride
    (16, 128, 7, 3, 20, 16, 1),       # tall kernel
])
@pytest.mark.parametrize('relu', [False, True])
def test_conv_bias_act_matches_f64(rpe, cin, cout, kh, kw, h, w, b, relu):
    from rpe_amd import ops
    rng = np.random.default_rng(cin + cout + kh * 7 + kw)
    x, wt, bias = _rand(rng, b, cin, h, w), _rand(rng, cout, cin, kh, kw, s=0.05), _rand(rng, cout, s=0.1)
    ref = _ref_conv(x, wt, bias, None)
    if relu:
        ref = ref.clamp_min(0)
    # input and outputs are channel slices of wider buffers, as in the update block
    xbuf = torch.full((b, cin + 7, h, w), float('nan'), device='cuda'); xbuf[:, 3:3 + cin] = x.cuda()
    obuf = torch.full((b, cout + 5, h, w), -7.0, device='cuda'); o2buf = torch.full((b, cout + 2, h, w), -7.0, device='cuda')
    pc = ops.PackedConv(wt.cuda(), bias.cuda())
    ops.conv_fused(xbuf[:, 3:3 + cin], pc, ops.CONV_RELU if relu else ops.CONV_LINEAR, obuf[:, 4:4 + cout], out2=o2buf[:, 2:])
    got = obuf[:, 4:4 + cout].cpu().double()
    assert (got - ref).abs().max() < _tol(x, wt)
    assert torch.equal(obuf[:, 4:4 + cout], o2buf[:, 2:])
    assert (obuf[:, :4] == -7.0).all() and (obuf[:, 4 + cout:] == -7.0).all()            # neighbours untouched


@pytest.mark.parametrize('impl', ['direct', 'winograd'])
@pytest.mark.parametrize('kh,kw,h,w', [(1, 5, 64, 80), (5, 1, 64, 80), (1, 5, 44, 48), (5, 1, 44, 48), (1, 5, 30, 40), (5, 1, 30, 40)])
def test_gru_half_step_matches_f64(rpe, kh, kw, h, w, impl):
    """z, r*h and the blended hidden state of one SepConvGRU half, fused into the two convolutions, against f64: the direct
    implicit GEMM (rpe_conv_fused) and Winograd F(4,5) along the filter axis (rpe_conv_wino1d), same tolerances."""
    from rpe_amd import ops
    Packed, conv = (ops.PackedConv, ops.conv_fused) if impl == 'direct' else (ops.PackedWino1d, ops.conv_wino1d)
    c, b = 128, 2
    rng = np.random.default_rng(kh * 10 + kw + h)
    hx = _rand(rng, b, 2 * c, h, w, s=0.5)
    wzr, bzr, azr = _rand(rng, 2 * c, 2 * c, kh, kw, s=0.03), _rand(rng, 2 * c, s=0.1), _rand(rng, b, 2 * c, h, w, s=0.3)
    wq, bq, aq = _rand(rng, c, 2 * c, kh, kw, s=0.03), _rand(rng, c, s=0.1), _rand(rng, b, c, h, w, s=0.3)
    hid = hx[:, :c].double()
    zr = torch.sigmoid(_ref_conv(hx, wzr, bzr, azr))
    z, r = zr[:, :c], zr[:, c:]
    rhx = torch.cat((r * hid, hx[:, c:].double()), 1)
    q = torch.tanh(_ref_conv(rhx.float(), wq, bq, aq))      # the q convolution reads the f32 r*h the kernel stored
    hnew = (1 - z) * hid + z * q

    g_hx, g_rhx = hx.cuda(), hx.cuda().clone()
    g_z = torch.empty(b, c, h, w, device='cuda')
    pzr, pq = Packed(wzr.cuda(), bzr.cuda()), Packed(wq.cuda(), bq.cuda())
    conv(g_hx, pzr, ops.CONV_GATE_ZR, g_z, out2=g_rhx[:, :c], add=azr.cuda(), hidden=g_hx[:, :c], gate_channels=c)
    tz = _tol(hx, wzr) * 0.25 + 2e-7
    assert (g_z.cpu().double() - z).abs().max() < tz
    assert (g_rhx[:, :c].cpu().double() - r * hid).abs().max() < tz * float(hx.abs().max())
    assert torch.equal(g_rhx[:, c:], g_hx[:, c:])
    conv(g_rhx, pq, ops.CONV_GATE_H, g_hx[:, :c], add=aq.cuda(), hidden=g_hx[:, :c], zgate=g_z)   # in place on h
    assert (g_hx[:, :c].cpu().double() - hnew).abs().max() < _tol(hx, wq) + tz * 2
    assert torch.equal(g_hx[:, c:].cpu(), hx[:, c:])


@pytest.mark.parametrize('cin,cout,kh,kw,h,w,b', [
    (256, 256, 1, 5, 64, 80, 2),      # convz1|convr1 at bench geometry
    (256, 128, 5, 1, 64, 80, 2),      # convq2
    (64, 96, 1, 5, 20, 24, 1),        # ragged output channels (96 = 64 + 32), map smaller than two tiles
    (64, 70, 5, 1, 7, 36, 2),         # 7 rows: the last 4-pixel tile is cut, 36 columns: the last 16-column block is cut
    (8, 16, 1, 5, 33, 4, 1),          # one quad wide
    (8, 16, 5, 1, 3, 4, 1),           # shorter than the filter
])
@pytest.mark.parametrize('relu', [False, True])
def test_winograd_1d_matches_f64(rpe, cin, cout, kh, kw, h, w, b, relu):
    """rpe_conv_wino1d (F(4,5) along the filter axis) with the plain epilogues, on channel slices of wider buffers."""
    from rpe_amd import ops
    rng = np.random.default_rng(cin + cout + kh * 7 + kw + h)
    x, wt, bias, add = _rand(rng, b, cin, h, w), _rand(rng, cout, cin, kh, kw, s=0.05), _rand(rng, cout, s=0.1), _rand(rng, b, cout, h, w, s=0.3)
    ref = _ref_conv(x, wt, bias, add)
    if relu:
        ref = ref.clamp_min(0)
    xbuf = torch.full((b, cin + 8, h, w), float('nan'), device='cuda'); xbuf[:, 4:4 + cin] = x.cuda()
    obuf = torch.full((b, cout + 8, h, w), -7.0, device='cuda'); o2buf = torch.full((b, cout + 4, h, w), -7.0, device='cuda')
    pw = ops.PackedWino1d(wt.cuda(), bias.cuda())
    ops.conv_wino1d(xbuf[:, 4:4 + cin], pw, ops.CONV_RELU if relu else ops.CONV_LINEAR, obuf[:, 4:4 + cout], out2=o2buf[:, 4:], add=add.cuda())
    got = obuf[:, 4:4 + cout].cpu().double()
    assert (got - ref).abs().max() < _tol(x, wt) * 2                       # (the transforms' constants reach 5.25)
    assert torch.equal(obuf[:, 4:4 + cout], o2buf[:, 4:])
    assert (obuf[:, :4] == -7.0).all() and (obuf[:, 4 + cout:] == -7.0).all()            # neighbours untouched


@pytest.mark.parametrize('cin,cout,h,w,b', [(128, 128, 64, 80, 32), (128, 128, 64, 80, 1), (128, 576, 44, 48, 2)])
def test_1x1_output_layers_and_tanh_epilogue(rpe, cin, cout, h, w, b):
    """The encoders' / mask head's 1x1 output layers on rpe_conv_fused, and RPE_CONV_TANH (the hidden-state half of the context
    encoder: net = tanh(net)) on the hardware exp / rcp path: absolute 1e-6 on tanh of pre-activations spread over +-8."""
    from rpe_amd import ops
    rng = np.random.default_rng(cin + cout + b)
    x, wt, bias = _rand(rng, b, cin, h, w, s=2.0), _rand(rng, cout, cin, 1, 1, s=0.1), _rand(rng, cout, s=0.5)
    ref = _ref_conv(x, wt, bias, None)
    pc = ops.PackedConv(wt.cuda(), bias.cuda())
    buf = torch.full((b, cout + 3, h, w), -7.0, device='cuda')
    ops.conv_fused(x.cuda(), pc, ops.CONV_TANH, buf[:, 1:1 + cout])
    assert float(ref.abs().max()) > 6.0
    assert float((buf[:, 1:1 + cout].cpu().double() - torch.tanh(ref)).abs().max()) < _tol(x, wt) + 1e-6
    assert bool((buf[:, 0] == -7.0).all()) and bool((buf[:, 1 + cout:] == -7.0).all())
    lin = ops.conv_fused(x.cuda(), pc, ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda'))
    assert float((lin.cpu().double() - ref).abs().max()) < _tol(x, wt)
    with pytest.raises(rpe.RpeError, match='UNSUPPORTED'):                      # plain epilogue only
        ops.conv_fused(x.cuda(), pc, ops.CONV_TANH, torch.empty(b, cout, h, w, device='cuda'), scale=torch.ones(cout, device='cuda'))


def test_winograd_1d_rejects_what_it_cannot_do(rpe):
    from rpe_amd import ops
    with pytest.raises(rpe.RpeError):
        ops.PackedWino1d(torch.zeros(8, 16, 3, 3, device='cuda'))
    pw = ops.PackedWino1d(torch.zeros(8, 16, 1, 5, device='cuda'))
    with pytest.raises(rpe.RpeError, match='UNSUPPORTED'):                      # width not a multiple of 4
        ops.conv_wino1d(torch.zeros(1, 16, 8, 10, device='cuda'), pw, ops.CONV_LINEAR, torch.empty(1, 8, 8, 10, device='cuda'))
    assert not ops.PackedWino1d.supported(torch.zeros(8, 16, 1, 5), 10) and ops.PackedWino1d.supported(torch.zeros(8, 16, 5, 1), 12)


def test_fused_gates_agree_with_the_separate_kernels(rpe):
    """Same inputs through a plain convolution + rpe_gru_gates_* (the stand-alone gate kernels of the generic route, used when the map width is not a multiple of 4)."""
    from rpe_amd import ops
    c, b, h, w = 128, 1, 32, 40
    rng = np.random.default_rng(5)
    hx = _rand(rng, b, 2 * c, h, w, s=0.5).cuda()
    wzr, bzr, azr = _rand(rng, 2 * c, 2 * c, 1, 5, s=0.03).cuda(), _rand(rng, 2 * c, s=0.1).cuda(), _rand(rng, b, 2 * c, h, w, s=0.3).cuda()
    z1, rh1 = torch.empty(b, c, h, w, device='cuda'), hx.clone()
    ops.gru_gates_zr(F.conv2d(hx, wzr, None, padding=(0, 2)), hx, c, z1, rh1, bias=bzr, add=azr)
    z2, rh2 = torch.empty_like(z1), hx.clone()
    ops.conv_fused(hx, ops.PackedConv(wzr, bzr), ops.CONV_GATE_ZR, z2, out2=rh2[:, :c], add=azr, hidden=hx[:, :c], gate_channels=c)
    assert (z1 - z2).abs().max() < 2e-5 and (rh1 - rh2).abs().max() < 4e-5


def test_unsupported_shapes_are_refused(rpe):
    from rpe_amd import ops
    x = torch.zeros(1, 16, 8, 10, device='cuda')                 # width not a multiple of 4
    pc = ops.PackedConv(torch.zeros(8, 16, 3, 3, device='cuda'))
    with pytest.raises(rpe.RpeError, match='UNSUPPORTED'):
        ops.conv_fused(x, pc, ops.CONV_LINEAR, torch.empty(1, 8, 8, 10, device='cuda'))
    assert not ops.PackedConv.supported(torch.zeros(8, 16, 3, 3), 10) and ops.PackedConv.supported(torch.zeros(8, 16, 3, 3), 12)
    with pytest.raises(rpe.RpeError):
        ops.conv_fused(torch.zeros(1, 8, 8, 12, device='cuda'), pc, ops.CONV_LINEAR, torch.empty(1, 8, 8, 12, device='cuda'))


@pytest.mark.parametrize('c,h,w,b', [(64, 64, 80, 3), (96, 44, 48, 2), (128, 32, 40, 2)])
def test_encoder_block_epilogues_match_f64(rpe, c, h, w, b):
    """The two encoder epilogues: folded batch norm (scale, shift) + ReLU + residual + ReLU in the convolution, and
    instance norm from the partial sums the convolution leaves behind (rpe_instnorm_apply)."""
    from rpe_amd import ops
    rng = np.random.default_rng(c + h)
    x, wt, bias = _rand(rng, b, c, h, w), _rand(rng, c, c, 3, 3, s=0.05), _rand(rng, c, s=0.5)
    res = _rand(rng, b, c, h, w).abs()
    scale, shift = _rand(rng, c).abs() + 0.5, _rand(rng, c, s=0.3)
    pc = ops.PackedConv(wt.cuda(), bias.cuda())
    conv = F.conv2d(x.double(), wt.double(), None, padding=1)
    # cnet: y = max(res + max(conv*scale + shift, 0), 0)
    ref = (res.double() + (conv * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]).clamp_min(0)).clamp_min(0)
    got = ops.conv_fused(x.cuda(), pc, ops.CONV_RELU, torch.empty(b, c, h, w, device='cuda'), scale=scale.cuda(), bias=shift.cuda(),
                         residual=res.cuda())
    assert (got.cpu().double() - ref).abs().max() < _tol(x, wt) * 2.5
    # fnet: instance norm of conv + bias, ReLU, residual, ReLU
    pre = conv + bias.double()[None, :, None, None]
    mean, var = pre.mean((2, 3), keepdim=True), pre.var((2, 3), unbiased=False, keepdim=True)
    ref2 = (res.double() + ((pre - mean) / torch.sqrt(var + 1e-5)).clamp_min(0)).clamp_min(0)
    stats = ops.conv_stats_buffer(b, c, h, w, 'cuda')
    raw = ops.conv_fused(x.cuda(), pc, ops.CONV_LINEAR, torch.empty(b, c, h, w, device='cuda'), stats=stats)
    assert (raw.cpu().double() - pre).abs().max() < _tol(x, wt)
    st = stats.cpu().double()                                              # per-tile (count, mean, M2)
    assert float(st[..., 0].sum(-1).min()) == float(st[..., 0].sum(-1).max()) == h * w
    ssum = (st[..., 0] * st[..., 1]).sum(-1) / (h * w)
    assert (ssum - mean[:, :, 0, 0]).abs().max() < 1e-5
    got2 = ops.instnorm_apply(raw, stats, eps=1e-5, relu=True, residual=res.cuda())
    inv = float((1 / torch.sqrt(var + 1e-5)).max())
    assert (got2.cpu().double() - ref2).abs().max() < (_tol(x, wt) + 2e-6) * inv * 2
    # and against the three-pass kernel on the same raw tensor
    raw2 = ops.conv_fused(x.cuda(), pc, ops.CONV_LINEAR, torch.empty(b, c, h, w, device='cuda'))
    three = ops.instnorm_act(raw2, None, eps=1e-5, relu=True, residual=res.cuda())
    assert (three - got2).abs().max() < 2e-5 * inv


def test_encoder_epilogue_moments_with_addend_and_second_output(rpe):
    """The encoder epilogue of rpe_conv_fused is instantiated per shape (moments or not) x (residual / addend / second output or none of
    them): the combination the encoders themselves never use -- moments TOGETHER with an addend and a second output -- against f64:
    values, both outputs, and the (count, mean, M2) records of v = conv + addend + bias."""
    from rpe_amd import ops
    rng = np.random.default_rng(77)
    b, c, h, w = 2, 64, 36, 40
    x, wt, bias, add = _rand(rng, b, c, h, w), _rand(rng, c, c, 3, 3, s=0.05), _rand(rng, c, s=0.5), _rand(rng, b, c, h, w, s=0.4)
    pc = ops.PackedConv(wt.cuda(), bias.cuda())
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1) + add.double()
    stats = ops.conv_stats_buffer(b, c, h, w, 'cuda')
    o2 = torch.full((b, c + 3, h, w), -7.0, device='cuda')
    got = ops.conv_fused(x.cuda(), pc, ops.CONV_LINEAR, torch.empty(b, c, h, w, device='cuda'), out2=o2[:, 2:2 + c], add=add.cuda(), stats=stats)
    assert float((got.cpu().double() - ref).abs().max()) < _tol(x, wt) * 1.5
    assert torch.equal(o2[:, 2:2 + c], got) and bool((o2[:, :2] == -7.0).all()) and bool((o2[:, 2 + c:] == -7.0).all())
    st = stats.cpu().double()
    n = st[..., 0].sum(-1)
    assert float(n.min()) == float(n.max()) == h * w
    mean = (st[..., 0] * st[..., 1]).sum(-1) / n
    m2 = (st[..., 2] + st[..., 0] * (st[..., 1] - mean[..., None]) ** 2).sum(-1)
    assert float((mean - ref.mean((2, 3))).abs().max()) < 1e-5
    assert float((m2 / n - ref.var((2, 3), unbiased=False)).abs().max()) < 1e-4


@pytest.mark.parametrize('cin,cout,kh,kw,mode', [(256, 256, 1, 5, 'zr'), (256, 128, 5, 1, 'relu'), (256, 192, 3, 3, 'relu'), (324, 256, 1, 1, 'relu')])
def test_large_and_small_tiles_agree_bitwise(rpe, cin, cout, kh, kw, mode):
    """rpe_conv_fused picks 128x128 / 64x256 tiles for launches that fill the chip and 64x64 tiles for small ones
    (batch-1 tracking).  Every output element accumulates the same products in the same order either way, so a batch
    of 16 maps (large tiles) must equal the same maps convolved two at a time (small tiles) bit for bit."""
    from rpe_amd import ops
    rng = np.random.default_rng(cin + cout + kw)
    b, h, w, c = 16, 64, 80, 128
    x, wt, bias = _rand(rng, b, cin, h, w, s=0.5).cuda(), _rand(rng, cout, cin, kh, kw, s=0.05).cuda(), _rand(rng, cout, s=0.1).cuda()
    pc = ops.PackedConv(wt, bias)

    def run(xs):
        n = xs.shape[0]
        if mode == 'zr':
            z, rh = torch.empty(n, c, h, w, device='cuda'), torch.empty(n, c, h, w, device='cuda')
            ops.conv_fused(xs, pc, ops.CONV_GATE_ZR, z, out2=rh, hidden=xs[:, :c], gate_channels=c)
            return torch.cat((z, rh), 1)
        return ops.conv_fused(xs, pc, ops.CONV_RELU, torch.empty(n, cout, h, w, device='cuda'))
    big = run(x)
    for i in range(0, b, 8):
        assert torch.equal(big[i:i + 2], run(x[i:i + 2].contiguous()))
    ref = F.conv2d(x[:1], wt, bias, padding=(kh // 2, kw // 2))          # and sanity against the library on one map
    ref = torch.cat((torch.sigmoid(ref[:, :c]), torch.sigmoid(ref[:, c:]) * x[:1, :c]), 1) if mode == 'zr' else ref.clamp_min(0)
    assert (big[:1] - ref).abs().max() < 2e-4


@pytest.mark.parametrize('h,w', [(64, 80), (44, 48), (30, 36)])
def test_vertical_patch_tiles_match_library(rpe, h, w):
    """5x1 convolutions of launches that fill the chip use 16x8 pixel patches (taps as LDS row offsets); maps that are not
    a multiple of the patch exercise the zero-filled halo rows and the masked stores.  Reference: the library's f32
    convolution on the same GPU (tolerance of an f32 dot product of length 1280) and, bitwise, the small-tile kernel."""
    from rpe_amd import ops
    rng = np.random.default_rng(h * w)
    b, cin, cout = 24, 256, 256
    x, wt, bias = _rand(rng, b, cin, h, w, s=0.5).cuda(), _rand(rng, cout, cin, 5, 1, s=0.03).cuda(), _rand(rng, cout, s=0.1).cuda()
    pc = ops.PackedConv(wt, bias)
    out = torch.full((b, cout + 1, h, w), -3.0, device='cuda')
    ops.conv_fused(x, pc, ops.CONV_LINEAR, out[:, :cout])
    ref = F.conv2d(x, wt, bias, padding=(2, 0))
    assert (out[:, :cout] - ref).abs().max() < 3e-6 * np.sqrt(1280) * float(x.abs().max()) * float(wt.abs().max()) * 4
    assert (out[:, cout] == -3.0).all()
    one = ops.conv_fused(x[5:6].contiguous(), pc, ops.CONV_LINEAR, torch.empty(1, cout, h, w, device='cuda'))     # small tiles
    assert torch.equal(one, out[5:6, :cout])


@pytest.mark.parametrize('cin,cout,k,h,w,b', [(64, 96, 3, 64, 80, 3), (96, 128, 3, 44, 48, 2), (64, 96, 1, 64, 80, 2), (96, 128, 1, 36, 40, 3),
                                              (16, 32, 3, 12, 16, 1), (64, 96, 3, 128, 160, 14), (64, 96, 1, 128, 160, 14)])
def test_stride2_convolutions_match_f64(rpe, cin, cout, k, h, w, b):
    """The encoders' down-sampling convolutions: 3x3 stride 2 pad 1 and 1x1 stride 2, with both encoder epilogues
    (folded batch norm + ReLU; instance-norm partial sums + rpe_instnorm_apply).  The first five cases are launches of fewer than
    512 workgroups of 128 x 128 (64 x 64 tiles, one moment record per 64 pixels), the last two run on the 128 x 128 tiles."""
    from rpe_amd import ops
    rng = np.random.default_rng(cin + cout + k + h)
    x, wt, bias = _rand(rng, b, cin, h, w), _rand(rng, cout, cin, k, k, s=0.05), _rand(rng, cout, s=0.5)
    scale, shift = _rand(rng, cout).abs() + 0.5, _rand(rng, cout, s=0.3)
    pc = ops.PackedConv(wt.cuda(), bias.cuda())
    conv = F.conv2d(x.double(), wt.double(), None, stride=2, padding=k // 2)
    ho, wo = h // 2, w // 2
    assert conv.shape[-2:] == (ho, wo)
    ref = (conv * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]).clamp_min(0)
    got = ops.conv_fused(x.cuda(), pc, ops.CONV_RELU, torch.empty(b, cout, ho, wo, device='cuda'), scale=scale.cuda(), bias=shift.cuda(), stride=2)
    assert (got.cpu().double() - ref).abs().max() < _tol(x, wt) * 2.5
    pre = conv + bias.double()[None, :, None, None]
    mean, var = pre.mean((2, 3), keepdim=True), pre.var((2, 3), unbiased=False, keepdim=True)
    stats = ops.conv_stats_buffer(b, cout, h, w, 'cuda', stride=2)
    raw = ops.conv_fused(x.cuda(), pc, ops.CONV_LINEAR, torch.empty(b, cout, ho, wo, device='cuda'), stats=stats, stride=2)
    assert (raw.cpu().double() - pre).abs().max() < _tol(x, wt)
    got2 = ops.instnorm_apply(raw, stats, eps=1e-5, relu=False)
    inv = float((1 / torch.sqrt(var + 1e-5)).max())
    assert (got2.cpu().double() - (pre - mean) / torch.sqrt(var + 1e-5)).abs().max() < (_tol(x, wt) + 2e-6) * inv * 2


def test_stride2_tile_classes_agree_bitwise_including_statistics(rpe):
    """A stride-2 launch below 512 workgroups runs on 64 x 64 tiles: same products, same order, so a batch of 14 (128 x 128 tiles)
    equals the same maps run two at a time bit for bit -- and so do the instance-norm statistics: both classes leave one record per
    32 output pixels (rpe_conv_stats_tiles), hence an image's normalisation does not depend on the batch it was launched in."""
    from rpe_amd import ops
    rng = np.random.default_rng(99)
    b, cin, cout, h, w = 14, 64, 96, 128, 160
    x, wt, bias = _rand(rng, b, cin, h, w).cuda(), _rand(rng, cout, cin, 3, 3, s=0.05).cuda(), _rand(rng, cout, s=0.5).cuda()
    pc = ops.PackedConv(wt, bias)
    L = rpe.lib()
    assert L.rpe_conv_stats_tiles(cout, h, w, 2) == L.rpe_conv_stats_tiles_batch(cout, h, w, 2, b) == L.rpe_conv_stats_tiles_batch(cout, h, w, 2, 2) == 160
    big = ops.conv_fused(x, pc, ops.CONV_RELU, torch.empty(b, cout, h // 2, w // 2, device='cuda'), stride=2)
    st_big = ops.conv_stats_buffer(b, cout, h, w, 'cuda', stride=2)
    raw_big = ops.conv_fused(x, pc, ops.CONV_LINEAR, torch.empty(b, cout, h // 2, w // 2, device='cuda'), stats=st_big, stride=2)
    n_big = ops.instnorm_apply(raw_big, st_big, eps=1e-5, relu=True, out=torch.empty_like(raw_big))
    mi_big = ops.instnorm_finalize(st_big, (h // 2) * (w // 2), channels=cout)
    assert float(st_big[..., 0].sum(dim=2).min()) == float(st_big[..., 0].sum(dim=2).max()) == (h // 2) * (w // 2)
    for i in (0, 6, 12):
        xs = x[i:i + 2].contiguous()
        small = ops.conv_fused(xs, pc, ops.CONV_RELU, torch.empty(2, cout, h // 2, w // 2, device='cuda'), stride=2)
        assert torch.equal(big[i:i + 2], small)
        st = ops.conv_stats_buffer(2, cout, h, w, 'cuda', stride=2)
        raw = ops.conv_fused(xs, pc, ops.CONV_LINEAR, torch.empty(2, cout, h // 2, w // 2, device='cuda'), stats=st, stride=2)
        assert torch.equal(raw, raw_big[i:i + 2]) and torch.equal(st, st_big[i:i + 2])
        assert torch.equal(ops.instnorm_apply(raw, st, eps=1e-5, relu=True, out=torch.empty_like(raw)), n_big[i:i + 2])
        assert torch.equal(ops.instnorm_finalize(st, (h // 2) * (w // 2), channels=cout), mi_big[i:i + 2])
    # a plane that is not a whole number of 32-pixel blocks / 64-pixel tiles: 18 x 20 outputs = 360 = 11 blocks + 8 pixels
    x2 = _rand(rng, 3, cin, 36, 40).cuda()
    st2 = ops.conv_stats_buffer(3, cout, 36, 40, 'cuda', stride=2)
    assert st2.shape[2] == 12
    raw2 = ops.conv_fused(x2, pc, ops.CONV_LINEAR, torch.empty(3, cout, 18, 20, device='cuda'), stats=st2, stride=2)
    assert st2[..., 0].sum(dim=2).unique().tolist() == [360.0]
    mi2 = ops.instnorm_finalize(st2, 360, channels=cout).cpu().double()
    ref = raw2.cpu().double().reshape(3, cout, -1)
    assert float((mi2[..., 0] - ref.mean(-1)).abs().max()) < 1e-5
    assert float((mi2[..., 1] * torch.sqrt(ref.var(-1, unbiased=False) + 1e-5) - 1).abs().max()) < 1e-5
    with pytest.raises(rpe.RpeError):                        # a buffer of another record count is refused
        ops.conv_fused(x[:2].contiguous(), pc, ops.CONV_LINEAR, torch.empty(2, cout, h // 2, w // 2, device='cuda'),
                       stats=torch.empty(2, cout, 41, 3, device='cuda'), stride=2)


def test_stride2_refusals(rpe):
    from rpe_amd import ops
    pc = ops.PackedConv(torch.zeros(32, 16, 3, 3, device='cuda'))
    with pytest.raises(rpe.RpeError, match='UNSUPPORTED'):          # odd map height
        ops.conv_fused(torch.zeros(1, 16, 9, 12, device='cuda'), pc, ops.CONV_LINEAR, torch.empty(1, 32, 4, 6, device='cuda'), stride=2)
    pc5 = ops.PackedConv(torch.zeros(32, 16, 1, 5, device='cuda'))
    with pytest.raises(rpe.RpeError, match='UNSUPPORTED'):          # only 3x3 and 1x1 have a stride-2 kernel
        ops.conv_fused(torch.zeros(1, 16, 8, 12, device='cuda'), pc5, ops.CONV_LINEAR, torch.empty(1, 32, 4, 6, device='cuda'), stride=2)


@pytest.mark.parametrize('c,h,w,b', [(64, 32, 40, 2), (128, 22, 24, 3)])
def test_loader_side_instance_norm_matches_f64(rpe, c, h, w, b):
    """conv2 of a fnet residual block reads conv1's RAW output and normalises it (+ReLU) while staging it: the result must
    equal the convolution of the explicitly normalised tensor; zero padding applies after the normalisation."""
    from rpe_amd import ops
    rng = np.random.default_rng(c + h)
    x, w1, w2 = _rand(rng, b, c, h, w), _rand(rng, c, c, 3, 3, s=0.05), _rand(rng, c, c, 3, 3, s=0.05)
    b1, b2 = _rand(rng, c, s=0.5), _rand(rng, c, s=0.5)
    pre = F.conv2d(x.double(), w1.double(), b1.double(), padding=1)
    mean, var = pre.mean((2, 3), keepdim=True), pre.var((2, 3), unbiased=False, keepdim=True)
    y = ((pre - mean) / torch.sqrt(var + 1e-5)).clamp_min(0)
    ref = F.conv2d(y, w2.double(), b2.double(), padding=1)
    p1, p2 = ops.PackedConv(w1.cuda(), b1.cuda()), ops.PackedConv(w2.cuda(), b2.cuda())
    stats = ops.conv_stats_buffer(b, c, h, w, 'cuda')
    raw1 = ops.conv_fused(x.cuda(), p1, ops.CONV_LINEAR, torch.empty(b, c, h, w, device='cuda'), stats=stats)
    mi = ops.instnorm_finalize(stats, h * w, eps=1e-5)
    assert (mi[..., 0].cpu().double() - mean[:, :, 0, 0]).abs().max() < 1e-5
    assert ((mi[..., 1].cpu().double() - 1 / torch.sqrt(var + 1e-5)[:, :, 0, 0]).abs() * torch.sqrt(var + 1e-5)[:, :, 0, 0]).max() < 1e-5
    got = ops.conv_fused(raw1, p2, ops.CONV_LINEAR, torch.empty(b, c, h, w, device='cuda'), pre_norm=mi)
    inv = float((1 / torch.sqrt(var + 1e-5)).max())
    assert (got.cpu().double() - ref).abs().max() < _tol(y.float(), w2) + _tol(x, w1) * inv * float(w2.abs().sum((1, 2, 3)).max())


@pytest.mark.parametrize('h,w,b', [(64, 80, 2), (44, 72, 1), (352, 384, 1)])
def test_stem_conv_matches_f64(rpe, h, w, b):
    """The encoders' first layer on the raw 0..255 image: 7x7 stride 2 pad 3 of 2*(x/255)-1 (zero padding of the normalised
    image), with the folded-batch-norm + ReLU epilogue and with the instance-norm partial sums."""
    from rpe_amd import ops
    rng = np.random.default_rng(h + w)
    img = torch.from_numpy(rng.integers(0, 256, size=(b, 3, h, w)).astype(np.float32))
    wt, bias = _rand(rng, 64, 3, 7, 7, s=0.1), _rand(rng, 64, s=0.3)
    scale, shift = _rand(rng, 64).abs() + 0.5, _rand(rng, 64, s=0.3)
    xn = (2 * (img / 255.0) - 1.0)
    conv = F.conv2d(xn.double(), wt.double(), None, stride=2, padding=3)
    ps = ops.PackedStem(wt.cuda())
    tol = _tol(xn, wt) * 2
    got = ops.stem_conv(img.cuda(), ps, bias=shift.cuda(), scale=scale.cuda(), relu=True)
    ref = (conv * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]).clamp_min(0)
    assert got.shape == ref.shape and (got.cpu().double() - ref).abs().max() < tol * 2.5
    pre = conv + bias.double()[None, :, None, None]
    raw, stats = ops.stem_conv(img.cuda(), ps, bias=bias.cuda(), relu=False, stats=True)
    assert (raw.cpu().double() - pre).abs().max() < tol
    mean, var = pre.mean((2, 3), keepdim=True), pre.var((2, 3), unbiased=False, keepdim=True)
    out = ops.instnorm_apply(raw, stats, eps=1e-5, relu=True)
    inv = float((1 / torch.sqrt(var + 1e-5)).max())
    assert (out.cpu().double() - ((pre - mean) / torch.sqrt(var + 1e-5)).clamp_min(0)).abs().max() < (tol + 2e-6) * inv * 2
    # scale AND moments in one launch (no caller of this package does; the epilogue's table and moment code must still compose)
    both, st2 = ops.stem_conv(img.cuda(), ps, bias=shift.cuda(), scale=scale.cuda(), relu=False, stats=True)
    lin = conv * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    assert (both.cpu().double() - lin).abs().max() < tol * 2.5
    s2 = st2.cpu().double()
    assert float(s2[..., 0].sum(-1).min()) == float(s2[..., 0].sum(-1).max()) == lin.shape[-1] * lin.shape[-2]
    assert float(((s2[..., 0] * s2[..., 1]).sum(-1) / s2[..., 0].sum(-1) - lin.mean((2, 3))).abs().max()) < 1e-4


@pytest.mark.parametrize('h,w,b', [(64, 80, 3), (44, 48, 2), (13, 37, 1)])
def test_convf1_patch_kernel_matches_f64(rpe, h, w, b):
    """The motion encoder's 7x7 convolution of the two flow channels (stride 1, 128 outputs, bias + ReLU) on the patch-staged kernel."""
    from rpe_amd import ops
    rng = np.random.default_rng(h * w)
    flow, wt, bias = _rand(rng, b, 2, h, w, s=5.0), _rand(rng, 128, 2, 7, 7, s=0.1), _rand(rng, 128, s=0.3)
    ref = F.conv2d(flow.double(), wt.double(), bias.double(), padding=3).clamp_min(0)
    got = ops.stem_conv(flow.cuda(), ops.PackedStem(wt.cuda()), bias=bias.cuda(), relu=True, div=1.0, mul=1.0, sub=0.0)
    assert got.shape == ref.shape and (got.cpu().double() - ref).abs().max() < _tol(flow, wt)


def test_stem_small_and_large_launches_agree_bitwise(rpe):
    """Launches of fewer than 256 patches run the 7x7 kernels on 32-channel workgroups (twice the workgroups, half the chain each):
    same products, same order, same statistics records -- the same bits as inside a large batch."""
    from rpe_amd import ops
    rng = np.random.default_rng(77)
    # convf1: one frame pair (2 x 20 patches x 2 tiles of 64 = 80 < 256) against a batch of 8
    flow, wt, bias = _rand(rng, 8, 2, 64, 80, s=5.0).cuda(), _rand(rng, 128, 2, 7, 7, s=0.1).cuda(), _rand(rng, 128, s=0.3).cuda()
    ps = ops.PackedStem(wt)
    big = ops.stem_conv(flow, ps, bias=bias, relu=True, div=1.0, mul=1.0, sub=0.0)
    for sl in (slice(0, 2), slice(7, 8)):
        assert torch.equal(ops.stem_conv(flow[sl].contiguous(), ps, bias=bias, relu=True, div=1.0, mul=1.0, sub=0.0), big[sl])
    # the encoder stem with statistics: one 128 x 160 image (40 patches) against a batch of 8
    img = torch.from_numpy(rng.integers(0, 256, size=(8, 3, 128, 160)).astype(np.float32)).cuda()
    w3, b3 = _rand(rng, 64, 3, 7, 7, s=0.1).cuda(), _rand(rng, 64, s=0.3).cuda()
    p3 = ops.PackedStem(w3)
    raw8, st8 = ops.stem_conv(img, p3, bias=b3, relu=False, stats=True)
    raw1, st1 = ops.stem_conv(img[3:4].contiguous(), p3, bias=b3, relu=False, stats=True)
    assert torch.equal(raw1, raw8[3:4]) and torch.equal(st1, st8[3:4])


def test_random_shapes_against_library(rpe):
    """Forty seeded random problems (channel counts that are no multiple of anything, maps that end inside tiles, every
    supported kernel shape and stride, small and large launches) against the library's f32 convolution on the same GPU."""
    from rpe_amd import ops
    rng = np.random.default_rng(20261002)
    kinds = [(1, 1, 1), (3, 3, 1), (1, 5, 1), (5, 1, 1), (1, 3, 1), (7, 3, 1), (3, 3, 2), (1, 1, 2)]
    for case in range(40):
        kh, kw, stride = kinds[int(rng.integers(len(kinds)))]
        cin, cout = int(rng.integers(1, 200)), int(rng.integers(1, 300))
        h, w = int(rng.integers(2, 40)) * (2 if stride == 2 else 1), int(rng.integers(1, 24)) * 4
        if stride == 2 and ((h // 2) * (w // 2)) % 4:
            h += 2 if ((h // 2 + 1) * (w // 2)) % 4 == 0 else 0
        b = int(rng.choice([1, 2, 5, 40]))
        x = torch.from_numpy(rng.normal(size=(b, cin, h, w)).astype(np.float32)).cuda()
        wt = torch.from_numpy((rng.normal(size=(cout, cin, kh, kw)) * 0.1).astype(np.float32)).cuda()
        bias = torch.from_numpy(rng.normal(size=(cout,)).astype(np.float32)).cuda()
        relu = bool(rng.integers(2))
        ref = F.conv2d(x, wt, bias, stride=stride, padding=(kh // 2, kw // 2))
        ref = ref.clamp_min(0) if relu else ref
        out = torch.full_like(ref, float('nan'))
        ops.conv_fused(x, ops.PackedConv(wt, bias), ops.CONV_RELU if relu else ops.CONV_LINEAR, out, stride=stride)
        tol = 1e-5 * np.sqrt(cin * kh * kw) * float(x.abs().max()) * float(wt.abs().max()) + 1e-5
        err = float((out - ref).abs().max())
        assert err < tol, (case, kh, kw, stride, cin, cout, h, w, b, err, tol)


def test_winograd_random_shapes_against_library(rpe):
    """Sixty seeded random problems for the two Winograd kernels (channel counts that are no multiple of anything, maps that end
    inside tiles and inside border patches, one-tile maps, launches of one workgroup and of thousands) against the library's f32
    convolution on the same GPU; every epilogue the plain instantiations have (bias, ReLU, second destination, add)."""
    from rpe_amd import ops
    rng = np.random.default_rng(20261003)
    for case in range(60):
        kind = ('3x3', '1x5', '5x1')[case % 3]
        cin = int(rng.integers(1, 48)) * 4
        cout = int(rng.integers(1, 200))
        if kind == '3x3':
            h, w = int(rng.integers(1, 30)) * 2, int(rng.integers(1, 30)) * 2
            kh, kw = 3, 3
        else:
            h, w = int(rng.integers(1, 50)), int(rng.integers(1, 16)) * 4
            kh, kw = (1, 5) if kind == '1x5' else (5, 1)
        b = int(rng.choice([1, 2, 3, 24]))
        x = torch.from_numpy(rng.normal(size=(b, cin, h, w)).astype(np.float32)).cuda()
        wt = torch.from_numpy((rng.normal(size=(cout, cin, kh, kw)) * 0.1).astype(np.float32)).cuda()
        bias = torch.from_numpy(rng.normal(size=(cout,)).astype(np.float32)).cuda()
        relu, two = bool(rng.integers(2)), bool(rng.integers(2))
        ref = F.conv2d(x, wt, bias, padding=(kh // 2, kw // 2))
        out = torch.full_like(ref, float('nan')); out2 = torch.full_like(ref, float('nan')) if two else None
        if kind == '3x3':
            ops.conv_wino(x, ops.PackedWino(wt, bias), ops.CONV_RELU if relu else ops.CONV_LINEAR, out, out2=out2)
        else:
            add = torch.from_numpy(rng.normal(size=ref.shape).astype(np.float32)).cuda() if rng.integers(2) else None
            if add is not None:
                ref = ref + add
            ops.conv_wino1d(x, ops.PackedWino1d(wt, bias), ops.CONV_RELU if relu else ops.CONV_LINEAR, out, out2=out2, add=add)
        ref = ref.clamp_min(0) if relu else ref
        tol = 2e-5 * np.sqrt(cin * kh * kw) * float(x.abs().max()) * float(wt.abs().max()) + 1e-5
        err = float((out - ref).abs().max())
        assert err < tol, (case, kind, cin, cout, h, w, b, relu, err, tol)                     # (NaN left anywhere fails here too)
        if two:
            assert torch.equal(out, out2), (case, kind)


@pytest.mark.parametrize('cout,bias_val', [(64, 50.0), (64, -50.0), (96, 50.0), (128, -50.0), (112, 50.0)])
def test_instance_norm_statistics_survive_large_means(rpe, cout, bias_val):
    """Planes whose |mean| >> std (a conv bias of +-50, or a constant-plus-noise plane): E[x^2] - mean^2 from f32 sums
    would lose mean^2/var * 1e-7 of the variance; the epilogue's pivoted moments must not.  Bar: 2e-5 relative after
    normalisation.  cout = 112 also covers the tile-count rule for cout % 128 in 97..127 (128-pixel tiles with statistics)."""
    from rpe_amd import ops
    rng = np.random.default_rng(cout)
    b, cin, h, w = 2, 64, 48, 64
    x = _rand(rng, b, cin, h, w)
    wt = _rand(rng, cout, cin, 3, 3, s=0.05)
    wt[1] *= 1e-3                                                            # channel 1: a constant-plus-noise plane (std ~ 1e-3 around 50)
    bias = torch.full((cout,), bias_val)
    pc = ops.PackedConv(wt.cuda(), bias.cuda())
    pre = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    mean, var = pre.mean((2, 3), keepdim=True), pre.var((2, 3), unbiased=False, keepdim=True)
    eps = 1e-5
    stats = ops.conv_stats_buffer(b, cout, h, w, 'cuda')
    assert stats.shape[2] == -(-h * w // (256 if (cout % 128 != 0 and cout % 128 <= 96) else 128))
    guard = torch.full((4096,), 7.0, device='cuda')                          # allocated right behind: an overrun would be caught below
    raw = ops.conv_fused(x.cuda(), pc, ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda'), stats=stats)
    assert bool((guard == 7.0).all())
    mi = ops.instnorm_finalize(stats, h * w, eps=eps).cpu().double()
    inv_ref = 1.0 / torch.sqrt(var[:, :, 0, 0] + eps)
    assert float(((mi[..., 0] - mean[:, :, 0, 0]).abs() / mean[:, :, 0, 0].abs()).max()) < 1e-6
    assert float(((mi[..., 1] - inv_ref).abs() / inv_ref).max()) < 2e-5
    # the apply pass: (x - mean) * inv with the f32 (mean, inv) the kernel derives; then against the f64 truth.  On the
    # constant-plus-noise plane (channel 1: std ~ 1e-3 around 50) the f32 representation of the mean itself, ulp(50)/2 = 2e-6,
    # is 2e-3 standard deviations -- the reference's float32 instance norm has the same floor -- so the truth comparison
    # excludes that plane; its statistics were checked above.
    got = ops.instnorm_apply(raw.clone(), stats, eps=eps, relu=False).cpu().double()
    own = (raw.cpu().double() - mi[..., 0][:, :, None, None]) * mi[..., 1][:, :, None, None]
    scale = own.abs().amax((2, 3), keepdim=True)
    assert float(((got - own).abs() / scale).max()) < 2e-5
    want = (raw.cpu().double() - mean) / torch.sqrt(var + eps)
    keep = [c for c in range(cout) if c != 1]
    assert float(((got - want).abs() / scale)[:, keep].max()) < 2e-5


def test_stem_statistics_survive_large_means(rpe):
    from rpe_amd import ops
    rng = np.random.default_rng(5)
    img = torch.from_numpy(rng.uniform(0, 255, size=(2, 3, 64, 96)).astype(np.float32))
    wt = _rand(rng, 64, 3, 7, 7, s=0.05)
    bias = torch.full((64,), 50.0)
    ps = ops.PackedStem(wt.cuda())
    raw, stats = ops.stem_conv(img.cuda(), ps, bias=bias.cuda(), relu=False, stats=True)
    pre = F.conv2d(2 * (img.double() / 255) - 1, wt.double(), bias.double(), stride=2, padding=3)
    mean, var = pre.mean((2, 3)), pre.var((2, 3), unbiased=False)
    mi = ops.instnorm_finalize(stats, 32 * 48, eps=1e-5).cpu().double()
    inv_ref = 1.0 / torch.sqrt(var + 1e-5)
    assert float(((mi[..., 0] - mean).abs() / mean.abs()).max()) < 1e-6
    assert float(((mi[..., 1] - inv_ref).abs() / inv_ref).max()) < 2e-5


@pytest.mark.parametrize('cin,cout,h,w,b', [(256, 192, 64, 80, 2), (128, 64, 64, 80, 1), (256, 126, 44, 48, 2), (128, 256, 32, 40, 1),
                                            (8, 20, 6, 10, 3), (64, 64, 128, 160, 1)])
def test_winograd_3x3_matches_f64(rpe, cin, cout, h, w, b):
    """rpe_conv_wino (F(2x2,3x3) on the f32 matrix cores; the update block's convc2 / convf2 / conv / FlowHead.conv1 shapes, a
    ragged one and the 1280x1024 grid) against the f64 convolution: same bar as the direct kernel, times 3 for the transforms.
    Destinations are channel slices; out2 receives a copy; neighbours stay untouched."""
    from rpe_amd import ops
    rng = np.random.default_rng(cin + cout)
    x, wt, bias = _rand(rng, b, cin, h, w), _rand(rng, cout, cin, 3, 3, s=0.05), _rand(rng, cout, s=0.5)
    assert ops.PackedWino.supported(wt, h, w)
    pw = ops.PackedWino(wt.cuda(), bias.cuda())
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    obuf = torch.full((b, cout + 8, h, w), -7.0, device='cuda')
    o2buf = torch.full((b, cout + 2, h, w), -7.0, device='cuda')
    xbuf = torch.zeros(b, cin + 3, h, w, device='cuda')                      # the input is a channel slice too
    xbuf[:, 3:] = x.cuda()
    ops.conv_wino(xbuf[:, 3:], pw, ops.CONV_RELU, obuf[:, 4:4 + cout], out2=o2buf[:, 2:])
    got = obuf[:, 4:4 + cout].cpu().double()
    assert (got - ref.clamp_min(0)).abs().max() < 3 * _tol(x, wt)
    assert torch.equal(obuf[:, 4:4 + cout], o2buf[:, 2:])
    assert (obuf[:, :4] == -7.0).all() and (obuf[:, 4 + cout:] == -7.0).all() and (o2buf[:, :2] == -7.0).all()
    lin = ops.conv_wino(x.cuda(), pw, ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda')).cpu().double()
    assert (lin - ref).abs().max() < 3 * _tol(x, wt)
    # the prepared launcher gives the same bits
    out3 = torch.empty(b, cout, h, w, device='cuda')
    ops.conv_wino(x.cuda(), pw, ops.CONV_LINEAR, out3, prepare=True)()
    assert torch.equal(out3.cpu().double(), lin)


def test_winograd_rejects_what_it_cannot_do(rpe):
    from rpe_amd import ops
    with pytest.raises(rpe.RpeError):
        ops.PackedWino(torch.zeros(8, 6, 3, 3, device='cuda'))               # cin % 4
    with pytest.raises(rpe.RpeError):
        ops.PackedWino(torch.zeros(8, 8, 1, 5, device='cuda'))
    pw = ops.PackedWino(torch.zeros(8, 8, 3, 3, device='cuda'))
    with pytest.raises(rpe.RpeError, match='UNSUPPORTED'):
        ops.conv_wino(torch.zeros(1, 8, 7, 12, device='cuda'), pw, ops.CONV_RELU, torch.empty(1, 8, 7, 12, device='cuda'))   # odd height


@pytest.mark.parametrize('c,h,w,b', [(64, 64, 80, 3), (96, 44, 48, 2), (128, 32, 40, 2)])
def test_winograd_encoder_epilogues_match_f64(rpe, c, h, w, b):
    """The encoders' epilogues on the Winograd kernel: folded batch norm (scale, shift) + ReLU + residual + ReLU; instance-norm
    moments per 16x8-pixel patch (-> rpe_instnorm_finalize / rpe_instnorm_apply); and the input normalised + ReLU'd while it
    is transformed (pre_norm).  Same references and bars as test_encoder_block_epilogues_match_f64, times 3 for the transforms."""
    from rpe_amd import ops
    rng = np.random.default_rng(c + h + 1)
    x, wt, bias = _rand(rng, b, c, h, w), _rand(rng, c, c, 3, 3, s=0.05), _rand(rng, c, s=0.5)
    res = _rand(rng, b, c, h, w).abs()
    scale, shift = _rand(rng, c).abs() + 0.5, _rand(rng, c, s=0.3)
    pw = ops.PackedWino(wt.cuda(), None)
    conv = F.conv2d(x.double(), wt.double(), None, padding=1)
    ref = (res.double() + (conv * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]).clamp_min(0)).clamp_min(0)
    got = ops.conv_wino(x.cuda(), pw, ops.CONV_RELU, torch.empty(b, c, h, w, device='cuda'), scale=scale.cuda(), bias=shift.cuda(), residual=res.cuda())
    assert (got.cpu().double() - ref).abs().max() < 3 * _tol(x, wt) * 2.5
    # instance norm of conv + bias from the per-patch moments
    pre = conv + bias.double()[None, :, None, None]
    mean, var = pre.mean((2, 3)), pre.var((2, 3), unbiased=False)
    stats = ops.conv_wino_stats_buffer(b, c, h, w, 'cuda')
    raw = ops.conv_wino(x.cuda(), pw, ops.CONV_LINEAR, torch.empty(b, c, h, w, device='cuda'), bias=bias.cuda(), stats=stats)
    assert (raw.cpu().double() - pre).abs().max() < 3 * _tol(x, wt)
    st = stats.cpu().double()                                              # tile-major: (b, records, c, 3)
    assert float(st[..., 0].sum(1).min()) == float(st[..., 0].sum(1).max()) == h * w
    mi = ops.instnorm_finalize(stats, h * w, eps=1e-5).cpu().double()
    assert float((mi[..., 0] - mean).abs().max()) < 1e-5 and float((mi[..., 1] * torch.sqrt(var + 1e-5) - 1).abs().max()) < 2e-5
    ref2 = (res.double() + ((pre - mean[:, :, None, None]) / torch.sqrt(var + 1e-5)[:, :, None, None]).clamp_min(0)).clamp_min(0)
    got2 = ops.instnorm_apply(raw, stats, eps=1e-5, relu=True, residual=res.cuda())
    inv = float((1 / torch.sqrt(var + 1e-5)).max())
    assert (got2.cpu().double() - ref2).abs().max() < (3 * _tol(x, wt) + 2e-6) * inv * 2
    # pre_norm: conv(relu((x - m) * i)) with (m, i) per input plane
    m_i = torch.stack((_rand(rng, b, c, s=0.3), _rand(rng, b, c).abs() + 0.5), dim=-1).contiguous()
    xin = ((x.double() - m_i[..., 0].double()[:, :, None, None]) * m_i[..., 1].double()[:, :, None, None]).clamp_min(0)
    ref3 = F.conv2d(xin, wt.double(), bias.double(), padding=1)
    got3 = ops.conv_wino(x.cuda(), pw, ops.CONV_LINEAR, torch.empty(b, c, h, w, device='cuda'), bias=bias.cuda(), pre_norm=m_i.cuda())
    assert (got3.cpu().double() - ref3).abs().max() < 3 * _tol(xin.float(), wt) + 1e-5


# ------------------------------------------------------------------------------------ trained-like statistics (round 3)
def _trained_like(rng, b, cin, cout, kh, kw, h, w, flow_channels=True):
    """What the layers see with trained weights rather than N(0,1) test data: post-ReLU activations with a positive mean
    (relu(N(2,1))), the two raw flow channels the motion encoder concatenates in front of the GRU at +-50 px, a hidden state
    saturated at +-1, heavy-tailed weights (Student-t, 3 degrees of freedom) with a few 10x outliers."""
    x = torch.from_numpy(rng.normal(2.0, 1.0, size=(b, cin, h, w)).astype(np.float32)).clamp_min(0)
    q = cin // 4
    x[:, :q] = torch.from_numpy(np.sign(rng.normal(size=(b, q, h, w))).astype(np.float32)) * (1 - 1e-3 * torch.rand(b, q, h, w))
    if flow_channels:
        smooth = F.interpolate(torch.from_numpy(rng.normal(size=(b, 2, 5, 6)).astype(np.float32)), size=(h, w), mode='bilinear', align_corners=True)
        x[:, -2:] = 50.0 * smooth / smooth.abs().max()
    wt = torch.from_numpy((rng.standard_t(3, size=(cout, cin, kh, kw)) * 0.02).astype(np.float32))
    idx = rng.integers(0, wt.numel(), size=12)
    wt.view(-1)[idx] *= 10.0
    bias = torch.from_numpy(rng.normal(0, 0.3, size=cout).astype(np.float32))
    return x, wt, bias


def _errs(got, ref):
    e = (got.cpu().double() - ref).abs()
    return float(e.max()), float(torch.sqrt((e * e).mean()))


@pytest.mark.parametrize('kind,cin,cout,h,w', [('3x3', 256, 192, 64, 80), ('3x3', 128, 256, 64, 80), ('3x3', 64, 64, 128, 160),
                                               ('1x5', 256, 256, 64, 80), ('5x1', 256, 128, 64, 80), ('1x5', 256, 128, 44, 48)])
@pytest.mark.parametrize('flow_channels', [False, True])
def test_winograd_error_relative_to_direct_on_trained_like_statistics(rpe, kind, cin, cout, h, w, flow_channels):
    """F(2x2,3x3) and F(4,5) are exact in real arithmetic but their transforms (constants up to 5.25) cost float32 bits that
    depend on the data's statistics; zero-mean Gaussian test data is the friendliest case.  Here: trained-like statistics,
    error of each Winograd kernel against the f64 convolution RELATIVE TO THE DIRECT f32 KERNEL'S OWN ERROR on the same data
    (rpe_conv_fused: a plain f32 fma chain).  Bars: RMS ratio <= 3, max ratio <= 4."""
    from rpe_amd import ops
    rng = np.random.default_rng(cin * 1000 + cout * 7 + h + 3 * len(kind) + ord(kind[0]) + int(flow_channels))
    kh, kw = {'3x3': (3, 3), '1x5': (1, 5), '5x1': (5, 1)}[kind]
    b = 2
    x, wt, bias = _trained_like(rng, b, cin, cout, kh, kw, h, w, flow_channels)
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=(kh // 2, kw // 2))
    direct = ops.conv_fused(x.cuda(), ops.PackedConv(wt.cuda(), bias.cuda()), ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda'))
    if kind == '3x3':
        wino = ops.conv_wino(x.cuda(), ops.PackedWino(wt.cuda(), bias.cuda()), ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda'))
    else:
        wino = ops.conv_wino1d(x.cuda(), ops.PackedWino1d(wt.cuda(), bias.cuda()), ops.CONV_LINEAR, torch.empty(b, cout, h, w, device='cuda'))
    dmax, drms = _errs(direct, ref)
    wmax, wrms = _errs(wino, ref)
    print(f'{kind} {cin}->{cout} flow_channels={flow_channels}: |out| {float(ref.abs().max()):.1f}; direct max {dmax:.2e} rms {drms:.2e}; '
          f'winograd max {wmax:.2e} rms {wrms:.2e}; ratios {wmax / dmax:.2f} / {wrms / drms:.2f}')
    assert wrms <= 3.0 * drms and wmax <= 4.0 * dmax


@pytest.mark.parametrize('impl', ['direct', 'winograd'])
def test_gru_gates_on_saturated_state_and_large_flow(rpe, impl):
    """One SepConvGRU half on trained-like inputs: hidden state at +-1, the flow channels at +-50, pre-activations spread
    over +-30 (so sigmoid / tanh run into both saturated ends on the hardware exp/rcp path).  Bar: absolute 2e-6 on the gate
    (a saturated sigmoid must be 0 or 1 to rounding, not NaN / denormal garbage) and the f64 blend to the conv tolerance."""
    from rpe_amd import ops
    Packed, conv = (ops.PackedConv, ops.conv_fused) if impl == 'direct' else (ops.PackedWino1d, ops.conv_wino1d)
    c, b, h, w = 128, 2, 64, 80
    rng = np.random.default_rng(77)
    hx, wzr, bzr = _trained_like(rng, b, 2 * c, 2 * c, 1, 5, h, w)
    hx[:, :c] = torch.sign(hx[:, :c] - 1.0) * (1 - 1e-4 * torch.rand(b, c, h, w))          # saturated hidden state
    _, wq, bq = _trained_like(rng, b, 2 * c, c, 1, 5, h, w)
    azr = torch.from_numpy(rng.normal(0, 10.0, size=(b, 2 * c, h, w)).astype(np.float32))   # context terms push the gates to both ends
    aq = torch.from_numpy(rng.normal(0, 10.0, size=(b, c, h, w)).astype(np.float32))
    hid = hx[:, :c].double()
    pre = _ref_conv(hx, wzr, bzr, azr)
    zr = torch.sigmoid(pre)
    z, r = zr[:, :c], zr[:, c:]
    g_hx, g_rhx = hx.cuda(), hx.cuda().clone()
    g_z = torch.empty(b, c, h, w, device='cuda')
    conv(g_hx, Packed(wzr.cuda(), bzr.cuda()), ops.CONV_GATE_ZR, g_z, out2=g_rhx[:, :c], add=azr.cuda(), hidden=g_hx[:, :c], gate_channels=c)
    assert bool(torch.isfinite(g_z).all()) and float(g_z.min()) >= 0.0 and float(g_z.max()) <= 1.0
    # sigmoid' <= 1/4: the gate error is a quarter of the pre-activation error + the exp/rcp path's 3e-7
    tpre = _tol(hx, wzr) * (2 if impl == 'winograd' else 1)
    assert float((g_z.cpu().double() - z).abs().max()) < 0.25 * tpre + 2e-6
    assert float(pre.abs().max()) > 25 and float((z < 1e-9).float().mean()) > 0.001 and float((z > 1 - 1e-9).float().mean()) > 0.001
    rh = g_rhx[:, :c].cpu()
    q = torch.tanh(_ref_conv(torch.cat((rh, hx[:, c:]), 1), wq, bq, aq))
    conv(g_rhx, Packed(wq.cuda(), bq.cuda()), ops.CONV_GATE_H, g_hx[:, :c], add=aq.cuda(), hidden=g_hx[:, :c], zgate=g_z)
    hnew = (1 - g_z.cpu().double()) * hid + g_z.cpu().double() * q
    assert bool(torch.isfinite(g_hx).all()) and float(g_hx[:, :c].abs().max()) <= 1.0 + 1e-6
    assert float((g_hx[:, :c].cpu().double() - hnew).abs().max()) < _tol(hx, wq) * (2 if impl == 'winograd' else 1) + 4e-6


def test_instnorm_apply_with_a_raw_residual(rpe):
    """rpe_instnorm_apply_ex: the shortcut of fnet's first residual block is the stem's RAW output, normalised + ReLU'd while the
    block's last pass reads it -- against the f64 evaluation, and bit-identical to normalising the shortcut in a pass of its own."""
    from rpe_amd import ops
    rng = np.random.default_rng(9)
    b, c, h, w = 2, 64, 32, 48
    x, wt, bias = _rand(rng, b, c, h, w), _rand(rng, c, c, 3, 3, s=0.05), _rand(rng, c, s=0.5)
    res_raw = _rand(rng, b, c, h, w, s=2.0) + 1.0
    rm, rv = res_raw.double().mean((2, 3), keepdim=True), res_raw.double().var((2, 3), unbiased=False, keepdim=True)
    res_mi = torch.stack(((rm[:, :, 0, 0]).float(), (1 / torch.sqrt(rv[:, :, 0, 0] + 1e-5)).float()), dim=-1).contiguous().cuda()
    pw = ops.PackedWino(wt.cuda(), None)
    stats = ops.conv_wino_stats_buffer(b, c, h, w, 'cuda')
    raw = ops.conv_wino(x.cuda(), pw, ops.CONV_LINEAR, torch.empty(b, c, h, w, device='cuda'), bias=bias.cuda(), stats=stats)
    got = ops.instnorm_apply(raw.clone(), stats, eps=1e-5, relu=True, residual=res_raw.cuda(), residual_norm=res_mi)
    pre = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    mean, var = pre.mean((2, 3), keepdim=True), pre.var((2, 3), unbiased=False, keepdim=True)
    shortcut = ((res_raw.double() - rm) / torch.sqrt(rv + 1e-5)).clamp_min(0)
    ref = (shortcut + ((pre - mean) / torch.sqrt(var + 1e-5)).clamp_min(0)).clamp_min(0)
    inv = float((1 / torch.sqrt(var + 1e-5)).max())
    assert float((got.cpu().double() - ref).abs().max()) < (3 * _tol(x, wt) + 2e-6) * inv * 2 + 2e-6
    own = ((res_raw.cuda() - res_mi[..., 0][:, :, None, None]) * res_mi[..., 1][:, :, None, None]).clamp_min(0)
    two_pass = ops.instnorm_apply(raw.clone(), stats, eps=1e-5, relu=True, residual=own)
    assert torch.equal(got, two_pass)
    # a stride-2 block's shortcut: norm3 has NO ReLU (residual_relu=False) -- again the same bits as normalising it in a pass of its own
    got_nr = ops.instnorm_apply(raw.clone(), stats, eps=1e-5, relu=True, residual=res_raw.cuda(), residual_norm=res_mi, residual_relu=False)
    own_nr = (res_raw.cuda() - res_mi[..., 0][:, :, None, None]) * res_mi[..., 1][:, :, None, None]
    assert torch.equal(got_nr, ops.instnorm_apply(raw.clone(), stats, eps=1e-5, relu=True, residual=own_nr))
    assert not torch.equal(got_nr, got)
    with pytest.raises(rpe.RpeError):
        ops.instnorm_apply(raw.clone(), stats, residual_norm=res_mi)              # a norm without a residual


@pytest.mark.parametrize('b,c,h,w', [(3, 64, 256, 320), (2, 96, 128, 160), (2, 128, 64, 80), (1, 64, 44, 52)])
def test_instnorm_apply_with_the_moments_given(rpe, b, c, h, w):
    """rpe_instnorm_apply_ex with tiles = 0: the (mean, 1/std) pairs of rpe_instnorm_finalize instead of the records, the plane split
    over up to 16 workgroups (16 at 256 x 320, 4 at 128 x 160, 1 below 32 KB; 44 x 52 = 2288 pixels: a plane that is no whole number of
    64-byte pieces per workgroup).  Against the f64 evaluation, and equal to the records route up to the rounding of the pairs to f32;
    every element written exactly once (NaN canaries), nothing outside the tensor."""
    from rpe_amd import ops
    rng = np.random.default_rng(b * 7 + c + h)
    x, wt, bias = _rand(rng, b, c, h, w), _rand(rng, c, c, 3, 3, s=0.05), _rand(rng, c, s=0.5)
    res = _rand(rng, b, c, h, w, s=0.7)
    pw = ops.PackedWino(wt.cuda(), None)
    stats = ops.conv_wino_stats_buffer(b, c, h, w, 'cuda')
    raw = ops.conv_wino(x.cuda(), pw, ops.CONV_LINEAR, torch.empty(b, c, h, w, device='cuda'), bias=bias.cuda(), stats=stats)
    mi = ops.instnorm_finalize(stats, h * w, eps=1e-5, channels=c)
    buf = torch.full((b * c * h * w + 64,), float('nan'), device='cuda')
    out = buf[32:32 + b * c * h * w].view(b, c, h, w)
    got = ops.instnorm_apply(raw, mi, relu=True, residual=res.cuda(), out=out)
    assert bool(torch.isnan(buf[:32]).all()) and bool(torch.isnan(buf[32 + b * c * h * w:]).all()) and not bool(torch.isnan(got).any())
    rec = ops.instnorm_apply(raw.clone(), stats, eps=1e-5, relu=True, residual=res.cuda())
    pre = raw.cpu().double()
    mean, var = pre.mean((2, 3), keepdim=True), pre.var((2, 3), unbiased=False, keepdim=True)
    ref = (res.double() + ((pre - mean) / torch.sqrt(var + 1e-5)).clamp_min(0)).clamp_min(0)
    scale = float((pre - mean).abs().max() / torch.sqrt(var + 1e-5).min())
    assert float((got.cpu().double() - ref).abs().max()) < 4e-7 * scale + 1e-6
    assert float((got - rec).abs().max()) < 4e-7 * scale + 1e-6
    with pytest.raises(rpe.RpeError):
        ops.instnorm_apply(raw, mi[:, :c - 1].contiguous(), relu=True)


@pytest.mark.parametrize('cin,cout,h,w,mode', [(324, 256, 44, 48, 'relu'), (256, 576, 44, 48, 'linear'), (128, 256, 64, 80, 'tanh'), (20, 70, 6, 12, 'linear')])
def test_conv1x1_routes_agree_bitwise(rpe, cin, cout, h, w, mode):
    """ops.Conv1x1 sends a launch to rpe_conv1x1 (128 x 128 tiles on LDS-DMA rings) or to rpe_conv_fused (128- or 64-wide tiles) by its
    workgroup count -- i.e. by the batch.  All three sum the same products in the same order (k_conv1x1's operand rows are permuted to
    k_conv_igemm's pairing of channels j and 8 + j), so a map's result is the same bits whichever route its batch took."""
    from rpe_amd import ops
    rng = np.random.default_rng(cin * 7 + cout)
    b = 14
    x, wt, bias = _rand(rng, b, cin, h, w).cuda(), _rand(rng, cout, cin, 1, 1, s=0.05).cuda(), _rand(rng, cout, s=0.3).cuda()
    m = {'relu': ops.CONV_RELU, 'tanh': ops.CONV_TANH, 'linear': ops.CONV_LINEAR}[mode]
    gemm = ops.conv1x1(x, ops.PackedConv1x1(wt, bias), m, torch.empty(b, cout, h, w, device='cuda'))
    fused = ops.conv_fused(x, ops.PackedConv(wt, bias), m, torch.empty(b, cout, h, w, device='cuda'))
    assert torch.equal(gemm, fused)
    both = ops.Conv1x1(wt, bias)
    for sl in (slice(0, 1), slice(5, 7)):                      # small launches: rpe_conv_fused's 64 x 64 tiles
        assert torch.equal(both(x[sl].contiguous(), m, torch.empty(sl.stop - sl.start, cout, h, w, device='cuda')), gemm[sl])
    assert torch.equal(both(x, m, torch.empty(b, cout, h, w, device='cuda')), gemm)


@pytest.mark.parametrize('cin,cout,h,w,b,mode', [(324, 256, 64, 80, 32, 'relu'), (324, 256, 64, 80, 1, 'relu'), (128, 256, 44, 48, 2, 'tanh'),
                                                 (256, 576, 44, 48, 2, 'linear'), (20, 70, 6, 10, 3, 'linear'), (17, 130, 2, 2, 1, 'relu')])
def test_conv1x1_gemm_matches_f64(rpe, cin, cout, h, w, b, mode):
    """rpe_conv1x1 (LDS-DMA GEMM for 1x1 convolutions: convc1 behind the lookup, the 1x1 output layers, the mask head) against the f64
    convolution: channel counts that end inside a 16-channel step / a 128-channel tile, pixel counts that end inside a 128-pixel tile,
    input and outputs as channel slices of wider buffers, a second destination."""
    from rpe_amd import ops
    rng = np.random.default_rng(cin + cout + h)
    x, wt, bias = _rand(rng, b, cin, h, w), _rand(rng, cout, cin, 1, 1, s=0.05), _rand(rng, cout, s=0.3)
    ref = _ref_conv(x, wt, bias, None)
    ref = {'relu': ref.clamp_min(0), 'tanh': torch.tanh(ref), 'linear': ref}[mode]
    m = {'relu': ops.CONV_RELU, 'tanh': ops.CONV_TANH, 'linear': ops.CONV_LINEAR}[mode]
    xbuf = torch.full((b, cin + 8, h, w), float('nan'), device='cuda'); xbuf[:, 4:4 + cin] = x.cuda()
    obuf = torch.full((b, cout + 5, h, w), -7.0, device='cuda'); o2 = torch.full((b, cout + 1, h, w), -7.0, device='cuda')
    pc = ops.PackedConv1x1(wt.cuda(), bias.cuda())
    ops.conv1x1(xbuf[:, 4:4 + cin], pc, m, obuf[:, 2:2 + cout], out2=o2[:, 1:])
    got = obuf[:, 2:2 + cout].cpu().double()
    assert float((got - ref).abs().max()) < _tol(x, wt) + (1e-6 if mode == 'tanh' else 0.0)
    assert torch.equal(obuf[:, 2:2 + cout], o2[:, 1:])
    assert bool((obuf[:, :2] == -7.0).all()) and bool((obuf[:, 2 + cout:] == -7.0).all()) and bool((o2[:, 0] == -7.0).all())
    again = ops.conv1x1(xbuf[:, 4:4 + cin], pc, m, torch.empty(b, cout, h, w, device='cuda'), prepare=True)()
    assert torch.equal(again, obuf[:, 2:2 + cout])
    with pytest.raises(rpe.RpeError, match='UNSUPPORTED'):
        ops.conv_fused(xbuf[:, 4:4 + cin], pc, ops.CONV_GATE_H, obuf[:, 2:2 + cout], hidden=obuf[:, 2:2 + cout], zgate=obuf[:, 2:2 + cout], entry='rpe_conv1x1')


@pytest.mark.parametrize('cin,cout', [(256, 192), (128, 64), (256, 126), (128, 256)])
def test_winograd_small_and_large_launches_agree_bitwise(rpe, cin, cout):
    """rpe_conv_wino runs launches below 512 workgroups on 32-channel tiles (two workgroups per CU hide each other's DMA latency at
    batch 1-2) and larger ones on 64-channel tiles (+ a 32-channel tail).  Every output element accumulates the same products in the
    same order either way: a batch of 16 maps (large) must equal the same maps convolved two at a time (small) bit for bit."""
    from rpe_amd import ops
    rng = np.random.default_rng(cin + cout)
    b, h, w = 16, 64, 80
    x, wt, bias = _rand(rng, b, cin, h, w, s=0.5).cuda(), _rand(rng, cout, cin, 3, 3, s=0.05).cuda(), _rand(rng, cout, s=0.1).cuda()
    pw = ops.PackedWino(wt, bias)
    big = ops.conv_wino(x, pw, ops.CONV_RELU, torch.empty(b, cout, h, w, device='cuda'))
    assert 40 * -(-cout // 64) * b >= 512 > 40 * -(-cout // 64) * 2                      # the two launch classes
    for i in range(0, b, 6):
        small = ops.conv_wino(x[i:i + 2].contiguous(), pw, ops.CONV_RELU, torch.empty(2, cout, h, w, device='cuda'))
        assert torch.equal(big[i:i + 2], small)


@pytest.mark.parametrize('kh,kw', [(1, 5), (5, 1)])
def test_winograd_1d_tile_classes_agree_bitwise(rpe, kh, kw):
    """rpe_conv_wino1d runs launches below 768 workgroups (1.5 rounds of the chip's 512 slots) on 32-channel tiles and larger
    ones on 64-channel tiles; an output element accumulates the same products in the same order either way.  One GRU half on 24
    maps (z|r: 1 920 workgroups -> 64-channel tiles) must equal the same maps run two at a time (32-channel tiles) bit for bit --
    gate epilogues, in-place hidden update and all."""
    from rpe_amd import ops
    c, b, h, w = 128, 24, 64, 80
    rng = np.random.default_rng(kh * 7 + kw)
    hx = _rand(rng, b, 2 * c, h, w, s=0.5).cuda()
    wzr, bzr, azr = _rand(rng, 2 * c, 2 * c, kh, kw, s=0.03).cuda(), _rand(rng, 2 * c, s=0.1).cuda(), _rand(rng, b, 2 * c, h, w, s=0.3).cuda()
    wl, bl = _rand(rng, 4 * c, 2 * c, kh, kw, s=0.03).cuda(), _rand(rng, 4 * c, s=0.1).cuda()
    pzr, pl = ops.PackedWino1d(wzr, bzr), ops.PackedWino1d(wl, bl)
    assert 20 * 4 * b >= 768 > 20 * 8 * 2                 # (the 512-channel layer of two maps: 320 workgroups of 64 channels)

    def half(hx_in, add):
        n = hx_in.shape[0]
        g_hx, g_rhx, g_z = hx_in.clone(), hx_in.clone(), torch.empty(n, c, h, w, device='cuda')
        ops.conv_wino1d(g_hx, pzr, ops.CONV_GATE_ZR, g_z, out2=g_rhx[:, :c], add=add, hidden=g_hx[:, :c], gate_channels=c)
        lin = ops.conv_wino1d(g_rhx, pl, ops.CONV_RELU, torch.empty(n, 4 * c, h, w, device='cuda'))     # 512 channels: eight 64-tiles
        return g_z, g_rhx, lin
    big = half(hx, azr)
    for i in (0, 10, 22):
        small = half(hx[i:i + 2].contiguous(), azr[i:i + 2].contiguous())
        for a, s_ in zip(big, small):
            assert torch.equal(a[i:i + 2], s_)


@pytest.mark.parametrize('cin,cout,kh,kw,stride,pad,h,w,b', [
    (3, 64, 7, 7, 2, (3, 3), 46, 50, 2), (64, 96, 3, 3, 2, (1, 1), 45, 45, 2), (96, 96, 3, 3, 1, (1, 1), 23, 45, 3), (64, 96, 1, 1, 2, (0, 0), 45, 90, 1),
    (256, 256, 1, 5, 1, (0, 2), 44, 45, 2), (256, 128, 5, 1, 1, (2, 0), 45, 45, 1), (324, 256, 1, 1, 1, (0, 0), 45, 45, 2), (2, 128, 7, 7, 1, (3, 3), 44, 45, 2),
    (17, 70, 3, 3, 1, (1, 1), 5, 7, 2), (130, 2, 3, 3, 1, (1, 1), 9, 11, 1)])
def test_generic_convolution_matches_f64(rpe, cin, cout, kh, kw, stride, pad, h, w, b):
    """rpe_conv_direct (the route of map sizes the tuned kernels refuse) against torch's f64 convolution: every kernel shape of RAFT,
    odd maps, channel counts that end inside a tile or a K step, channel-slice inputs and outputs, bias and ReLU."""
    from rpe_amd import ops
    rng = np.random.default_rng(cin * 31 + cout + h)
    x, wt, bias = _rand(rng, b, cin, h, w), _rand(rng, cout, cin, kh, kw, s=0.05), _rand(rng, cout, s=0.3)
    ref = F.conv2d(x.double(), wt.double(), bias.double(), stride=stride, padding=pad)
    xbuf = torch.full((b, cin + 3, h, w), float('nan'), device='cuda'); xbuf[:, 2:2 + cin] = x.cuda()
    ho, wo = ref.shape[-2:]
    obuf = torch.full((b, cout + 4, ho, wo), -7.0, device='cuda')
    ops.conv_direct(xbuf[:, 2:2 + cin], wt.cuda(), bias.cuda(), stride, pad, out=obuf[:, 1:1 + cout])
    tol = _tol(x, wt) * max(1.0, (cin * kh * kw / 256) ** 0.5) + 1e-6
    assert float((obuf[:, 1:1 + cout].cpu().double() - ref).abs().max()) < tol
    assert float(obuf[:, 0].min()) == float(obuf[:, 1 + cout:].max()) == -7.0          # nothing outside the slice is touched
    got = ops.conv_direct(x.cuda(), wt.cuda(), None, stride, pad, relu=True)
    assert tuple(got.shape) == tuple(ref.shape)
    assert float((got.cpu().double() - (ref - bias.double()[None, :, None, None]).clamp_min(0)).abs().max()) < tol
    with pytest.raises(rpe.RpeError):
        ops.conv_direct(x.cuda(), wt.cuda(), None, 3, pad)                              # stride 3
