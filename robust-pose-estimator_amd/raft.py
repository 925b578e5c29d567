"""RAFT host mirror: same constructor / forward contract as the reference's RAFT fork
(``RAFT(config)(img1, img2, upsample=True) -> (flow_predictions, hidden, context)``, call sites
core/pose/pose_net.py:21-22,47,65,129) and the upstream parameter names, so ``raft-things.pth`` /
``poseNet_*.pth`` state dicts load unchanged.

What runs where (MI355X-first split), all hand-written HIP through the C ABI:
  * all-pairs correlation + pyramid and the per-iteration window lookup (csrc/corr.hip)
  * every convolution of the update block with its bias / ReLU / concat / GRU-gate epilogue, the encoders' residual
    blocks (3x3 stride 1 and 2, 1x1 stride-2 shortcut) with folded batch norm / instance-norm statistics, as f32-MFMA
    implicit GEMMs (csrc/conv.hip); the 7x7 stems and convf1 (csrc/stem.hip)
  * the update block's and the encoders' 3x3 stride-1 layers as Winograd F(2x2,3x3) (csrc/conv_wino.hip), the GRU's 1x5 / 5x1
    layers with their gate epilogues as Winograd F(4,5) (csrc/conv_wino1d.hip); the encoders' final 1x1 (cnet's with its
    tanh | ReLU split in the epilogue) and the mask head's 1x1 on the implicit GEMM
  * flow-head output layer with the coords / flow bookkeeping of the loop, convex up-sampling, norm / bias passes, slice copies
    (csrc/raft_ops.hip)
  * map sizes the tuned kernels refuse (odd maps, rows that are not whole 16-byte quads: image widths that are not a multiple of 32,
    heights not a multiple of 16): the same layers on the generic implicit GEMM (csrc/conv_direct.hip) with the stand-alone epilogue
    kernels -- robust, slower; no launch of a pass goes to a library at any size
Exact re-associations used (results identical up to float rounding):
  * convz/convr of each GRU half share their input, so their weights are stacked into one 256-channel conv
  * the GRU input is (h | inp | motion | flow) and the context `inp` is the same in all 12 iterations, so the
    inp-channel part of every GRU convolution (+ bias) is computed once per pass and added inside the gate kernels:
    the per-iteration convolutions read 256 instead of 384 channels (a third of the GRU's FLOPs hoisted out of the loop)
  * the (h | x) concatenations live in two persistent buffers; the gate kernels write r*h and the new h in place
  * the mask head + convex up-sampling only run for the predictions that are returned (``all_flows=False``
    returns just the final one, which is all the reference's PoseNet reads: ``[0][-1]``)
The RAFT architecture itself is restated from princeton-vl/RAFT (the reference's submodule is empty);
see oracle/raft.py for the CPU restatement these kernels are tested against.
"""
import torch
import torch.nn as nn

from . import ops

# Module-level route constants (no environment switches in the product: tools/ and bench.py's labelled experiments set these attributes)
WINOGRAD = True          # 3x3 layers as F(2x2,3x3), the GRU's 1x5 / 5x1 as F(4,5); False = the direct implicit GEMM (A/B measurements)
CORR_BF16X3 = False      # EXPERIMENT: correlation products as six bf16 products of an exact 3-way split (bench.py --corr-bf16x3)
CONV_BF16X3 = False      # LABELLED VARIANT (bench.py --conv-bf16x3; never the headline): the update block's 3x3 layers with >= 128 input
#                          channels (convc2, conv, FlowHead.conv1, the mask head's 3x3) through rpe_conv_wino_x3 -- Winograd products as six
#                          bf16 products of an exact 3-way split --, the 1x1 layers with LINEAR / RELU epilogues (convc1, fnet's output layer,
#                          the ReLU half of cnet's, the mask head's 1x1) through rpe_conv1x1_x3, and the correlation build through
#                          k_corr_build_x3.  The layers it does NOT cover (measured no faster) stay on the f32 matrix cores: the encoders'
#                          3x3 layers, convf2, the GRU's 1x5 / 5x1 layers (unless X3_GRU), the tanh half of cnet's output layer, the stems.
X3_MIN_CIN = 128         # rpe_conv_wino_x3 pays from 8 K steps of 16 channels on
X3_GRU = False           # under CONV_BF16X3, also the SepConvGRU's 1x5 / 5x1 layers on rpe_conv_wino1d_x3.  Off: in the bench step its launches
#                          take 449 / 410 us (z|r, 1x5 / 5x1) and 251 / 237 us (q) against rpe_conv_wino1d's 407 / 395 and 240 / 224 -- one
#                          workgroup per CU has nothing to run beside its memory-bound gate pass (csrc/conv_wino1d_x3.hip); kept, tested, opt-in.
# The motion encoder's flow branch (convf1 -> convf2) on a side stream beside lookup -> convc1 -> convc2.  Measured (MI355X, 640x512):
# batch 1-2 (sequential tracking) 9.22 -> 9.08 ms per frame pair, 110.2 -> 111.9 frames/s; batch 32: 72.49 vs 72.41 ms per step (every
# launch fills the chip on its own), so it is used for small passes only.
SIDE_STREAM = True
# The update loop as ONE call into the library (ops.OpList -> rpe_run_ops): the 12 iterations' ~130-180 launches and stream fork / joins are
# enqueued by a C loop over prepared argument blocks instead of one Python-dispatched ctypes call each (bit-identical: the same entry
# points with the same arguments).  False = launch by launch from Python (bench.py's per-convolution diagnostic pass, A/B runs).
LOOP_OPLIST = True
# bench.py's hook for timing the lookup INSIDE the timed region: a callable iters -> 2 * iters raw hipEvent_t handles (ints), recorded in
# front of and behind each iteration's lookup by the launch list itself; None = the list's timing cells stay empty (no-ops)
LOOKUP_EVENT_SINK = None
# Sequential tracking encodes 1-3 images and runs one or two flow pairs per frame: ~75 launches around the update loop whose Python dispatch
# (argument checks, descriptor construction, output allocation) costs more host time than the loop's list.  For such small passes an
# encoder pass and the two ends of RAFT.forward are RECORDED once (ops.Recorder: an ordinary pass that also logs its launches and keeps its
# intermediates as a private workspace) and replayed as launch lists with the input / output pointers rewritten.  Bit-identical (the same
# entry points with the same arguments); keyed on shapes, stream and the weights' versions.  Larger passes (batch mode) keep the
# call-by-call route: their host time is hidden behind 60 ms of kernels and their intermediates would pin gigabytes.
FRAME_OPLISTS = True
# The correlation lookup fused into convc1 (rpe_corr_lookup_conv1x1: the 324-channel lookup result never leaves LDS; bit-identical to the
# two kernels) in the update loop's launch list, for passes of at most this many workgroups (64 queries each): ONE round of workgroups on
# the chip.  Measured at 640x512 (tools/bench_lookup_conv.py; fused vs lookup + convc1 back to back): 1 pair 33 vs 33 us, 2 pairs (a tracker
# frame: 160 workgroups) 35 vs 45, 4 pairs 67 vs 66, 8 pairs 103 vs 106, 16 pairs 171 vs 169, 32 pairs 331 vs 313 -- a workgroup takes
# 33 us wherever it runs (20 us of matrix instructions behind a lookup phase that nothing overlaps: its 150 KB of LDS give it the CU to
# itself), so the fusion pays exactly where the two kernels leave most of the chip idle.
LOOKUP_FUSED = True
LOOKUP_FUSED_MAX_WGS = 256
FRAME_OPLISTS_MAX_IMAGES = 4                                      # images per encoder pass / flow pairs per RAFT pass up to which passes are recorded
ENC_STREAMS = True                                                 # RAFT.encode_both: the context encoder on the side stream beside the feature encoder
ENC_STREAMS_MIN = 8                                                # ... for at least this many context images (below: the fork / join costs more than it hides)
SIDE_STREAM_MAX = 8 * 5120                                     # queries per pass (batch * h/8 * w/8) up to which the side stream is used


def _norm(kind, ch):
    if kind == 'batch':
        return nn.BatchNorm2d(ch)
    if kind == 'instance':
        return nn.InstanceNorm2d(ch)
    if kind == 'none':
        return nn.Sequential()
    raise ValueError(kind)


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn='batch', stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = _norm(norm_fn, planes)
        self.norm2 = _norm(norm_fn, planes)
        if stride == 1:
            self.downsample = None
        else:
            self.norm3 = _norm(norm_fn, planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)

    def takes_raw_input(self, x):
        """Can this block read its input as the RAW output of an instance-normalised convolution (+ its (mean, 1/std)), i.e. run
        conv1 with the loader-side normalisation and normalise the shortcut inside the last pass?  (fnet's first block behind the stem.)"""
        return (isinstance(self.norm1, nn.InstanceNorm2d) and self.downsample is None and _fusable(self.conv1, x) and x.shape[-1] % 4 == 0
                and _wino(self.conv1, x) is not None)

    def forward(self, x, x_norm=None):
        """relu(x' + relu(norm2(conv2(relu(norm1(conv1 x)))))), x' = x or norm3(conv1x1 x); every convolution with its
        (bias, norm, ReLU, residual) epilogue fused (conv_norm_act).  With instance norm (fnet) norm1 + ReLU never exist
        as a tensor: conv1 leaves its raw output and partial sums, and conv2 normalises that input while staging it.
        ``x_norm`` (b,c,2): x is a raw convolution output whose relu((x - mean) * inv) is this block's real input (takes_raw_input)."""
        fused_in = isinstance(self.norm1, nn.InstanceNorm2d) and _fusable(self.conv1, x) and x.shape[-1] // self.conv1.stride[0] % 4 == 0
        if x_norm is not None and not (fused_in and self.takes_raw_input(x)):
            raise RuntimeError('ResidualBlock: x_norm needs the fused stride-1 Winograd route')
        if fused_in:
            b, _, hh, ww = x.shape
            st = self.conv1.stride[0]
            w1 = _wino(self.conv1, x)
            raw1 = torch.empty(b, self.conv1.out_channels, hh // st, ww // st, device=x.device)
            if w1 is not None:                                 # stride 1: Winograd, moments per 16x8-pixel patch
                stats1 = ops.conv_wino_stats_buffer(b, self.conv1.out_channels, hh, ww, x.device)
                ops.conv_wino(x, w1, ops.CONV_LINEAR, raw1, bias=self.conv1.bias.detach(), stats=stats1, pre_norm=x_norm)
            else:
                stats1 = ops.conv_stats_buffer(b, self.conv1.out_channels, hh, ww, x.device, stride=st)
                ops.conv_fused(x, _packed(self.conv1), ops.CONV_LINEAR, raw1, bias=self.conv1.bias.detach(), stats=stats1, stride=st)
            mi = ops.instnorm_finalize(stats1, (hh // st) * (ww // st), eps=self.norm1.eps, channels=self.conv1.out_channels)
            res_relu = True
            if self.downsample is not None:
                sc = self.downsample[0]
                if _fusable(sc, x):
                    # the shortcut norm3(conv1x1 x) never exists as a tensor either: its raw convolution output and (mean, 1/std) go to
                    # the block's last pass, which normalises it on the fly (bit-identical to a pass of its own, tested)
                    stats_sc = ops.conv_stats_buffer(b, sc.out_channels, hh, ww, x.device, stride=st)
                    x = ops.conv_fused(x, _packed(sc), ops.CONV_LINEAR, torch.empty(b, sc.out_channels, hh // st, ww // st, device=x.device),
                                       bias=sc.bias.detach(), stats=stats_sc, stride=st)
                    x_norm = ops.instnorm_finalize(stats_sc, (hh // st) * (ww // st), eps=self.norm3.eps, channels=sc.out_channels)
                    res_relu = False
                else:
                    x = conv_norm_act(sc, self.norm3, x, relu=False)
            ho, wo = raw1.shape[-2:]
            w2 = _wino(self.conv2, raw1)
            raw2 = torch.empty(b, self.conv2.out_channels, ho, wo, device=x.device)
            if w2 is not None:
                stats2 = ops.conv_wino_stats_buffer(b, self.conv2.out_channels, ho, wo, x.device)
                ops.conv_wino(raw1, w2, ops.CONV_LINEAR, raw2, bias=self.conv2.bias.detach(), stats=stats2, pre_norm=mi)
            else:
                stats2 = ops.conv_stats_buffer(b, self.conv2.out_channels, ho, wo, x.device)
                ops.conv_fused(raw1, _packed(self.conv2), ops.CONV_LINEAR, raw2, bias=self.conv2.bias.detach(), stats=stats2, pre_norm=mi)
            if isinstance(stats2, ops.TileMajorStats):            # merge the Winograd kernel's records once (a small launch), then stream
                stats2 = ops.instnorm_finalize(stats2, ho * wo, eps=self.norm2.eps, channels=self.conv2.out_channels)
            return ops.instnorm_apply(raw2, stats2, eps=self.norm2.eps, relu=True, residual=x, residual_norm=x_norm, residual_relu=res_relu)
        y = conv_norm_act(self.conv1, self.norm1, x, relu=True)
        if self.downsample is not None:
            x = conv_norm_act(self.downsample[0], self.norm3, x, relu=False)
        return conv_norm_act(self.conv2, self.norm2, y, relu=True, residual=x)


def _bounded_put(cache, key, value, keep=2):
    """Insert into a small insertion-ordered cache, evicting the oldest entries (and the buffers / launch descriptors they pin)."""
    while len(cache) >= keep:
        cache.pop(next(iter(cache)))
    cache[key] = value


def _bn_affine(conv, norm):
    """Eval-mode BatchNorm2d folded with the conv bias: y = conv_nobias(x) * scale + shift (cached per module)."""
    key = tuple(t._version for t in (conv.bias, norm.weight, norm.bias, norm.running_mean, norm.running_var)) + (norm.weight.data_ptr(),)
    cached = getattr(norm, '_rpe_affine', None)
    if cached is None or cached[0] != key:
        scale = (norm.weight / torch.sqrt(norm.running_var + norm.eps)).detach().float().contiguous()
        shift = ((conv.bias - norm.running_mean) * scale + norm.bias).detach().float().contiguous()
        norm._rpe_affine = cached = (key, scale, shift)
    return cached[1], cached[2]


def _packed(conv):
    """PackedConv of a module's weight (no bias), cached on the module until the weight changes."""
    key = (conv.weight._version, conv.weight.data_ptr())
    cached = getattr(conv, '_rpe_packed', None)
    if cached is None or cached[0] != key:
        conv._rpe_packed = cached = (key, ops.PackedConv(conv.weight, None))
    return cached[1]


def _wino(conv, x):
    """PackedWino of a 3x3 stride-1 convolution when rpe_conv_wino can run it on this input (cached on the module), else None."""
    if not WINOGRAD or conv.stride != (1, 1) or conv.kernel_size != (3, 3) or conv.padding != (1, 1):
        return None
    if not ops.PackedWino.supported(conv.weight, x.shape[-2], x.shape[-1]) or conv.in_channels > 128 or not x.is_contiguous():
        return None
    key = (conv.weight._version, conv.weight.data_ptr())
    cached = getattr(conv, '_rpe_wino', None)
    if cached is None or cached[0] != key:
        conv._rpe_wino = cached = (key, ops.PackedWino(conv.weight, None))
    return cached[1]


def _fusable(conv, x):
    """Can rpe_conv_fused / rpe_conv_wino run this encoder convolution on this input?  (3x3 stride 1; 3x3 pad 1 / 1x1 stride 2 on even
    maps; rows of whole 16-byte quads.)  Every launch size takes this route: the kernels pick smaller tiles for small launches
    themselves, and their results do not depend on that choice (tests/test_gpu_conv.py::*_agree_bitwise)."""
    _, _, hh, ww = x.shape
    s1 = conv.stride == (1, 1) and conv.kernel_size == (3, 3) and conv.padding == (1, 1)
    s2 = conv.stride == (2, 2) and hh % 2 == 0 and ww % 2 == 0 and \
        ((conv.kernel_size == (3, 3) and conv.padding == (1, 1)) or (conv.kernel_size == (1, 1) and conv.padding == (0, 0)))
    stride = 2 if s2 else 1
    return (s1 or s2) and ww % 4 == 0 and ((hh // stride) * (ww // stride)) % 4 == 0 and x.is_contiguous()


def conv_norm_act(conv, norm, x, relu, residual=None):
    """norm(conv(x) + bias) [ReLU] [+ residual, ReLU].  The residual blocks' convolutions -- 3x3 stride 1, 3x3 stride 2 and
    the 1x1 stride-2 shortcut -- run on the fused HIP implicit GEMM when the map width is a multiple of 4 (folded batch
    norm / ReLU / residual inside its epilogue; for instance norm the epilogue leaves per-tile partial sums and one more
    read+write pass normalises); shapes those kernels refuse run on the generic kernel (ops.conv_direct) + a stand-alone epilogue pass."""
    b, _, hh, ww = x.shape
    fused = _fusable(conv, x)
    stride = conv.stride[0] if fused else 1
    if isinstance(norm, nn.BatchNorm2d):
        if norm.training:
            raise RuntimeError('the RAFT encoders run with frozen batch norm (RAFT.freeze_bn, pose_net.py:22)')
        scale, shift = _bn_affine(conv, norm)
        pw = _wino(conv, x) if fused else None
        if pw is not None:                                     # cnet's stride-1 3x3 layers: Winograd with the folded batch norm in the epilogue
            out = torch.empty(b, conv.out_channels, hh, ww, device=x.device)
            return ops.conv_wino(x, pw, ops.CONV_RELU if relu else ops.CONV_LINEAR, out, scale=scale, bias=shift, residual=residual)
        if fused:
            out = torch.empty(b, conv.out_channels, hh // stride, ww // stride, device=x.device)
            return ops.conv_fused(x, _packed(conv), ops.CONV_RELU if relu else ops.CONV_LINEAR, out, scale=scale, bias=shift, residual=residual,
                                  stride=stride)
        return ops.affine_act(ops.conv_direct(x, conv.weight, None, conv.stride, conv.padding), scale, shift, relu=relu, residual=residual)
    if isinstance(norm, nn.InstanceNorm2d):
        pw = _wino(conv, x) if fused else None
        if pw is not None:
            stats = ops.conv_wino_stats_buffer(b, conv.out_channels, hh, ww, x.device)
            pre = ops.conv_wino(x, pw, ops.CONV_LINEAR, torch.empty(b, conv.out_channels, hh, ww, device=x.device), bias=conv.bias.detach(), stats=stats)
            return ops.instnorm_apply(pre, stats, eps=norm.eps, relu=relu, residual=residual)
        if fused:
            stats = ops.conv_stats_buffer(b, conv.out_channels, hh, ww, x.device, stride=stride)
            pre = ops.conv_fused(x, _packed(conv), ops.CONV_LINEAR, torch.empty(b, conv.out_channels, hh // stride, ww // stride, device=x.device),
                                 bias=conv.bias.detach(), stats=stats, stride=stride)
            return ops.instnorm_apply(pre, stats, eps=norm.eps, relu=relu, residual=residual)
        return ops.instnorm_act(ops.conv_direct(x, conv.weight, None, conv.stride, conv.padding), conv.bias, eps=norm.eps, relu=relu, residual=residual)
    raise NotImplementedError(type(norm))


def _key_sources(module):
    """(dict, name) of every parameter / buffer slot below ``module``: the slots are read afresh for every key (a Parameter object that is
    REPLACED shows up, not only one that is updated in place), the walk over the module tree is done once."""
    src = []
    for m in module.modules():
        src += [(m._parameters, n) for n in m._parameters] + [(m._buffers, n) for n in m._buffers]
    return src


def _tensor_key(sources):
    """Versions and addresses of the tensors in the slots of _key_sources (in-place updates bump the version, .to() / .float() move the
    data, an assignment puts another tensor into the slot)."""
    ts = [d[n] for d, n in sources]
    return tuple((t._version, t.data_ptr()) for t in ts if t is not None)


def _overlap(tensors):
    spans = sorted((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()) for t in tensors)
    return any(a[1] > b[0] for a, b in zip(spans, spans[1:]))


class _Recorded:
    """A small cache of recorded passes (ops.Recorder) keyed by the caller; a key whose recording came out incomplete three times (a launch
    the recorder cannot carry: a fallback route, a weight packing made on first use) is given up on and stays on the call-by-call route."""

    def __init__(self, keep=4):
        self._progs, self._failed, self._keep = {}, {}, keep

    def get(self, key):
        return self._progs.get(key)

    def wanted(self, key):
        return self._failed.get(key, 0) < 3

    def put(self, key, rec):
        if rec.complete:
            _bounded_put(self._progs, key, rec, keep=self._keep)
        else:
            if len(self._failed) > 64:
                self._failed.clear()
            self._failed[key] = self._failed.get(key, 0) + 1

    def clear(self):
        self._progs.clear()
        self._failed.clear()


class BasicEncoder(nn.Module):
    def __init__(self, output_dim=128, norm_fn='batch', dropout=0.0):
        super().__init__()
        self.norm_fn = norm_fn
        self.norm1 = _norm(norm_fn, 64)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.in_planes = 64
        self.layer1 = self._make_layer(64, stride=1)
        self.layer2 = self._make_layer(96, stride=2)
        self.layer3 = self._make_layer(128, stride=2)
        self.conv2 = nn.Conv2d(128, output_dim, kernel_size=1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def _make_layer(self, dim, stride=1):
        l1 = ResidualBlock(self.in_planes, dim, self.norm_fn, stride=stride)
        l2 = ResidualBlock(dim, dim, self.norm_fn, stride=1)
        self.in_planes = dim
        return nn.Sequential(l1, l2)

    def _stem_pack(self):
        """conv1 + norm1 + ReLU run on the RAW 0..255 image: the normalisation 2*(x/255)-1 happens while the kernel stages its
        input patch (rpe_stem_conv)."""
        key = (self.conv1.weight._version, self.conv1.weight.data_ptr())
        if getattr(self, '_stem_packed', None) is None or self._stem_packed[0] != key:
            self._stem_packed = (key, ops.PackedStem(self.conv1.weight))
        return self._stem_packed[1]

    def _stem_many(self, images):
        """The stem over several image batches, written into batch slices of ONE output (no torch.cat of the inputs)."""
        ps = self._stem_pack()
        n = sum(im.shape[0] for im in images)
        hh, ww = images[0].shape[-2:]
        out = torch.empty(n, 64, hh // 2, ww // 2, device=images[0].device)
        bn = isinstance(self.norm1, nn.BatchNorm2d)
        if bn:
            if self.norm1.training:
                raise RuntimeError('the RAFT encoders run with frozen batch norm (RAFT.freeze_bn, pose_net.py:22)')
            scale, shift = _bn_affine(self.conv1, self.norm1)
        stats = None if bn else torch.empty(n, 64, ops.lib().rpe_stem_tiles(hh, ww, 2), 3, device=out.device)
        i = 0
        for im in images:
            k = im.shape[0]
            if bn:
                ops.stem_conv(im.contiguous(), ps, bias=shift, scale=scale, relu=True, out=out[i:i + k])
            else:
                ops.stem_conv(im.contiguous(), ps, bias=self.conv1.bias.detach(), relu=False, stats=stats[i:i + k], out=out[i:i + k])
            i += k
        if bn:
            return out, None
        if self.layer1[0].takes_raw_input(out):               # norm1 + ReLU happen inside layer1's first block (its loader and its last pass)
            return out, ops.instnorm_finalize(stats, (hh // 2) * (ww // 2), eps=self.norm1.eps, channels=64)
        return ops.instnorm_apply(out, stats, eps=self.norm1.eps, relu=True), None

    def _final(self, x, split_act):
        """conv2, the 1x1 output layer (128 -> output_dim), on the implicit GEMM.  ``split_act`` (the context encoder as RAFT uses
        it, core/RAFT/core/raft.py: net, inp = split(cnet); net = tanh(net); inp = relu(inp)): channels [0, 128) leave through
        tanh, the rest through ReLU, in the convolution's epilogue."""
        m = self.conv2
        b, _, hh, ww = x.shape
        half = m.out_channels // 2
        if ww % 4 == 0 and x.is_contiguous():
            key = (m.weight._version, m.weight.data_ptr(), m.bias._version)
            cached = getattr(self, '_final_packed', None)
            if cached is None or cached[0] != key:
                w, bv = m.weight.detach(), m.bias.detach()
                self._final_packed = cached = (key, ops.Conv1x1(w, bv), ops.Conv1x1(w[:half].contiguous(), bv[:half].contiguous()),
                                               ops.Conv1x1(w[half:].contiguous(), bv[half:].contiguous()))
            out = torch.empty(b, m.out_channels, hh, ww, device=x.device)
            if not split_act:
                return cached[1](x, ops.CONV_LINEAR, out, x3=CONV_BF16X3)
            cached[2](x, ops.CONV_TANH, out[:, :half])
            cached[3](x, ops.CONV_RELU, out[:, half:], x3=CONV_BF16X3)
            return out
        y = ops.conv_direct(x, m.weight, m.bias, 1, 0)           # (maps whose rows are not whole 16-byte quads)
        if split_act:
            y[:, :half] = torch.tanh(y[:, :half])
            ops.bias_act(y[:, half:].contiguous(), None, relu=True, out=y, out_offset=half)
        return y

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.__dict__.pop('_key_tensors', None)                # .to() / .float(): the recorded passes point at the old storage
        if getattr(self, '_recorded', None) is not None:
            self._recorded.clear()
        return r

    def forward(self, x, raw255=False, split_act=False):
        """``raw255``: x is the raw 0..255 image (RAFT.forward's normalisation is then done inside the first kernel), or a list of
        such image batches, encoded as one batch.  ``split_act``: see _final.
        Small inference passes (FRAME_OPLISTS) are recorded once per (shapes, stream, weights) and replayed as one launch list."""
        images = list(x) if isinstance(x, (list, tuple)) else [x]
        first = images[0]
        if FRAME_OPLISTS and raw255 and first.is_cuda and first.device.index == torch.cuda.current_device() and not torch.is_grad_enabled() \
                and sum(im.shape[0] for im in images) <= FRAME_OPLISTS_MAX_IMAGES \
                and all(im.is_contiguous() and im.dtype == torch.float32 and im.device == first.device for im in images) and not _overlap(images):
            if getattr(self, '_recorded', None) is None:
                self._recorded = _Recorded()
            if '_key_tensors' not in self.__dict__:
                self.__dict__['_key_tensors'] = _key_sources(self)
            key = (tuple(tuple(im.shape) for im in images), first.device.index, ops.raw_stream(), split_act, WINOGRAD, CONV_BF16X3, self.training,
                   _tensor_key(self.__dict__['_key_tensors']))
            rec = self._recorded.get(key)
            if rec is not None:
                out = torch.empty(rec.out_shape, device=first.device)
                rec.replay({**{i: im for i, im in enumerate(images)}, 'out': out})
                return out
            if self._recorded.wanted(key):
                rec = ops.Recorder()
                with rec:
                    out = self._forward(x, raw255, split_act)
                if rec.complete and out.is_contiguous():
                    for i, im in enumerate(images):
                        rec.complete = rec.complete and rec.bind(i, im) > 0
                    rec.complete = rec.complete and rec.bind('out', out) > 0
                    rec.out_shape = tuple(out.shape)
                self._recorded.put(key, rec)
                return out
        return self._forward(x, raw255, split_act)

    def _forward(self, x, raw255=False, split_act=False):
        many = isinstance(x, (list, tuple))
        first = x[0] if many else x
        hh, ww = first.shape[-2:]
        stem_ok = raw255 and hh % 2 == 0 and ww % 2 == 0 and ((hh // 2) * (ww // 2)) % 4 == 0 and isinstance(self.norm1, (nn.BatchNorm2d, nn.InstanceNorm2d))
        x_norm = None
        if stem_ok:
            x, x_norm = self._stem_many(list(x) if many else [x])
        else:
            x = (torch.cat(list(x), dim=0) if many else x).contiguous()
            x = conv_norm_act(self.conv1, self.norm1, 2 * (x / 255.0) - 1.0 if raw255 else x, relu=True)
        x = self.layer1[1](self.layer1[0](x, x_norm))
        x = self.layer3(self.layer2(x))
        return self._final(x, split_act)


class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.conv2(self.relu(self.conv1(x)))


class SepConvGRU(nn.Module):
    """Parameters as upstream; the forward lives in BasicUpdateBlock.step (fused with the HIP gate kernels)."""

    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        c = hidden_dim + input_dim
        self.convz1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convr1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convq1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convz2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convr2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convq2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))


class BasicMotionEncoder(nn.Module):
    def __init__(self, corr_levels=4, corr_radius=4):
        super().__init__()
        cor_planes = corr_levels * (2 * corr_radius + 1) ** 2
        self.convc1 = nn.Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)

    def _calls(self, corr, cat_buf, hx, rhx, packed):
        """Prepared launchers of the fused route for one buffer set (descriptors checked once; the 12 iterations reuse them)."""
        key = (corr.data_ptr(), cat_buf.data_ptr(), hx.data_ptr(), rhx.data_ptr(), tuple(corr.shape))
        cache = packed.setdefault('_enc_calls', {})               # one entry per workspace (a tracker alternates between batch n and 2n)
        calls = cache.get(key)
        if calls is None:
            cor = packed['cor_buf'](corr)
            flo = packed['flo_buf'](corr)
            wino = dict(packed['wino']) if corr.shape[-1] % 2 == 0 and corr.shape[-2] % 2 == 0 else {}
            if wino and corr.shape[-1] % 4 == 0:
                wino.update(packed.get('wino_x3', {}))          # (CONV_BF16X3: conv_wino runs the kernel that belongs to the packing)

            def c3(name, x, out, out2=None):            # a 3x3 layer: Winograd when available, else the direct implicit GEMM
                if name in wino:
                    return ops.conv_wino(x, wino[name], ops.CONV_RELU, out, out2=out2, prepare=True)
                return ops.conv_fused(x, packed[name], ops.CONV_RELU, out, out2=out2, prepare=True)
            calls = (key, cor, flo,
                     packed['convc1_1x1'](corr, ops.CONV_RELU, cor, prepare=True, x3=CONV_BF16X3),
                     c3('convc2', cor, cat_buf[:, :192]), c3('convf2', flo, cat_buf[:, 192:]),
                     c3('conv', cat_buf, hx[:, 128:254], rhx[:, 128:254]))
            _bounded_put(cache, key, calls, keep=4)
        return calls

    def flow_branch_launchers(self, flow, corr, cat_buf, hx, rhx, packed):
        """(convf1, convf2) as prepared launchers on these buffers (``flow`` must be the persistent flow buffer of the workspace)."""
        _, cor, flo_buf, c1, c2, f2, cv_ = self._calls(corr, cat_buf, hx, rhx, packed)
        key = (flow.data_ptr(), flo_buf.data_ptr())
        cache = packed.setdefault('_f1_calls', {})
        f1 = cache.get(key)
        if f1 is None:
            f1 = ops.stem_conv(flow, packed['convf1'], bias=self.convf1.bias.detach(), relu=True, div=1.0, mul=1.0, sub=0.0, out=flo_buf, prepare=True)
            _bounded_put(cache, key, f1, keep=4)
        return f1, f2

    def flow_branch(self, flow, corr, cat_buf, hx, rhx, packed):
        """convf1 -> convf2 (fused route): depends on the flow only, not on the correlation lookup, so RAFT.forward may run it on a
        side stream beside lookup -> convc1 -> convc2."""
        _, cor, flo_buf, c1, c2, f2, cv_ = self._calls(corr, cat_buf, hx, rhx, packed)
        ops.stem_conv(flow, packed['convf1'], bias=self.convf1.bias.detach(), relu=True, div=1.0, mul=1.0, sub=0.0, out=flo_buf)   # 7x7 on 2 channels
        f2()

    def forward(self, flow, corr, cat_buf, hx, rhx, packed=None, flow_in_place=False, flow_branch_done=None):
        """Writes relu(conv(cat[cor, flo])) (126 ch) and flow (2 ch) into channels [128,256) of hx and rhx.
        ``packed`` (BasicUpdateBlock.packed_convs) selects the fused HIP convolutions (conv + bias + ReLU + cat in
        one kernel each); otherwise the generic convolution (ops.conv_direct) runs without bias and rpe_bias_act does the rest.
        ``flow_branch_done``: an event recorded behind flow_branch() on another stream (it then is not run here)."""
        def cv(m, x):
            return ops.conv_direct(x, m.weight, None, m.stride, m.padding)
        if packed is not None:
            _, cor, flo_buf, c1, c2, f2, cv_ = self._calls(corr, cat_buf, hx, rhx, packed)
            c1(); c2()
            if flow_branch_done is None:
                self.flow_branch(flow, corr, cat_buf, hx, rhx, packed)
            else:
                torch.cuda.current_stream().wait_event(flow_branch_done)
            cv_()
        else:
            cor = ops.bias_act(cv(self.convc1, corr), self.convc1.bias)
            ops.bias_act(cv(self.convc2, cor), self.convc2.bias, out=cat_buf, out_offset=0)
            flo = ops.bias_act(cv(self.convf1, flow), self.convf1.bias)
            ops.bias_act(cv(self.convf2, flo), self.convf2.bias, out=cat_buf, out_offset=192)
            ops.bias_act(cv(self.conv, cat_buf), self.conv.bias, out=hx, out_offset=128, out2=rhx, out2_offset=128)
        if not flow_in_place:                                   # (the fused flow head has already written flow behind the motion features)
            ops.bias_act(flow, None, relu=False, out=hx, out_offset=254, out2=rhx, out2_offset=254)


class BasicUpdateBlock(nn.Module):
    def __init__(self, corr_levels=4, corr_radius=4, hidden_dim=128):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.encoder = BasicMotionEncoder(corr_levels, corr_radius)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(256, 64 * 9, 1, padding=0))
        self._stacked = None

    def gate_weights(self):
        """GRU conv weights re-arranged once (cached until a parameter changes):
          * convz|convr stacked on the output-channel axis (they read the same input),
          * input channels split into the loop-varying part [h | motion | flow] (256 ch) and the context part
            `inp` (128 ch, identical in every GRU iteration).
        Returns dict name -> (w_var, w_ctx, bias) for 'zr1', 'q1', 'zr2', 'q2'."""
        g = self.gru
        gp = [t for m in (g.convz1, g.convr1, g.convq1, g.convz2, g.convr2, g.convq2) for t in (m.weight, m.bias)]
        key = tuple(p._version for p in gp) + tuple(p.data_ptr() for p in gp)
        if self._stacked is None or self._stacked[0] != key:
            c = self.hidden_dim

            def split(w):                      # (out, 384, kh, kw) -> varying (out,256,..), context (out,128,..)
                return torch.cat((w[:, :c], w[:, 2 * c:]), 1).detach().contiguous(), w[:, c:2 * c].detach().contiguous()
            W = {}
            for name, convs in (('zr1', (g.convz1, g.convr1)), ('q1', (g.convq1,)), ('zr2', (g.convz2, g.convr2)), ('q2', (g.convq2,))):
                w = torch.cat([m.weight for m in convs], 0)
                bvec = torch.cat([m.bias for m in convs], 0).detach().contiguous()
                W[name] = (*split(w), bvec)
            self._stacked = (key, W)
        return self._stacked[1]

    def context_terms(self, inp, out=None):
        """conv(inp; context channels) + bias of the four GRU convolutions: loop-invariant, computed once per pass.
        ``out`` = dict of persistent destination buffers (RAFT._workspace): the fused HIP convolution writes into them, so
        the GRU loop's prepared launchers see the same addresses pass after pass."""
        W = self.gate_weights()
        P = self.packed_convs(inp.shape[-1]) if out is not None else None
        if P is not None:
            for k in W:
                (ops.conv_wino1d if isinstance(P['ctx_' + k], ops.PackedWino1d) else ops.conv_fused)(inp, P['ctx_' + k], ops.CONV_LINEAR, out[k])
            return out
        pads = {'zr1': (0, 2), 'q1': (0, 2), 'zr2': (2, 0), 'q2': (2, 0)}
        return {k: ops.conv_direct(inp, W[k][1], W[k][2], 1, pads[k]) for k in W}

    def packed_convs(self, width):
        """Weights of the update block's convolutions in rpe_conv_fused's layout (cached until a parameter changes), or
        None when the fused kernels do not apply (map width not a multiple of 4)."""
        if width % 4 != 0:
            return None
        e, fh = self.encoder, self.flow_head
        mods = (e.convc1, e.convc2, e.convf2, e.conv, fh.conv1)
        keymods = mods + (e.convf1,)
        # (the convolutions' own weight / bias attributes: walking module.parameters() costs ~100 us a call, and this runs 26 times a frame)
        ps = [t for m in keymods for t in (m.weight, m.bias) if t is not None]
        key = tuple(p._version for p in ps) + tuple(p.data_ptr() for p in ps) + (id(self.gate_weights()), CONV_BF16X3, X3_GRU)
        if getattr(self, '_packed', None) is None or self._packed[0] != key:
            W = self.gate_weights()
            P = {n: ops.PackedConv(m.weight, m.bias) for n, m in zip(('convc1', 'convc2', 'convf2', 'conv', 'fh1'), mods)}
            # the four 3x3 layers also in Winograd form (rpe_conv_wino: 2.25x fewer matrix FLOPs); used on even maps
            P['wino'] = {n: ops.PackedWino(m.weight, m.bias) for n, m in zip(('convc2', 'convf2', 'conv', 'fh1'), mods[1:])
                         if ops.PackedWino.supported(m.weight, 2, 2)} if WINOGRAD else {}
            # the labelled bf16x3 variant's packings of the same layers (used on maps rpe_conv_wino_x3 accepts: even h, w % 4 == 0)
            P['wino_x3'] = {n: ops.PackedWinoX3(m.weight, m.bias) for n, m in zip(('convc2', 'convf2', 'conv', 'fh1'), mods[1:])
                            if n != 'convf2' and m.in_channels >= X3_MIN_CIN and ops.PackedWinoX3.supported(m.weight, 2, 4)} if (WINOGRAD and CONV_BF16X3) else {}
            P['convf1'] = ops.PackedStem(e.convf1.weight)
            P['convc1_1x1'] = ops.Conv1x1(e.convc1.weight, e.convc1.bias)
            for n in ('zr1', 'q1', 'zr2', 'q2'):
                # the loop-varying 256 channels: Winograd F(4,5) along the filter axis (rpe_conv_wino1d: 2.5x fewer matrix FLOPs), else the
                # direct implicit GEMM; bias is part of the context term (context_terms)
                # (under CONV_BF16X3 with X3_GRU: the labelled variant's packing -- rpe_conv_wino1d_x3; packed_convs only runs for widths it accepts)
                P[n] = (ops.PackedWino1dX3 if CONV_BF16X3 and X3_GRU and ops.PackedWino1dX3.supported(W[n][0], 4) else ops.PackedWino1d)(W[n][0]) if WINOGRAD \
                    else ops.PackedConv(W[n][0])
                P['ctx_' + n] = (ops.PackedWino1d if WINOGRAD else ops.PackedConv)(W[n][1], W[n][2])
            scratch = {}

            def buf(name, like, c):                            # scratch for intermediate activations, one set per workspace (``like`` is one of
                k = (name, like.data_ptr(), like.shape[0], c, like.shape[2], like.shape[3], like.device)   # RAFT._workspace's buffers)
                if k not in scratch:
                    _bounded_put(scratch, k, torch.empty(like.shape[0], c, like.shape[2], like.shape[3], device=like.device), keep=12)
                return scratch[k]
            P['cor_buf'] = lambda like: buf('cor', like, 256)
            P['fh_buf'] = lambda like: buf('fh', like, 256)
            P['flo_buf'] = lambda like: buf('flo', like, 128)
            self._packed = (key, P)
        return self._packed[1]

    def gru_launchers(self, hx, rhx, z_buf, ctx, flow, coords1, in_place, P):
        """Prepared launchers of the GRU, FlowHead.conv1 and (``in_place``) the flow head's output layer with the coords / flow bookkeeping,
        on these buffers (descriptors checked once; every iteration of every pass on the same workspace reuses them)."""
        c = self.hidden_dim
        fh = self.flow_head
        # each GRU half = two implicit-GEMM convolutions whose epilogues are the gates:
        #   z = s(convz hx + ctx), r*h -> rhx ;  h <- (1-z) h + z tanh(convq rhx + ctx)   (in place on hx[:, :c])
        key = (hx.data_ptr(), rhx.data_ptr(), z_buf.data_ptr(), tuple(ctx[k].data_ptr() for k in ('zr1', 'q1', 'zr2', 'q2')), tuple(hx.shape),
               (coords1.data_ptr(), flow.data_ptr()) if in_place else None)
        cache = P.setdefault('_gru_calls', {})
        calls = cache.get(key)
        if calls is None:
            seq = []
            gconv = ops.conv_wino1d if isinstance(P['zr1'], (ops.PackedWino1d, ops.PackedWino1dX3)) else ops.conv_fused
            for zr, q in (('zr1', 'q1'), ('zr2', 'q2')):
                seq.append(gconv(hx, P[zr], ops.CONV_GATE_ZR, z_buf, out2=rhx[:, :c], add=ctx[zr], hidden=hx[:, :c], gate_channels=c,
                                 prepare=True))
                seq.append(gconv(rhx, P[q], ops.CONV_GATE_H, hx[:, :c], add=ctx[q], hidden=hx[:, :c], zgate=z_buf, prepare=True))
            if 'fh1' in P['wino'] and hx.shape[-1] % 2 == 0 and hx.shape[-2] % 2 == 0:
                pfh = P['wino_x3']['fh1'] if 'fh1' in P.get('wino_x3', {}) and hx.shape[-1] % 4 == 0 else P['wino']['fh1']
                seq.append(ops.conv_wino(hx[:, :c], pfh, ops.CONV_RELU, P['fh_buf'](hx), prepare=True))
            else:
                seq.append(ops.conv_fused(hx[:, :c], P['fh1'], ops.CONV_RELU, P['fh_buf'](hx), prepare=True))
            if in_place:
                seq.append(ops.flow_update(P['fh_buf'](hx), fh.conv2.weight, fh.conv2.bias.detach(), coords1, coords1, flow_out=flow,
                                           dst1=hx[:, 2 * c - 2:], dst2=rhx[:, 2 * c - 2:], prepare=True))
            calls = (key, seq)
            _bounded_put(cache, key, calls, keep=4)
        return calls[1]

    def step(self, hx, rhx, z_buf, cat_buf, h_buf, ctx, corr, flow, coords1, in_place=False, flow_branch_done=None):
        """One update.  hx = (h | motion | flow), rhx = (r*h | motion | flow), both (b,256,h,w); ctx = context_terms().
        Returns coords1 + delta_flow; the new hidden state is left in hx[:, :128] (h_buf is written by the generic
        path only: the fused flow head reads the slice directly).  ``in_place`` (fused route only): ``coords1`` and ``flow`` are
        persistent buffers; the flow head's output layer updates coords1 in place and writes flow = coords1 - grid into ``flow``
        and behind the motion features of hx / rhx, so the loop runs without subtract / copy launches."""
        c = self.hidden_dim
        P = self.packed_convs(hx.shape[-1])
        if in_place and P is None:
            raise RuntimeError('in_place update needs the fused route')
        self.encoder(flow, corr, cat_buf, hx, rhx, packed=P, flow_in_place=in_place, flow_branch_done=flow_branch_done)
        fh = self.flow_head
        if P is not None:
            calls = (None, self.gru_launchers(hx, rhx, z_buf, ctx, flow, coords1, in_place, P))
            if in_place:
                for launch in calls[1]:
                    launch()
                return coords1
            for launch in calls[1][:-1]:
                launch()
            t = calls[1][-1]()
            return ops.conv3x3_to2(t, fh.conv2.weight, fh.conv2.bias, add=coords1)  # coords1 + delta_flow
        W = self.gate_weights()
        # horizontal half: z = s(convz1 hx), r = s(convr1 hx), q = tanh(convq1 [r*h, x]), h = (1-z) h + z q
        zr = ops.conv_direct(hx, W['zr1'][0], None, 1, (0, 2))
        ops.gru_gates_zr(zr, hx, c, z_buf, rhx, add=ctx['zr1'])
        q = ops.conv_direct(rhx, W['q1'][0], None, 1, (0, 2))
        ops.gru_gates_h(z_buf, q, hx, c, hx, add=ctx['q1'])
        # vertical half
        zr = ops.conv_direct(hx, W['zr2'][0], None, 1, (2, 0))
        ops.gru_gates_zr(zr, hx, c, z_buf, rhx, add=ctx['zr2'])
        q = ops.conv_direct(rhx, W['q2'][0], None, 1, (2, 0))
        ops.gru_gates_h(z_buf, q, hx, c, hx, add=ctx['q2'])
        h_buf.copy_(hx[:, :c])                                            # contiguous h for the heads
        t = ops.conv_direct(h_buf, fh.conv1.weight, fh.conv1.bias, 1, 1, relu=True)
        return ops.conv3x3_to2(t, fh.conv2.weight, fh.conv2.bias, add=coords1)      # coords1 + delta_flow

    def up_mask(self, net):
        """.25 * mask(net) (upstream scales the mask to balance gradients).  The 3x3 layer + ReLU runs on the Winograd kernel, the
        factor is folded into the 1x1 layer's parameters (a power of two: bit-identical to scaling the result)."""
        c1, c2 = self.mask[0], self.mask[2]
        mp = (c1.weight, c1.bias, c2.weight, c2.bias)
        hh, ww = net.shape[-2:]
        x3 = CONV_BF16X3 and c1.in_channels >= X3_MIN_CIN and ops.PackedWinoX3.supported(c1.weight, hh, ww)
        key = tuple(p._version for p in mp) + tuple(p.data_ptr() for p in mp) + (x3,)
        cached = getattr(self, '_mask_packed', None)
        if cached is None or cached[0] != key:
            pw = (ops.PackedWinoX3 if x3 else ops.PackedWino)(c1.weight, c1.bias) if WINOGRAD and c1.weight.is_cuda else None
            w2, b2 = (0.25 * c2.weight).detach(), (0.25 * c2.bias).detach()
            self._mask_packed = cached = (key, pw, w2, b2, ops.Conv1x1(w2, b2) if c2.weight.is_cuda else None)
        _, pw, w2, b2, p2 = cached
        if pw is not None and not torch.is_grad_enabled() and hh % 2 == 0 and ww % 2 == 0:      # (net may be a channel slice: hx[:, :128])
            t = ops.conv_wino(net, pw, ops.CONV_RELU, torch.empty(net.shape[0], c1.out_channels, hh, ww, device=net.device))
        else:
            t = ops.conv_direct(net, c1.weight, c1.bias, 1, 1, relu=True)
        if p2 is not None and ww % 4 == 0 and not torch.is_grad_enabled():
            return p2(t, ops.CONV_LINEAR, torch.empty(net.shape[0], c2.out_channels, hh, ww, device=net.device), x3=CONV_BF16X3)
        return ops.conv_direct(t, w2, b2, 1, 0)


def coords_grid(batch, ht, wd, device):
    ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing='ij')
    return torch.stack((xs, ys), dim=0).float()[None].repeat(batch, 1, 1, 1)


class RAFT(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.get('small', False):
            raise NotImplementedError("RAFT-small is not on the reference's inference path (train.yaml:5 small: False)")
        self.config = config
        # upstream RAFT's ``mixed_precision`` runs the encoders under autocast, i.e. hands fp16 feature maps to the
        # correlation (BASELINE config 5 "fp16 features").  Here the encoders stay f32 and the feature maps are rounded to
        # fp16 where the correlation consumes them (16-bit matrix cores, f32 accumulation, f32 pyramid).
        self.mixed_precision = bool(config.get('mixed_precision', False))
        self.iters = int(config.get('iters', 12))
        self.hidden_dim = self.context_dim = 128
        self.corr_levels, self.corr_radius = 4, 4
        drop = config.get('dropout', 0.0)
        self.fnet = BasicEncoder(output_dim=256, norm_fn='instance', dropout=drop)
        self.cnet = BasicEncoder(output_dim=256, norm_fn='batch', dropout=drop)
        self.update_block = BasicUpdateBlock(self.corr_levels, self.corr_radius, hidden_dim=128)
        self._pyr = None
        self._ws = None

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def _side_stream(self, device):
        s = getattr(self, '_side', None)
        if s is None or s[0].device != device:
            with torch.cuda.device(device):
                self._side = s = (torch.cuda.Stream(device=device), torch.cuda.Event(), torch.cuda.Event())
        return s

    def _pyramid(self, b, h8, w8, device):
        p = self._pyr
        x3 = CORR_BF16X3 or CONV_BF16X3
        if p is None or (p.b, p.h8, p.w8) != (b, h8, w8) or p.buf.device != device or getattr(p, 'x3', False) != x3:
            self._pyr = ops.CorrPyramid(b, h8, w8, self.corr_levels, self.corr_radius, device=device, bf16x3=x3)
            self._pyr.x3 = x3
        return self._pyr

    def _workspace(self, n, h8, w8, device):
        """Activation buffers that live INSIDE a forward pass -- (h | motion | flow), (r*h | motion | flow), z, the motion
        encoder's concat buffer, the lookup output and the four context terms -- allocated once per (batch, map, device):
        the prepared launch descriptors of the GRU loop are keyed on these addresses, so they are built once and hit on
        every later pass (and nothing from a previous pass stays pinned besides this one set).  Single-stream use, like
        the rest of the module; tensors handed back to the caller (flows, hidden, context) are always fresh."""
        key = (n, h8, w8, str(device))
        if self._ws is None:
            self._ws = {}
        if key not in self._ws:
            if len(self._ws) >= 2:                             # a tracker alternates between two shapes at most (batch n / 2n)
                self._ws.pop(next(iter(self._ws)))
            c = self.hidden_dim
            e = lambda ch: torch.empty(n, ch, h8, w8, device=device)
            self._ws[key] = dict(hx=e(2 * c), rhx=e(2 * c), z=e(c), cat=e(2 * c),
                                 corr=e(self.corr_levels * (2 * self.corr_radius + 1) ** 2),
                                 ctx=dict(zr1=e(2 * c), q1=e(c), zr2=e(2 * c), q2=e(c)),
                                 coords0=coords_grid(n, h8, w8, device), coords1=e(2), flow=e(2),
                                 zero2=torch.zeros(n, 2, h8, w8, device=device))
        return self._ws[key]

    def _loop_program(self, pyr, ws, iters, side):
        """The update loop on workspace ``ws`` as a launch list (ops.OpList): per iteration [fork: convf1 -> convf2 on the side stream]
        lookup, convc1, convc2, [join | convf1, convf2], conv, the four GRU convolutions, FlowHead.conv1 and the flow update -- built once
        per (weights, pyramid, workspace, iters, streams) and replayed by one rpe_run_ops call per pass.  Cells 0 / 1: the fork / join
        events; cells 2 + 2k, 3 + 2k: timing events around iteration k's lookup (LOOKUP_EVENT_SINK; empty = skipped).
        Returns (list, marks) with marks[k] = first op of iteration k, marks[iters] = the end."""
        ub = self.update_block
        hx, rhx, z_buf, cat_buf, corr, coords1, flow = (ws[k] for k in ('hx', 'rhx', 'z', 'cat', 'corr', 'coords1', 'flow'))
        P = ub.packed_convs(hx.shape[-1])
        key = (id(P), pyr.buf.data_ptr(), iters, None if side is None else side[0].cuda_stream, LOOKUP_FUSED, LOOKUP_FUSED_MAX_WGS)
        cached = ws.get('_program')
        if cached is not None and cached[0] == key:
            return cached[1], cached[2]
        _, _, _, c1, c2, _, cv_ = ub.encoder._calls(corr, cat_buf, hx, rhx, P)
        f1, f2 = ub.encoder.flow_branch_launchers(flow, corr, cat_buf, hx, rhx, P)
        seq = ub.gru_launchers(hx, rhx, z_buf, ws['ctx'], flow, coords1, True, P)
        lk = pyr.lookup(coords1, out=corr, prepare=True)
        n_wgs = hx.shape[0] * -(-(hx.shape[2] * -(-hx.shape[3] // 8)) // 8)
        if LOOKUP_FUSED and not CONV_BF16X3 and n_wgs <= LOOKUP_FUSED_MAX_WGS and ops.PackedLookupConv.supported(pyr.levels, pyr.radius, pyr.w8):
            cor = P['cor_buf'](corr)
            if 'convc1_lookup' not in P:
                P['convc1_lookup'] = ops.PackedLookupConv(ub.encoder.convc1.weight, ub.encoder.convc1.bias)
            lk, c1 = pyr.lookup_conv1x1(coords1, P['convc1_lookup'], cor, relu=True, prepare=True), None      # one launch for lookup -> convc1 -> ReLU
        prog = ops.OpList(n_cells=2 + 2 * iters)
        if side is not None:
            for cell, ev in ((0, side[1]), (1, side[2])):
                ev.record()                                       # (torch creates the hipEvent_t at the first record)
                prog.cells[cell] = ev.cuda_event
        marks = []
        for itr in range(iters):
            marks.append(prog.mark())
            if side is not None:                                  # the flow branch needs only the flow: beside lookup -> convc1 -> convc2
                prog.record(0, 0).wait(0, 1).add(f1, 1).add(f2, 1).record(1, 1)
            prog.record(2 + 2 * itr, 0).add(lk).record(3 + 2 * itr, 0)
            if c1 is not None:
                prog.add(c1)
            prog.add(c2)
            if side is not None:
                prog.wait(1, 0)
            else:
                prog.add(f1).add(f2)
            prog.add(cv_)
            for launch in seq:
                prog.add(launch)
        marks.append(prog.mark())
        prog.keep = (P, pyr, side)
        prog.armed = False
        ws['_program'] = (key, prog, marks)
        return prog, marks

    @torch.no_grad()
    def encode_features(self, images):
        """fnet on raw 0..255 images (normalised like forward does); one batch or a list of batches encoded as one."""
        return self.fnet(images, raw255=True).float()

    @torch.no_grad()
    def encode_context(self, images):
        """cnet on raw 0..255 images (one batch or a list of batches): (N,256,H/8,W/8) = (tanh(net) | relu(inp)), the initial
        hidden state and the context features of core/RAFT/core/raft.py, activated in the output layer's epilogue."""
        return self.cnet(images, raw255=True, split_act=True)

    @torch.no_grad()
    def encode_both(self, feature_images, context_images):
        """(encode_features(feature_images), encode_context(context_images)).  The two encoders share nothing but their input images,
        so on a GPU the context encoder runs on the side stream beside the feature encoder (ENC_STREAMS): two independent chains of
        launches fill each other's tails and latencies -- at bench geometry 23.3 -> 23.0 ms for the pair (step 67.65 -> 67.24 ms); not for
        the one to three images of sequential tracking, where the fork / join costs what it hides (138 vs 136 frames/s with it).
        Same kernels on the same inputs: identical results."""
        many = isinstance(context_images, (list, tuple))
        first = context_images[0] if many else context_images
        count = sum(t.shape[0] for t in context_images) if many else first.shape[0]
        if not (ENC_STREAMS and first.is_cuda and count >= ENC_STREAMS_MIN):
            return self.encode_features(feature_images), self.encode_context(context_images)
        stream, ev_in, ev_done = self._side_stream(first.device)
        cur = torch.cuda.current_stream(first.device)
        ev_in.record(cur)
        with torch.cuda.stream(stream):
            stream.wait_event(ev_in)                          # the images (and the weights) are ready on the caller's stream
            cn = self.encode_context(context_images)
            ev_done.record(stream)
        f = self.encode_features(feature_images)
        cur.wait_event(ev_done)
        cn.record_stream(cur)                                 # allocated on the side stream, consumed (and freed) on the caller's
        return f, cn

    def _begin(self, pyr, ws, fmap1, fmap2, cnet, fused):
        """Everything of a pass in front of the update loop: the correlation pyramid, h <- tanh half of the context encoder's output, the
        GRU's four loop-invariant context terms, and (fused route) coords1 <- grid, flow <- 0 in the workspace's persistent buffers,
        which the flow head's output layer then updates in place."""
        c = self.hidden_dim
        pyr.build(fmap1.float(), fmap2.float(), fp16_features=self.mixed_precision, bf16x3=(CORR_BF16X3 or CONV_BF16X3) and not self.mixed_precision)
        hx, rhx = ws['hx'], ws['rhx']
        ops.copy_planes(cnet[:, :c], hx[:, :c])
        ctx = self.update_block.context_terms(cnet[:, c:], out=ws['ctx'] if fused else None)      # (fused: written into the persistent buffers)
        if fused:
            ops.copy_planes(ws['coords0'], ws['coords1'])
            ops.copy_planes(ws['zero2'], ws['flow'])
            ops.copy_planes(ws['zero2'], hx[:, 2 * c - 2:])
            ops.copy_planes(ws['zero2'], rhx[:, 2 * c - 2:])
        return ctx

    def _finish(self, ws, upsample):
        """Behind the loop of the fused route: the returned prediction (mask head + convex x8 up-sampling, or a copy of the 1/8 flow) and
        the returned hidden state, both fresh tensors."""
        c = self.hidden_dim
        hx, flow = ws['hx'], ws['flow']
        pred = ops.upsample_convex(flow, self.update_block.up_mask(hx[:, :c])) if upsample else ops.copy_planes(flow, torch.empty_like(flow))
        h_buf = ops.copy_planes(hx[:, :c], torch.empty(hx.shape[0], c, hx.shape[2], hx.shape[3], device=hx.device))
        return pred, h_buf

    def _run_loop(self, pyr, ws, iters, side):
        prog, marks = self._loop_program(pyr, ws, iters, side)
        if LOOKUP_EVENT_SINK is not None:
            for i, h in enumerate(LOOKUP_EVENT_SINK(iters)):
                prog.cells[2 + i] = h
            prog.armed = True
        elif prog.armed:
            for i in range(2 * iters):
                prog.cells[2 + i] = None
            prog.armed = False
        streams = (ops.raw_stream(),) if side is None else (ops.raw_stream(), side[0].cuda_stream)
        return prog, marks, streams

    def _forward_recorded(self, fmap1, fmap2, cnet, iters, upsample):
        """forward() for a small pass on given encoder outputs (sequential tracking: one or two flow pairs per frame) as THREE calls into
        the library: the recorded front (_begin), the loop's launch list, the recorded tail (_finish).  None = this pass cannot be
        recorded (the caller goes on call by call)."""
        N, _, h8, w8 = fmap1.shape
        dev = fmap1.device
        c = self.hidden_dim
        ts = (fmap1, fmap2, cnet)
        if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device == dev for t in ts) or _overlap(ts) \
                or tuple(cnet.shape) != (N, 2 * c, h8, w8) or dev.index != torch.cuda.current_device():
            return None                                               # (the call-by-call route raises what needs raising)
        if getattr(self, '_recorded', None) is None:
            self._recorded = _Recorded()
        if '_key_tensors' not in self.__dict__:
            self.__dict__['_key_tensors'] = _key_sources(self.update_block)
        pyr = self._pyramid(N, h8, w8, dev)
        ws = self._workspace(N, h8, w8, dev)
        key = (N, h8, w8, dev.index, ops.raw_stream(), iters, upsample, self.mixed_precision, WINOGRAD, CORR_BF16X3, CONV_BF16X3, X3_GRU, SIDE_STREAM,
               pyr.buf.data_ptr(), ws['hx'].data_ptr(), _tensor_key(self.__dict__['_key_tensors']))
        side = self._side_stream(dev) if SIDE_STREAM and N * h8 * w8 <= SIDE_STREAM_MAX else None
        entry = self._recorded.get(key)
        if entry is not None:
            pre, post = entry.pre, entry
            pre.replay({'f1': fmap1, 'f2': fmap2, 'cnet': cnet})
            prog, _, streams = self._run_loop(pyr, ws, iters, side)
            prog.run(streams)
            pred, h_buf = torch.empty(post.pred_shape, device=dev), torch.empty(N, c, h8, w8, device=dev)
            post.replay({'pred': pred, 'h': h_buf})
            return [pred], h_buf, cnet[:, c:]
        if not self._recorded.wanted(key):
            return None
        prog, _, streams = self._run_loop(pyr, ws, iters, side)      # (built outside the recordings: its launchers must hold the real entry points)
        pre = ops.Recorder()
        with pre:
            self._begin(pyr, ws, fmap1, fmap2, cnet, True)
        prog.run(streams)
        post = ops.Recorder()
        with post:
            pred, h_buf = self._finish(ws, upsample)
        if pre.complete and post.complete:
            ok = pre.bind('f1', fmap1) > 0 and pre.bind('f2', fmap2) > 0 and pre.bind('cnet', cnet) > 0 and post.bind('pred', pred) > 0 and post.bind('h', h_buf) > 0
            pre.complete = post.complete = ok
        post.complete = post.complete and pre.complete
        post.pre, post.pred_shape = pre, tuple(pred.shape)
        self._recorded.put(key, post)
        return [pred], h_buf, cnet[:, c:]

    @torch.no_grad()
    def forward(self, image1, image2, upsample=True, iters=None, all_flows=False, fmaps=None, cnet=None):
        """image1, image2: (N,3,H,W) in 0..255.  Inference only (the reference freezes RAFT, train.yaml:51).
        ``fmaps`` / ``cnet`` accept encoder outputs computed elsewhere (both encoders normalise per sample -- instance
        norm / frozen batch norm -- so a caller may encode every distinct image once and reuse it); with both given the
        images may be None.  ``cnet`` is encode_context's output: (tanh(net) | relu(inp))."""
        iters = self.iters if iters is None else iters
        if image1 is None:                                    # encoder outputs given: the images are not needed again
            (N, _, h8, w8), dev = fmaps[0].shape, fmaps[0].device
        else:
            N, _, H, W = image1.shape
            h8, w8 = H // 8, W // 8
            dev = image1.device
        if fmaps is None:
            f = self.encode_features((image1, image2))
            fmap1, fmap2 = f[:N], f[N:]
        else:
            fmap1, fmap2 = fmaps                              # precomputed by encode_features (caller de-duplicates)
        if cnet is None:
            cnet = self.encode_context(image1)                # (tanh(net) | relu(inp))
        c = self.hidden_dim
        fused = self.update_block.packed_convs(w8) is not None
        if fused and FRAME_OPLISTS and LOOP_OPLIST and not all_flows and N <= FRAME_OPLISTS_MAX_IMAGES and LOOKUP_EVENT_SINK is None:
            r = self._forward_recorded(fmap1, fmap2, cnet, iters, upsample)
            if r is not None:
                return r
        pyr = self._pyramid(N, h8, w8, dev)
        ws = self._workspace(N, h8, w8, dev)
        ctx = self._begin(pyr, ws, fmap1, fmap2, cnet, fused)
        hx, rhx, z_buf, cat_buf, corr = ws['hx'], ws['rhx'], ws['z'], ws['cat'], ws['corr']   # hx = (h | motion | flow)
        h_buf = torch.empty(N, c, h8, w8, device=dev)         # returned to the caller: fresh
        inp = cnet[:, c:]
        coords0 = ws['coords0']
        flow_predictions = []
        if fused:
            coords1, flow = ws['coords1'], ws['flow']
        else:
            coords1 = coords0.clone()
        side = self._side_stream(dev) if fused and SIDE_STREAM and N * h8 * w8 <= SIDE_STREAM_MAX else None
        if fused and LOOP_OPLIST:
            prog, marks, streams = self._run_loop(pyr, ws, iters, side)
            if not all_flows:
                prog.run(streams)
            for itr in range(iters if all_flows else 0):
                prog.run(streams, marks[itr], marks[itr + 1])
                if itr < iters - 1:
                    flow_predictions.append(ops.upsample_convex(flow, self.update_block.up_mask(hx[:, :c])) if upsample
                                            else ops.copy_planes(flow, torch.empty_like(flow)))
            flow_predictions.append(ops.upsample_convex(flow, self.update_block.up_mask(hx[:, :c])) if upsample
                                    else ops.copy_planes(flow, torch.empty_like(flow)))
            iters = 0                                             # (the launch-by-launch loop below is the other route)
        for itr in range(iters):
            done = None
            if side is not None:
                # the motion encoder's flow branch (convf1 -> convf2) needs only the flow: it runs on a side stream beside
                # lookup -> convc1 -> convc2 and fills the partly idle last rounds of those launches
                stream, ev_flow, done = side
                ev_flow.record()
                with torch.cuda.stream(stream):
                    stream.wait_event(ev_flow)
                    self.update_block.encoder.flow_branch(flow, corr, cat_buf, hx, rhx, self.update_block.packed_convs(w8))
                    done.record()
            pyr.lookup(coords1, out=corr)
            if fused:
                self.update_block.step(hx, rhx, z_buf, cat_buf, h_buf, ctx, corr, flow, coords1, in_place=True, flow_branch_done=done)
            else:
                flow = coords1 - coords0
                coords1 = self.update_block.step(hx, rhx, z_buf, cat_buf, h_buf, ctx, corr, flow, coords1)
            if all_flows or itr == iters - 1:
                lowres = flow if fused else coords1 - coords0
                if upsample:
                    flow_predictions.append(ops.upsample_convex(lowres, self.update_block.up_mask(hx[:, :c])))
                else:
                    flow_predictions.append(ops.copy_planes(lowres, torch.empty_like(lowres)) if fused else lowres)
        ops.copy_planes(hx[:, :c], h_buf)
        return flow_predictions, h_buf, inp
