"""TinyUNet weight head: parameter tree + the blob ``rpe_unet_heads`` (csrc/unet.hip, the inference route PoseNet uses)
reads; ``forward`` itself is the module-level route -- convolutions on PyTorch-ROCm, every epilogue a fused HIP pass --
kept for training, for A/B runs (RPE_FUSED_HEADS=0) and for calling one head on its own.

Host mirror of the reference's ``TinyUNet(in_channels, output_size)`` (core/unet/unet.py:80-82; architecture :7-77)
with its parameter names (``encoder.enc_blocks.i.conv1/norm/conv2``, ``decoder.upconvs.i``, ``decoder.dec_blocks.i``,
``head``) so checkpoints load unchanged.  Inference: the batch norms run frozen and are folded, together with
the preceding conv bias, into one per-channel affine map applied by ``rpe_affine_act``:
    encoder stage : conv1 (no bias) -> [affine + ReLU]                 -> conv2            (conv-norm-relu-conv, :15-16)
    decoder stage : conv1 (no bias) -> [bias + ReLU] -> [affine]       -> conv2            (conv-relu-norm-conv, :18-20)
Training (gradients enabled, or norms in train mode): the same architecture on plain differentiable PyTorch-ROCm ops, batch norm in the module's
train/eval state -- the heads are what the reference trains (scripts/train_posenet.py:97-136)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

WIDTHS = (16, 32, 64)


class _Stage(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3)
        self.norm = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3)

    def folded_norm(self, with_conv_bias):
        """Frozen BN as y = x*scale + shift; optionally absorbing conv1's bias (when the norm directly follows conv1)."""
        n = self.norm
        if n.training:
            raise RuntimeError('TinyUNet runs with frozen batch norm (inference only)')
        scale = (n.weight / torch.sqrt(n.running_var + n.eps)).detach()
        shift = n.bias.detach() - n.running_mean * scale
        if with_conv_bias:
            shift = shift + self.conv1.bias.detach() * scale
        return scale.contiguous(), shift.contiguous()


class _Tree(nn.Module):
    def __init__(self, **mods):
        super().__init__()
        for k, v in mods.items():
            setattr(self, k, v)


def _blob_sources(net):
    """Every tensor pack_params reads, fetched from the modules' current attributes."""
    out = []
    for st in list(net.encoder.enc_blocks) + list(net.decoder.dec_blocks):
        n = st.norm
        out += [st.conv1.weight, st.conv1.bias, st.conv2.weight, st.conv2.bias, n.weight, n.bias, n.running_mean, n.running_var]
    for upc in net.decoder.upconvs:
        out += [upc.weight, upc.bias]
    out += [net.head.weight, net.head.bias]
    return [t for t in out if t is not None]


def pack_params(net):
    """The parameter blob rpe_unet_heads reads (layout: csrc/unet.hip): 3x3 weights as [cin][9][cout], transposed-conv weights as
    [cin][4][cout], inference-mode batch norm as per-channel (scale, shift) -- after conv1 + bias in the encoder stages
    (conv-norm-relu-conv), after the ReLU in the decoder stages (conv-relu-norm-conv).  Cached until a parameter changes."""
    # the module's CURRENT Parameter and buffer objects every call (load_state_dict(assign=True) or re-assigning conv.weight replaces the
    # objects: a list cached once would keep serving the old blob); ~40 attribute reads, cheaper than walking parameters() / buffers()
    tensors = _blob_sources(net)
    key = tuple(id(t) for t in tensors) + tuple(t._version for t in tensors) + tuple(t.data_ptr() for t in tensors)
    cached = getattr(net, '_rpe_blob', None)
    if cached is not None and cached[0] == key:
        return cached[1]
    parts = []
    w3 = lambda w: w.detach().permute(1, 2, 3, 0).reshape(-1)            # (cout,cin,3,3) -> [cin][9][cout]
    for st in net.encoder.enc_blocks:
        scale, shift = st.folded_norm(with_conv_bias=True)
        parts += [w3(st.conv1.weight), scale, shift, w3(st.conv2.weight), st.conv2.bias.detach()]
    for upc, st in zip(net.decoder.upconvs, net.decoder.dec_blocks):
        scale, shift = st.folded_norm(with_conv_bias=False)
        parts += [upc.weight.detach().permute(0, 2, 3, 1).reshape(-1), upc.bias.detach(),            # (cin,cout,2,2) -> [cin][4][cout]
                  w3(st.conv1.weight), st.conv1.bias.detach(), scale, shift, w3(st.conv2.weight), st.conv2.bias.detach()]
    parts += [net.head.weight.detach().reshape(-1), net.head.bias.detach().reshape(-1)]
    blob = torch.cat([p.reshape(-1).float() for p in parts]).contiguous()
    net._rpe_blob = (key, blob)
    return blob


class TinyUNet(nn.Module):
    def __init__(self, in_channels, output_size):
        super().__init__()
        down = (in_channels,) + WIDTHS
        up = WIDTHS[::-1]
        self.encoder = _Tree(enc_blocks=nn.ModuleList(_Stage(down[i], down[i + 1]) for i in range(3)))
        self.decoder = _Tree(upconvs=nn.ModuleList(nn.ConvTranspose2d(up[i], up[i + 1], 2, 2) for i in range(2)),
                             dec_blocks=nn.ModuleList(_Stage(up[i], up[i + 1]) for i in range(2)))
        self.head = nn.Conv2d(WIDTHS[0], 1, 1)
        self.out_sz = output_size

    def _check_size(self, x):
        if x.shape[-2] < 44 or x.shape[-1] < 44:
            raise ValueError('TinyUNet needs a 1/8 grid of at least 44x44 (valid convolutions), as in the reference')

    def forward(self, x):
        """Batch-statistics norms (module in train mode, with or without gradients -- a validation loop that forgot
        ``.eval()`` runs batch statistics in the reference too) and anything that needs gradients take the differentiable
        route; frozen norms without gradients take the folded inference route."""
        self._check_size(x)
        batch_stats = any(st.norm.training for st in list(self.encoder.enc_blocks) + list(self.decoder.dec_blocks))
        if batch_stats or (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return self.forward_train(x)
        with torch.no_grad():
            return self.forward_infer(x)

    def forward_train(self, x):
        """core/unet/unet.py:7-77 on differentiable torch ops (encoder: conv-norm-relu-conv + max-pool; decoder: up-conv,
        centre-cropped skip, conv-relu-norm-conv; 1x1 head; bilinear resize to the output size)."""
        def bn(st, t):
            n = st.norm
            return F.batch_norm(t, n.running_mean, n.running_var, n.weight, n.bias, n.training, n.momentum, n.eps)
        skips = []
        for st in self.encoder.enc_blocks:
            x = F.conv2d(torch.relu(bn(st, F.conv2d(x, st.conv1.weight, st.conv1.bias))), st.conv2.weight, st.conv2.bias)
            skips.append(x)
            x = F.max_pool2d(x, 2)
        x = skips.pop()
        for upc, st in zip(self.decoder.upconvs, self.decoder.dec_blocks):
            x = F.conv_transpose2d(x, upc.weight, upc.bias, stride=2)
            sk = skips.pop()
            dh, dw = (sk.shape[-2] - x.shape[-2]) // 2, (sk.shape[-1] - x.shape[-1]) // 2
            x = torch.cat((x, sk[..., dh:sk.shape[-2] - dh, dw:sk.shape[-1] - dw]), dim=1)
            x = F.conv2d(bn(st, torch.relu(F.conv2d(x, st.conv1.weight, st.conv1.bias))), st.conv2.weight, st.conv2.bias)
        return F.interpolate(F.conv2d(x, self.head.weight, self.head.bias), self.out_sz, mode='bilinear')

    def forward_infer(self, x):
        skips = []
        for st in self.encoder.enc_blocks:
            scale, shift = st.folded_norm(with_conv_bias=True)
            y = ops.affine_act(F.conv2d(x.contiguous(), st.conv1.weight), scale, shift, relu=True)
            x = F.conv2d(y, st.conv2.weight, st.conv2.bias)
            skips.append(x)
            x = F.max_pool2d(x, 2)
        x = skips.pop()
        for upc, st in zip(self.decoder.upconvs, self.decoder.dec_blocks):
            x = F.conv_transpose2d(x, upc.weight, upc.bias, stride=2)
            sk = skips.pop()
            dh, dw = (sk.shape[-2] - x.shape[-2]) // 2, (sk.shape[-1] - x.shape[-1]) // 2
            x = torch.cat((x, sk[..., dh:sk.shape[-2] - dh, dw:sk.shape[-1] - dw]), dim=1)
            y = ops.bias_act(F.conv2d(x, st.conv1.weight), st.conv1.bias, relu=True)
            scale, shift = st.folded_norm(with_conv_bias=False)
            y = ops.affine_act(y, scale, shift, relu=False)
            x = F.conv2d(y, st.conv2.weight, st.conv2.bias)
        return F.interpolate(F.conv2d(x, self.head.weight, self.head.bias), self.out_sz, mode='bilinear')
