"""Oracle (test infrastructure): the TinyUNet weight head as one functional forward over a parameter tree.

Restates core/unet/unet.py:7-82 of the reference: three encoder stages of (3x3 valid conv -> BatchNorm -> ReLU ->
3x3 valid conv) with 2x2 max-pooling in between, two decoder stages of (2x2 stride-2 transposed conv, centre-cropped
skip concatenation, 3x3 valid conv -> ReLU -> BatchNorm -> 3x3 valid conv -- note the norm/ReLU order differs from the
encoder, :15-16 vs :18-20), a 1x1 head and a bilinear resize to the image size.  Only the parameter names are shared
with the reference (``encoder.enc_blocks.i.{conv1,norm,conv2}``, ``decoder.upconvs.i``, ``decoder.dec_blocks.i...``,
``head``) so that its checkpoints load; the computation is written out with torch.nn.functional calls."""
import torch
import torch.nn as nn
import torch.nn.functional as F

WIDTHS = (16, 32, 64)


def _stage(cin, cout):
    return nn.ModuleDict(dict(conv1=nn.Conv2d(cin, cout, 3), norm=nn.BatchNorm2d(cout), conv2=nn.Conv2d(cout, cout, 3)))


class _Params(nn.Module):
    """A bare namespace of sub-modules (gives the state-dict its prefix)."""

    def __init__(self, **mods):
        super().__init__()
        for k, v in mods.items():
            setattr(self, k, v)


def _bn(m, x):
    return F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, m.training, m.momentum, m.eps)


def _conv(m, x):
    return F.conv2d(x, m.weight, m.bias)


def _centre_crop(t, h, w):
    dh, dw = (t.shape[-2] - h) // 2, (t.shape[-1] - w) // 2
    return t[..., dh:t.shape[-2] - dh, dw:t.shape[-1] - dw]


class TinyUNet(nn.Module):
    def __init__(self, in_channels, output_size):
        super().__init__()
        down = (in_channels,) + WIDTHS
        up = WIDTHS[::-1]
        self.encoder = _Params(enc_blocks=nn.ModuleList(_stage(down[i], down[i + 1]) for i in range(3)))
        self.decoder = _Params(upconvs=nn.ModuleList(nn.ConvTranspose2d(up[i], up[i + 1], 2, 2) for i in range(2)),
                               dec_blocks=nn.ModuleList(_stage(up[i], up[i + 1]) for i in range(2)))
        self.head = nn.Conv2d(WIDTHS[0], 1, 1)
        self.out_sz = output_size

    def forward(self, x):
        skips = []
        for st in self.encoder.enc_blocks:
            x = _conv(st['conv2'], F.relu(_bn(st['norm'], _conv(st['conv1'], x))))
            skips.append(x)
            x = F.max_pool2d(x, 2)          # the reference pools after the last stage too (result unused) -- so the
        x = skips.pop()                     # smallest admissible 1/8 grid is 44x44, as there
        for up, st in zip(self.decoder.upconvs, self.decoder.dec_blocks):
            x = F.conv_transpose2d(x, up.weight, up.bias, stride=2)
            x = torch.cat((x, _centre_crop(skips.pop(), *x.shape[-2:])), dim=1)
            x = _conv(st['conv2'], _bn(st['norm'], F.relu(_conv(st['conv1'], x))))
        return F.interpolate(_conv(self.head, x), self.out_sz, mode='bilinear')
