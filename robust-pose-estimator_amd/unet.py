"""TinyUNet weight head -- stays on PyTorch-ROCm (north star).

Host-side mirror of the reference's core/unet/unet.py:7-82 with identical parameter names, so reference
checkpoints load unchanged: valid 3x3 convs, BatchNorm, 2x2 transposed-conv up-sampling, centre-cropped
skips, 1x1 head, bilinear resize to (H, W).  DownBlock = conv-norm-relu-conv (:15-16); UpBlock =
conv-relu-norm-conv (:18-20)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class DownBlock(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv1 = nn.Conv2d(in_ch, out_ch, 3)
        self.norm = nn.BatchNorm2d(out_ch)
        self.relu = nn.ReLU()
        self.conv2 = nn.Conv2d(out_ch, out_ch, 3)

    def forward(self, x):
        return self.conv2(self.relu(self.norm(self.conv1(x))))


class UpBlock(DownBlock):
    def forward(self, x):
        return self.conv2(self.norm(self.relu(self.conv1(x))))


class Encoder(nn.Module):
    def __init__(self, chs):
        super().__init__()
        self.enc_blocks = nn.ModuleList([DownBlock(chs[i], chs[i + 1]) for i in range(len(chs) - 1)])
        self.pool = nn.MaxPool2d(2)

    def forward(self, x):
        ftrs = []
        for block in self.enc_blocks:
            x = block(x)
            ftrs.append(x)
            x = self.pool(x)
        return ftrs


class Decoder(nn.Module):
    def __init__(self, chs):
        super().__init__()
        self.chs = chs
        self.upconvs = nn.ModuleList([nn.ConvTranspose2d(chs[i], chs[i + 1], 2, 2) for i in range(len(chs) - 1)])
        self.dec_blocks = nn.ModuleList([UpBlock(chs[i], chs[i + 1]) for i in range(len(chs) - 1)])

    def forward(self, x, encoder_features):
        for i in range(len(self.chs) - 1):
            x = self.upconvs[i](x)
            e = encoder_features[i]
            H, W = x.shape[-2:]
            H2, W2 = e.shape[-2:]
            dh, dw = (H2 - H) // 2, (W2 - W) // 2
            x = self.dec_blocks[i](torch.cat([x, e[..., dh:(H2 - dh), dw:(W2 - dw)]], dim=1))
        return x


class TinyUNet(nn.Module):
    def __init__(self, in_channels, output_size):
        super().__init__()
        self.encoder = Encoder((in_channels, 16, 32, 64))
        self.decoder = Decoder((64, 32, 16))
        self.head = nn.Conv2d(16, 1, 1)
        self.out_sz = output_size

    def forward(self, x):
        enc = self.encoder(x)
        out = self.decoder(enc[::-1][0], enc[::-1][1:])
        return F.interpolate(self.head(out), self.out_sz, mode='bilinear')
