"""Oracle (test infrastructure): RAFT optical flow, plain PyTorch-CPU float32.

PARITY UNPINNED.  The reference vendors RAFT as the git submodule ``core/RAFT`` (fork aimi-lab/RAFT,
.gitmodules:1-3) which is EMPTY in /root/reference and has no recoverable commit; no reference test
touches it.  This file restates the published princeton-vl/RAFT architecture (Teed & Deng, ECCV 2020:
core/raft.py, core/extractor.py, core/corr.py, core/update.py, core/utils/utils.py) subject to the
facts the reference's call sites fix:

  * constructed from a dict config (``small, dropout, iters, image_shape`` ...), core/pose/pose_net.py:21
  * ``freeze_bn()`` exists, pose_net.py:22
  * ``forward(img1, img2, upsample=True)`` returns the 3-tuple (flow_predictions list, gru hidden
    state, context), pose_net.py:47,65; hidden and context are 128 channels each at 1/8 resolution
    (weight heads take 128+128+8 channels, pose_net.py:24-27)
  * ``[0][-1]`` is the final full-resolution flow (N,2,H,W); with ``upsample=False`` the flow stays
    at 1/8 resolution in 1/8-pixel units (pose_net.py:129-132 divides the depth by 8)
  * the iteration count comes from ``config['iters']`` (configuration/train.yaml:3 -> 12)
  * state-dict keys are upstream's (``raft-things.pth`` loads after stripping ``module.``, pose_net.py:137-147)

The product package keeps an identically-keyed module tree, so one seeded state dict drives both.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- encoders (extractor.py)
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

    def forward(self, x):
        y = self.relu(self.norm1(self.conv1(x)))
        y = self.relu(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)


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
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def _make_layer(self, dim, stride=1):
        l1 = ResidualBlock(self.in_planes, dim, self.norm_fn, stride=stride)
        l2 = ResidualBlock(dim, dim, self.norm_fn, stride=1)
        self.in_planes = dim
        return nn.Sequential(l1, l2)

    def forward(self, x):
        is_list = isinstance(x, (tuple, list))
        if is_list:
            batch_dim = x[0].shape[0]
            x = torch.cat(x, dim=0)
        x = self.relu1(self.norm1(self.conv1(x)))
        x = self.layer3(self.layer2(self.layer1(x)))
        x = self.conv2(x)
        if is_list:
            x = torch.split(x, [batch_dim, batch_dim], dim=0)
        return x


# ----------------------------------------------------------------------------- correlation (corr.py)
def coords_grid(batch, ht, wd):
    ys, xs = torch.meshgrid(torch.arange(ht), torch.arange(wd), indexing='ij')
    coords = torch.stack((xs, ys), dim=0).float()          # channel 0 = x, channel 1 = y
    return coords[None].repeat(batch, 1, 1, 1)


def bilinear_sampler(img, coords):
    """utils.py: grid_sample(align_corners=True) on pixel coordinates."""
    H, W = img.shape[-2:]
    xgrid, ygrid = coords.split([1, 1], dim=-1)
    xgrid = 2 * xgrid / (W - 1) - 1
    ygrid = 2 * ygrid / (H - 1) - 1
    return F.grid_sample(img, torch.cat([xgrid, ygrid], dim=-1), align_corners=True)


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels, self.radius = num_levels, radius
        corr = CorrBlock.corr(fmap1, fmap2)
        b, h1, w1, dim, h2, w2 = corr.shape
        corr = corr.reshape(b * h1 * w1, dim, h2, w2)
        self.corr_pyramid = [corr]
        for _ in range(num_levels - 1):
            corr = F.avg_pool2d(corr, 2, stride=2)
            self.corr_pyramid.append(corr)

    def __call__(self, coords):
        r = self.radius
        coords = coords.permute(0, 2, 3, 1)
        b, h1, w1, _ = coords.shape
        out = []
        for i in range(self.num_levels):
            corr = self.corr_pyramid[i]
            dx = torch.linspace(-r, r, 2 * r + 1)
            dy = torch.linspace(-r, r, 2 * r + 1)
            # upstream quirk kept: meshgrid(dy, dx) stacked as (.., 2) is ADDED to (x, y), so window
            # index (i, j) samples at (x + dy[i], y + dx[j]) -- the 9x9 window is stored transposed.
            delta = torch.stack(torch.meshgrid(dy, dx, indexing='ij'), dim=-1)
            centroid = coords.reshape(b * h1 * w1, 1, 1, 2) / 2 ** i
            coords_lvl = centroid + delta.view(1, 2 * r + 1, 2 * r + 1, 2)
            c = bilinear_sampler(corr, coords_lvl)
            out.append(c.view(b, h1, w1, -1))
        out = torch.cat(out, dim=-1)
        return out.permute(0, 3, 1, 2).contiguous().float()

    @staticmethod
    def corr(fmap1, fmap2):
        b, dim, ht, wd = fmap1.shape
        f1 = fmap1.view(b, dim, ht * wd)
        f2 = fmap2.view(b, dim, ht * wd)
        corr = torch.matmul(f1.transpose(1, 2), f2).view(b, ht, wd, 1, ht, wd)
        return corr / torch.sqrt(torch.tensor(dim).float())


# ----------------------------------------------------------------------------- update block (update.py)
class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.conv2(self.relu(self.conv1(x)))


class SepConvGRU(nn.Module):
    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        self.convz1 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (1, 5), padding=(0, 2))
        self.convr1 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (1, 5), padding=(0, 2))
        self.convq1 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (1, 5), padding=(0, 2))
        self.convz2 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (5, 1), padding=(2, 0))
        self.convr2 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (5, 1), padding=(2, 0))
        self.convq2 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (5, 1), padding=(2, 0))

    def forward(self, h, x):
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(self.convz1(hx))
        r = torch.sigmoid(self.convr1(hx))
        q = torch.tanh(self.convq1(torch.cat([r * h, x], dim=1)))
        h = (1 - z) * h + z * q
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(self.convz2(hx))
        r = torch.sigmoid(self.convr2(hx))
        q = torch.tanh(self.convq2(torch.cat([r * h, x], dim=1)))
        h = (1 - z) * h + z * q
        return h


class BasicMotionEncoder(nn.Module):
    def __init__(self, corr_levels=4, corr_radius=4):
        super().__init__()
        cor_planes = corr_levels * (2 * corr_radius + 1) ** 2
        self.convc1 = nn.Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)

    def forward(self, flow, corr):
        cor = F.relu(self.convc1(corr))
        cor = F.relu(self.convc2(cor))
        flo = F.relu(self.convf1(flow))
        flo = F.relu(self.convf2(flo))
        out = F.relu(self.conv(torch.cat([cor, flo], dim=1)))
        return torch.cat([out, flow], dim=1)


class BasicUpdateBlock(nn.Module):
    def __init__(self, corr_levels=4, corr_radius=4, hidden_dim=128):
        super().__init__()
        self.encoder = BasicMotionEncoder(corr_levels, corr_radius)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(256, 64 * 9, 1, padding=0))

    def forward(self, net, inp, corr, flow):
        motion_features = self.encoder(flow, corr)
        inp = torch.cat([inp, motion_features], dim=1)
        net = self.gru(net, inp)
        delta_flow = self.flow_head(net)
        mask = .25 * self.mask(net)                      # scale mask to balance gradients (upstream)
        return net, mask, delta_flow


# ----------------------------------------------------------------------------- RAFT (raft.py)
def upsample_flow(flow, mask):
    """Convex 8x up-sampling: [N,2,H/8,W/8] -> [N,2,H,W] (raft.py: upsample_flow)."""
    N, _, H, W = flow.shape
    mask = mask.view(N, 1, 9, 8, 8, H, W)
    mask = torch.softmax(mask, dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(N, 2, 9, 1, 1, H, W)
    up = torch.sum(mask * up, dim=2)
    up = up.permute(0, 1, 4, 2, 5, 3)
    return up.reshape(N, 2, 8 * H, 8 * W)


class RAFT(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.get('small', False):
            raise NotImplementedError("RAFT-small is not on the reference's inference path (train.yaml:5 small: False)")
        self.config = config
        # "fp16 features" (BASELINE config 5): the feature maps reach the correlation rounded to fp16 -- what upstream's
        # mixed_precision autocast hands to corr.py before its .float().  The encoders themselves stay f32 on both sides, so
        # the comparison with the HIP path (16-bit MFMA, f32 accumulation) differs by f32 summation order only.
        self.mixed_precision = bool(config.get('mixed_precision', False))
        self.iters = int(config.get('iters', 12))
        self.hidden_dim = self.context_dim = 128
        self.corr_levels, self.corr_radius = 4, 4
        drop = config.get('dropout', 0.0)
        self.fnet = BasicEncoder(output_dim=256, norm_fn='instance', dropout=drop)
        self.cnet = BasicEncoder(output_dim=256, norm_fn='batch', dropout=drop)
        self.update_block = BasicUpdateBlock(self.corr_levels, self.corr_radius, hidden_dim=128)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def forward(self, image1, image2, upsample=True, iters=None, return_corr=False):
        iters = self.iters if iters is None else iters
        image1 = 2 * (image1 / 255.0) - 1.0
        image2 = 2 * (image2 / 255.0) - 1.0
        fmap1, fmap2 = self.fnet([image1.contiguous(), image2.contiguous()])
        if self.mixed_precision:
            fmap1, fmap2 = fmap1.half(), fmap2.half()
        fmap1, fmap2 = fmap1.float(), fmap2.float()
        corr_fn = CorrBlock(fmap1, fmap2, num_levels=self.corr_levels, radius=self.corr_radius)
        cnet = self.cnet(image1)
        net, inp = torch.split(cnet, [self.hidden_dim, self.context_dim], dim=1)
        net = torch.tanh(net)
        inp = torch.relu(inp)
        N, _, H, W = image1.shape
        coords0 = coords_grid(N, H // 8, W // 8).to(image1.device)
        coords1 = coords_grid(N, H // 8, W // 8).to(image1.device)
        flow_predictions = []
        for _ in range(iters):
            coords1 = coords1.detach()
            corr = corr_fn(coords1)
            flow = coords1 - coords0
            net, up_mask, delta_flow = self.update_block(net, inp, corr, flow)
            coords1 = coords1 + delta_flow
            if upsample:
                flow_predictions.append(upsample_flow(coords1 - coords0, up_mask))
            else:
                flow_predictions.append(coords1 - coords0)
        if return_corr:
            return flow_predictions, net, inp, corr_fn
        return flow_predictions, net, inp
