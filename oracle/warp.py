"""Oracle (test infrastructure): flow warps, disparity->depth, back-projection, 1/8 down-sampling.

Restates
  * remap_from_flow / remap_from_flow_nearest   <- core/interpol/flow_utils.py:4-26
  * flow -> depth (+valid)                      <- core/pose/pose_net.py:73-77 and :127-135
  * PoseNet.proj                                <- core/pose/pose_net.py:121-125
  * the 1/8 bilinear stacks of get_weight_maps  <- core/pose/pose_net.py:110-113
(paths relative to /root/reference) twice: once through the same torch calls the reference makes
(``*_torch``) and once with every float operation and every integer tap index written out
(``sample_taps``), so that the HIP kernels' pixel indices can be compared bit for bit.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .pose_head import img_coords


# ------------------------------------------------------------------ the reference's own torch calls
def _grid(flow):
    n, _, h, w = flow.shape
    row, col = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    g = torch.empty_like(flow)
    g[:, 1] = 2 * (flow[:, 1] + row) / (h - 1) - 1           # flow_utils.py:9
    g[:, 0] = 2 * (flow[:, 0] + col) / (w - 1) - 1           # flow_utils.py:10
    return g.permute(0, 2, 3, 1)


def remap_from_flow(x, flow):
    x = F.grid_sample(x, _grid(flow), align_corners=True)    # flow_utils.py:11 (bilinear, zero pad)
    valid = (x > 0).any(dim=1).unsqueeze(1)
    return x, valid


def remap_from_flow_nearest(x, flow):
    x = F.grid_sample(x.float(), _grid(flow), align_corners=True, mode='nearest')   # flow_utils.py:24
    valid = (x > 0).any(dim=1).unsqueeze(1)
    return x, valid


# ------------------------------------------------------------------ explicit arithmetic + integer taps
def source_coords(flow):
    """Un-normalised sampling position exactly as grid_sample computes it from flow_utils' grid:
    g = 2*(flow+idx)/(size-1) - 1  (float32), then ((g + 1) / 2) * (size - 1) (float32)."""
    f = flow.numpy().astype(np.float32)
    n, _, h, w = f.shape
    col = np.arange(w, dtype=np.float32)[None, None, :]
    row = np.arange(h, dtype=np.float32)[None, :, None]
    two = np.float32(2.0)
    one = np.float32(1.0)
    gx = two * (f[:, 0] + col) / np.float32(w - 1) - one
    gy = two * (f[:, 1] + row) / np.float32(h - 1) - one
    ix = ((gx + one) / two) * np.float32(w - 1)
    iy = ((gy + one) / two) * np.float32(h - 1)
    return ix.astype(np.float32), iy.astype(np.float32)


def sample_taps(flow):
    """Integer tap indices of both warps: bilinear (x0,y0 = floor) and nearest (round half to even)."""
    ix, iy = source_coords(flow)
    x0 = np.floor(ix).astype(np.int64)
    y0 = np.floor(iy).astype(np.int64)
    xn = np.rint(ix).astype(np.int64)                        # nearbyint, ties to even
    yn = np.rint(iy).astype(np.int64)
    return dict(ix=ix, iy=iy, x0=x0, y0=y0, xn=xn, yn=yn)


def remap_explicit(x, flow, nearest=False):
    """grid_sample written out (zero padding, align_corners=True); float32 throughout."""
    xs = x.numpy().astype(np.float32)
    n, c, h, w = xs.shape
    t = sample_taps(flow)
    out = np.zeros_like(xs)
    bi = np.arange(n)[:, None, None]
    if nearest:
        xn, yn = t['xn'], t['yn']
        ok = (xn >= 0) & (xn < w) & (yn >= 0) & (yn < h)
        xc, yc = np.clip(xn, 0, w - 1), np.clip(yn, 0, h - 1)
        for ch in range(c):
            out[:, ch] = np.where(ok, xs[bi, ch, yc, xc], np.float32(0))
        return torch.from_numpy(out)
    x0, y0, ix, iy = t['x0'], t['y0'], t['ix'], t['iy']
    x1, y1 = x0 + 1, y0 + 1
    # weights as in ATen's grid_sampler: nw = (ix_se - ix)*(iy_se - iy) ...
    fx0, fy0 = x0.astype(np.float32), y0.astype(np.float32)
    fx1, fy1 = x1.astype(np.float32), y1.astype(np.float32)
    nw = (fx1 - ix) * (fy1 - iy)
    ne = (ix - fx0) * (fy1 - iy)
    sw = (fx1 - ix) * (iy - fy0)
    se = (ix - fx0) * (iy - fy0)

    def tap(ch, yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        v = xs[bi, ch, np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return np.where(ok, v, np.float32(0))

    for ch in range(c):
        acc = tap(ch, y0, x0) * nw
        acc = acc + tap(ch, y0, x1) * ne
        acc = acc + tap(ch, y1, x0) * sw
        acc = acc + tap(ch, y1, x1) * se
        out[:, ch] = acc
    return torch.from_numpy(out)


# ------------------------------------------------------------------ depth / back-projection
def flow2depth(stereo_flow, baseline):
    """pose_net.py:73-76 / :130-135: depth = b / -flow.x ; valid = 0 < d <= 1 ; d[~valid] = 1."""
    depth = baseline[:, None, None] / -stereo_flow[:, 0]
    valid = (depth > 0) & (depth <= 1.0)
    depth = torch.where(valid, depth, torch.ones_like(depth))
    return depth.unsqueeze(1), valid.unsqueeze(1)


def backproject(depth, K):
    """PoseNet.proj (pose_net.py:121-125): depth * (K^-1 @ [x+.5, y+.5, 1])."""
    n, _, h, w = depth.shape
    rep = torch.linalg.inv(K) @ img_coords(h, w).view(1, 3, -1)
    return (depth.view(n, 1, -1) * rep).view(n, 3, h, w)


def down8(x):
    """F.interpolate(scale_factor=0.125, mode='bilinear') (pose_net.py:110-113)."""
    return F.interpolate(x, scale_factor=0.125, mode='bilinear')


def weight_inputs(pcl1, pcl2, image1l, image2l, mask2, time_flow, stereo_flow1, stereo_flow2):
    """The data half of PoseNet.get_weight_maps (pose_net.py:102-113): warps + the two 8-channel 1/8 stacks."""
    pcl2w, _ = remap_from_flow(pcl2, time_flow)
    image2w, _ = remap_from_flow(image2l, time_flow)
    sflow2w, _ = remap_from_flow(stereo_flow2, time_flow)
    m2w, valid_mapping = remap_from_flow_nearest(mask2, time_flow)
    mask2w = valid_mapping & m2w.to(bool)
    inp1 = down8(torch.cat((stereo_flow1, image1l, pcl1), dim=1))
    inp2 = down8(torch.cat((sflow2w, image2w, pcl2w), dim=1))
    return pcl2w, mask2w, inp1, inp2
