"""Oracle (test infrastructure): PoseNet.infer / flow2depth restated on PyTorch-CPU.

Follows core/pose/pose_net.py:14-27 (construction), :60-85 (infer), :102-119 (get_weight_maps),
:121-125 (proj), :127-135 (flow2depth) of the reference.  ``infer`` is generalised from the
reference's hard-coded batch of one frame (``flow_predictions[-1][0]`` / ``[1]``, :66-67) to n frames by
splitting the RAFT batch in halves; for n == 1 it is the same computation.  The solve runs n
independent problems (SURVEY.md section 3.2 "batch coupling").
"""
import torch
import torch.nn as nn

from . import pose_head, warp
from .raft import RAFT
from .unet import TinyUNet


class PoseNet(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.loss_weight = nn.Parameter(torch.tensor([1.0, 1.0]))
        H, W = config['image_shape']
        self.use_weights = config.get('use_weights', True)
        self.lbgfs_iters = config.get('lbgfs_iters', 100)
        self.solver = config.get('solver', 'lbfgs')
        self.flow = RAFT(config)
        self.flow.freeze_bn()
        self.weight_head_2d = nn.Sequential(TinyUNet(128 + 128 + 8, (H, W)), nn.Sigmoid())
        self.weight_head_3d = nn.Sequential(TinyUNet(128 + 128 + 8 + 8, (H, W)), nn.Sigmoid())

    @torch.no_grad()
    def flow2depth(self, imagel, imager, baseline, upsample=True):
        flow = self.flow(imagel, imager, upsample=upsample)[0][-1]
        if upsample:
            depth, valid = warp.flow2depth(flow, baseline)
            return depth, flow, valid
        depth = baseline[:, None, None] / -flow[:, 0]          # 1/8-resolution flow in 1/8-pixel units
        depth = depth / 8.0                                    # "factor 8 of upsampling", pose_net.py:131-132
        valid = (depth > 0) & (depth <= 1.0)
        depth = torch.where(valid, depth, torch.ones_like(depth))
        return depth.unsqueeze(1), flow, valid.unsqueeze(1)

    @torch.no_grad()
    def stages(self, image1l, image2l, intrinsics, baseline, depth1, image2r, mask1, mask2, stereo_flow1):
        """Every intermediate of infer(), for stage-wise parity checks."""
        n = image1l.shape[0]
        ref_imgs = torch.cat((image1l, image2l), dim=0)
        trg_imgs = torch.cat((image2l, image2r), dim=0)
        flow_predictions, hidden, context = self.flow(ref_imgs, trg_imgs, upsample=True)
        time_flow = flow_predictions[-1][:n]
        stereo_flow2 = flow_predictions[-1][n:]
        hidden, context = hidden[:n], context[:n]
        depth2, valid = warp.flow2depth(stereo_flow2, baseline)
        mask2 = mask2 & valid
        pcl1 = warp.backproject(depth1, intrinsics)
        pcl2 = warp.backproject(depth2, intrinsics)
        pcl2w, mask2w, inp1, inp2 = warp.weight_inputs(pcl1, pcl2, image1l, image2l, mask2, time_flow,
                                                       stereo_flow1, stereo_flow2)
        if self.use_weights:
            w2d = self.weight_head_2d(torch.cat((inp1, hidden, context), dim=1))
            w3d = self.weight_head_3d(torch.cat((inp1, inp2, hidden, context), dim=1))
        else:
            w2d = torch.ones_like(mask2w, dtype=torch.float32)
            w3d = torch.ones_like(mask2w, dtype=torch.float32)
        return dict(time_flow=time_flow, stereo_flow2=stereo_flow2, hidden=hidden, context=context,
                    depth2=depth2, mask2=mask2, pcl1=pcl1, pcl2=pcl2, pcl2w=pcl2w, mask2w=mask2w,
                    inp1=inp1, inp2=inp2, w2d=w2d, w3d=w3d)

    @torch.no_grad()
    def infer(self, image1l, image2l, intrinsics, baseline, depth1, image2r, mask1, mask2, stereo_flow1,
              ret_details=False):
        s = self.stages(image1l, image2l, intrinsics, baseline, depth1, image2r, mask1, mask2, stereo_flow1)
        n = image1l.shape[0]
        lw = self.loss_weight.detach()[None, :].repeat(n, 1)
        args = (s['time_flow'], s['pcl1'], s['pcl2w'], s['w2d'], s['w3d'], mask1.bool(), s['mask2w'].bool(),
                intrinsics, lw)
        if self.solver == 'lbfgs':
            T, info = pose_head.lbfgs_solve(*args, iters=self.lbgfs_iters, coupled=False)
        else:
            T, info = pose_head.gn_solve(*args, iters=self.lbgfs_iters)
        vec7, log6 = pose_head.declarative_forward(T)
        pose = vec7[:, 0]
        if ret_details:
            return pose, depth1, s['depth2'], (s['w2d'], s['w3d']), s['time_flow'], s['stereo_flow2'], s['mask2'], info
        return pose
