"""Oracle (test infrastructure): frame-to-frame tracker, restating
core/pose/pose_estimator.py:40-43 (1/250 scale), :50-96 (forward: failure gate, de-normalise, chain)
and :98-125 (get_pose_f2f) of the reference on top of oracle.pose_net.PoseNet."""
import torch

from . import se3 as _se3


class PoseEstimator:
    def __init__(self, model, intrinsics, baseline, depth_clip=250.0, init_pose=None):
        self.model = model
        self.intrinsics = intrinsics.unsqueeze(0).float()
        self.scale = torch.tensor(1.0 / depth_clip)
        self.baseline = torch.tensor(baseline).unsqueeze(0).float()
        self.last_pose = torch.tensor([[0, 0, 0, 0, 0, 0, 1.0]]) if init_pose is None else init_pose.float()
        self.frame = None
        self.last_frame = None
        self.rel_poses = []
        self.success = []

    def forward(self, limg, rimg, mask):
        self.last_frame = self.frame
        self.frame = dict(img=limg.contiguous(), rimg=rimg.contiguous(), mask=mask.bool().clone(),
                          depth=None, flow=None)
        if self.last_frame is None:
            rel = torch.tensor([[0, 0, 0, 0, 0, 0, 1.0]])
            depth, sflow, valid = self.model.flow2depth(limg, rimg, self.baseline * self.scale)
            self.frame['depth'] = depth / self.scale
            self.frame['flow'] = sflow
        else:
            lf = self.last_frame
            rel, _, depth2, _, _, sflow2, mask2, _ = self.model.infer(
                lf['img'], self.frame['img'], self.intrinsics, self.baseline * self.scale,
                depth1=lf['depth'] * self.scale, image2r=self.frame['rimg'], mask1=lf['mask'],
                mask2=self.frame['mask'], stereo_flow1=lf['flow'], ret_details=True)
            self.frame['mask'] = mask2                     # pose_net.py:77 mutates Frame.mask in place
            self.frame['depth'] = depth2 / self.scale
            self.frame['flow'] = sflow2
        rel = rel.reshape(1, 7)
        log = _se3.se3_log(rel)
        if bool(torch.isnan(rel).any()) or bool((log.abs() > 1.0e-1).any()):     # :81
            rel = torch.tensor([[0, 0, 0, 0, 0, 0, 1.0]])
            ok = False
        else:
            ok = True
        self.rel_poses.append(rel.clone())
        self.success.append(ok)
        rel = torch.cat((rel[:, :3] * (1.0 / self.scale), rel[:, 3:]), dim=-1)      # :90 scale(1/scale)
        self.last_pose = _se3.se3_mul(self.last_pose, _se3.se3_inv(rel))           # :91
        return self.last_pose


def chain(rel_poses, scale=250.0, init=None):
    """Prefix product P_t = P_{t-1} * (scale(rel_t))^-1 over an (m,7) stack of relative poses."""
    P = torch.tensor([[0, 0, 0, 0, 0, 0, 1.0]], dtype=rel_poses.dtype) if init is None else init.reshape(1, 7)
    out = []
    for r in rel_poses.reshape(-1, 1, 7):
        r = torch.cat((r[:, :3] * scale, r[:, 3:]), dim=-1)
        P = _se3.se3_mul(P, _se3.se3_inv(r))
        out.append(P[0])
    return torch.stack(out)
