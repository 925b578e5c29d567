"""PoseNet host mirror (boundary B3): ``PoseNet(config).infer(image1l, image2l, intrinsics, baseline, depth1,
image2r, mask1, mask2, stereo_flow1, ret_details)`` and ``.flow2depth(imagel, imager, baseline)`` with the
reference's argument meaning (core/pose/pose_net.py:14-27,60-85,102-135) and parameter names (checkpoints load
unchanged: ``flow.*``, ``weight_head_2d.0.*``, ``weight_head_3d.0.*``, ``loss_weight``).

Per call: one batch-2n RAFT pass (hand-written HIP throughout: raft.py), one fused HIP pass for depth + back-projection +
the four warps + both 1/8 stacks, both TinyUNet heads as one HIP kernel chain (csrc/unet.hip; PyTorch-ROCm ops only when
they train), and the device-resident SE(3) solve.  ``infer`` is generalised from the reference's hard-coded single frame
(``flow_predictions[-1][0]`` / ``[1]``, :66-67) to n frames by splitting the RAFT batch in halves.
"""
import torch
import torch.nn as nn

from . import ops
from .pose_head import DeclarativeLayerLie, DPoseSE3Head, create_img_coords_t
from .raft import RAFT
from .se3 import SE3
from .unet import TinyUNet, pack_params

FUSED_HEADS = True      # the heads on the HIP kernel chain (csrc/unet.hip); tools/ set this to False for A/B runs against the module route


class PoseNet(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.loss_weight = nn.Parameter(torch.tensor([1.0, 1.0]))
        H, W = config['image_shape']
        self.register_buffer('img_coords', create_img_coords_t(y=H, x=W), persistent=False)
        self.use_weights = config.get('use_weights', True)
        self.flow = RAFT(config)
        self.flow.freeze_bn()
        self.pose_head = DeclarativeLayerLie(DPoseSE3Head(self.img_coords, config.get('lbgfs_iters', 100),
                                                          solver=config.get('solver', 'lbfgs')))
        self.weight_head_2d = nn.Sequential(TinyUNet(in_channels=128 + 128 + 8, output_size=(H, W)), nn.Sigmoid())
        self.weight_head_3d = nn.Sequential(TinyUNet(in_channels=128 + 128 + 8 + 8, output_size=(H, W)), nn.Sigmoid())

    def train(self, mode=True):
        super().train(mode)
        self.flow.eval()
        return self

    @torch.no_grad()
    def encode_frame(self, imagel, imager):
        """The encoder work of one new stereo frame -- fnet(left | right) and cnet(left) -- as ``infer(..., cache1=..., enc2=...)`` and
        ``flow2depth(..., enc=...)`` accept it: {'f': (2n,256,h/8,w/8), 'c': (n,256,h/8,w/8)}.  It depends on the frame's two images
        only, so a tracker can run it for frame t+1 (on a side stream) while frame t's update loop is still on the GPU
        (PoseEstimator.submit / result); the kernels and their inputs are the same, the results bit-identical."""
        f, c = self._encode_both((imagel, imager), imagel)
        return dict(f=f, c=c)

    @torch.no_grad()
    def flow2depth(self, imagel, imager, baseline, upsample=True, ret_cache=False, enc=None):
        """pose_net.py:127-135 -> (depth (n,1,h,w), stereo flow (n,2,h,w), valid (n,1,h,w) bool).
        ``upsample=False``: everything at 1/8 resolution, the flow in 1/8-pixel units and the depth divided by 8 (:131-132;
        a division by a power of two commutes with the rounding of b / -flow.x, so the kernel is handed b / 8).
        ``ret_cache`` additionally returns the encoder outputs of ``imagel`` for reuse by the next ``infer``."""
        n = imagel.shape[0]
        f, cn = (enc['f'], enc['c']) if enc is not None else self._encode_both((imagel, imager), imagel)
        flow = self.flow(None, None, upsample=upsample, fmaps=(f[:n], f[n:]), cnet=cn)[0][-1]
        depth, valid = ops.flow2depth(flow, baseline if upsample else baseline / 8.0)
        if ret_cache:
            return depth, flow, valid, dict(fmap=f[:n], cnet=cn)
        return depth, flow, valid

    def forward(self, image1l, image2l, intrinsics, baseline, image1r, image2r, mask1=None, mask2=None, ret_confmap=False):
        """The reference's TRAINING forward (core/pose/pose_net.py:28-58): both stereo depths, the temporal flow, the weight
        maps and the declarative pose layer -> (log-pose (n,6), depth1, depth2[, (w2d, w3d)]).  Gradients reach what the
        reference trains while the flow network is frozen (train.yaml: freeze_flow_steps; RAFT.eval() always): the two
        weight heads and ``loss_weight``, through the layer's closed-form backward (csrc/pose_backward.hip).  Flow, depth
        and the warps are computed by the (non-differentiable) HIP kernels, i.e. this mirrors the frozen-flow phase."""
        n = image1l.shape[0]
        intrinsics = intrinsics.expand(n, 3, 3).contiguous()
        baseline = baseline.expand(n).contiguous()
        with torch.no_grad():
            depth1, stereo_flow1, valid1 = self.flow2depth(image1l, image1r, baseline)
            mask1 = (mask1.bool() & valid1) if mask1 is not None else valid1
            mask2 = mask2.bool().clone() if mask2 is not None else torch.ones_like(valid1)
            s = self.stages(image1l, image2l, intrinsics, baseline, depth1, image2r, mask1, mask2, stereo_flow1, heads=False)
        if self.use_weights:
            w2d = self.weight_head_2d(torch.cat((s['inp1'], s['hidden'], s['context']), dim=1))
            w3d = self.weight_head_3d(torch.cat((s['inp1'], s['inp2'], s['hidden'], s['context']), dim=1))
        else:
            w2d, w3d = torch.ones_like(s['depth2']), torch.ones_like(s['depth2'])
        _, log6 = self.pose_head(s['time_flow'], s['pcl1'], s['pcl2w'], w2d, w3d, mask1, s['mask2w'], s['intrinsics'],
                                 self.loss_weight.repeat(n, 1))
        pose_tan = log6.squeeze(1)
        if ret_confmap:
            return pose_tan, depth1, s['depth2'], (w2d, w3d)
        return pose_tan, depth1, s['depth2']

    def freeze_flow(self, freeze=True):
        """pose_net.py:148-153.  Un-freezing the flow network needs its backward, which is not built (SURVEY.md 8f-4 covers the
        declarative layer): refuse instead of training the heads against silently missing flow gradients."""
        if not freeze:
            raise NotImplementedError('training the flow network (freeze_flow(False)) is out of scope: RAFT has no backward here')
        for p in self.parameters():
            p.requires_grad = True
        for p in self.flow.parameters():
            p.requires_grad = False
        return self

    @torch.no_grad()
    def _encode_both(self, feature_images, context_images):
        """RAFT.encode_both (the two encoders side by side on two streams), or the two calls for a flow module that only has them."""
        both = getattr(self.flow, 'encode_both', None)
        if both is not None:
            return both(feature_images, context_images)
        return self.flow.encode_features(feature_images), self.flow.encode_context(context_images)

    @torch.no_grad()
    def stages(self, image1l, image2l, intrinsics, baseline, depth1, image2r, mask1, mask2, stereo_flow1, cache1=None, heads=True, enc2=None):
        """Every stage of infer() before the solve.  ``cache1`` = {'fmap','cnet'} of image1l from the previous call
        (streaming: frame t's image2l is frame t+1's image1l), so only the two new images are encoded -- or not even those when
        ``enc2`` = encode_frame(image2l, image2r) was computed ahead of the call (needs cache1)."""
        n = image1l.shape[0]
        intrinsics = intrinsics.expand(n, 3, 3).contiguous()
        baseline = baseline.expand(n).contiguous()
        # The reference's batch-2 RAFT call is (image1l, image2l) -> (image2l, image2r) (pose_net.py:63-64).  image2l sits in both
        # halves and the feature encoder normalises per sample, so it is encoded once: f = (f1l | f2l | f2r) in ONE encoder batch
        # whose overlapping slices f[:2n], f[n:] ARE the two operand batches of the correlation -- no torch.cat of images or maps
        if cache1 is None:
            f, cn = self._encode_both((image1l, image2l, image2r), (image1l, image2l))
            f2l = f[n:2 * n]
            fmaps = (f[:2 * n], f[n:])
            c2l = cn[n:]
        else:
            f, c2l = (enc2['f'], enc2['c']) if enc2 is not None else self._encode_both((image2l, image2r), image2l)
            f2l = f[:n]
            fmaps = (torch.cat((cache1['fmap'], f2l), dim=0), f)
            cn = torch.cat((cache1['cnet'], c2l), dim=0)
        flow_predictions, hidden, context = self.flow(None, None, upsample=True, fmaps=fmaps, cnet=cn)
        time_flow = flow_predictions[-1][:n].contiguous()
        stereo_flow2 = flow_predictions[-1][n:].contiguous()
        hidden, context = hidden[:n], context[:n]
        g = ops.depth_backproject_warp(stereo_flow2, time_flow, baseline, intrinsics, depth1, image1l, image2l,
                                       stereo_flow1, mask2)
        if not heads:
            w2d = w3d = None
        elif self.use_weights and FUSED_HEADS and not (self.weight_head_2d.training or self.weight_head_3d.training):
            # both heads + resize + sigmoid as one hand-written kernel chain (csrc/unet.hip), no concatenations
            w2d, w3d = ops.unet_heads(g['inp1'], g['inp2'], hidden, context, pack_params(self.weight_head_2d[0]),
                                      pack_params(self.weight_head_3d[0]), self.config['image_shape'])
        elif self.use_weights:
            w2d = self.weight_head_2d(torch.cat((g['inp1'], hidden, context), dim=1))
            w3d = self.weight_head_3d(torch.cat((g['inp1'], g['inp2'], hidden, context), dim=1))
        else:
            w2d = torch.ones_like(g['depth2'])
            w3d = torch.ones_like(g['depth2'])
        g.update(time_flow=time_flow, stereo_flow2=stereo_flow2, hidden=hidden, context=context, w2d=w2d, w3d=w3d,
                 intrinsics=intrinsics, cache2=dict(fmap=f2l, cnet=c2l))
        return g

    @torch.no_grad()
    def infer(self, image1l, image2l, intrinsics, baseline, depth1, image2r, mask1, mask2, stereo_flow1,
              ret_details=False, cache1=None, ret_cache=False, enc2=None):
        if enc2 is not None and cache1 is None:
            raise ValueError('infer: enc2 (the new frame encoded ahead of the call) needs cache1 (the previous frame\'s encoder outputs)')
        s = self.stages(image1l, image2l, intrinsics, baseline, depth1, image2r, mask1, mask2, stereo_flow1, cache1, enc2=enc2)
        mask2.copy_(s['mask2'])                               # `mask2 &= valid` mutates the caller's tensor (:77)
        n = image1l.shape[0]
        lw = self.loss_weight.detach()[None, :].repeat(n, 1)
        vec7, _ = self.pose_head(s['time_flow'], s['pcl1'], s['pcl2w'], s['w2d'], s['w3d'], mask1.bool(), s['mask2w'],
                                 s['intrinsics'], lw)
        pose = SE3(vec7[:, 0]) if n > 1 else SE3(vec7)[0]     # reference returns SE3(pose_se3)[0] for its n == 1
        out = (pose, depth1, s['depth2'], (s['w2d'], s['w3d']), s['time_flow'], s['stereo_flow2']) if ret_details else pose
        if ret_cache:
            return (*out, s['cache2']) if ret_details else (out, s['cache2'])
        return out

    @torch.no_grad()
    def infer_chunk(self, image0l, imagesl, imagesr, intrinsics, baseline, depth0, mask0, masks, stereo_flow0, cache0=None,
                    depth_roundtrip=None):
        """c consecutive calls of ``infer`` as ONE pass: frame t of the chunk is ``infer(image1l = frame t-1, image2l = frame t, ...)``
        with frame -1 = (image0l, depth0, mask0, stereo_flow0) and, for t > 0, depth1 / mask1 / stereo_flow1 = the depth2 / mask2 /
        stereo_flow2 that call t-1 produced (core/pose/pose_estimator.py:113-122 feeds them back through its Frame) -- a shift by one
        row INSIDE the batch, after the 2c RAFT pairs, which are independent.  imagesl, imagesr (c,3,h,w) 0..255; masks (c,1,h,w) bool,
        ANDed in place with the stereo validity like ``infer``'s mask2 (pose_net.py:77); depth0 / stereo_flow0 / mask0 (1,..) as ``infer``
        takes them; ``cache0`` = encoder outputs of image0l ({'fmap','cnet'}) when the caller has them.
        ``depth_roundtrip`` = s: depth1 of row t > 0 is (depth2[t-1] / s) * s, the two roundings PoseEstimator's Frame puts between
        consecutive calls (it stores depth / scale and hands over depth * scale, pose_estimator.py:107,116,121).
        Every kernel on the way computes a row independently of its batch (tests), and the solve runs with partition_rows = 1, so the
        result is BIT-IDENTICAL to the c single calls.  Returns (vec7 (c,7) f32, depth2 (c,1,h,w), (w2d, w3d), time_flow (c,2,h,w),
        stereo_flow2 (c,2,h,w), cache of the last frame)."""
        c = imagesl.shape[0]
        intrinsics = intrinsics.expand(c, 3, 3).contiguous()
        baseline = baseline.expand(c).contiguous()
        f, cn = self._encode_both((imagesl, imagesr), imagesl)          # f = (L_0..L_c-1 | R_0..R_c-1)
        fl = f[:c]
        if cache0 is None:
            f0, c0 = self.flow.encode_features(image0l), self.flow.encode_context(image0l)
        else:
            f0, c0 = cache0['fmap'], cache0['cnet']
        # pairs: temporal (L_t-1 -> L_t) for t = 0..c-1, then stereo (L_t -> R_t): the order PoseNet.stages uses per frame
        fmap1 = torch.cat((f0, fl[:c - 1], fl), dim=0)
        cnet = torch.cat((c0, cn[:c - 1], cn), dim=0)
        flows, hidden, context = self.flow(None, None, upsample=True, fmaps=(fmap1, f), cnet=cnet)
        time_flow = flows[-1][:c].contiguous()
        stereo_flow2 = flows[-1][c:].contiguous()
        hidden, context = hidden[:c], context[:c]
        depth2, valid = ops.flow2depth(stereo_flow2, baseline)            # the same division and tests rpe_depth_backproject_warp repeats
        masks &= valid
        d_prev = depth2[:c - 1]
        if depth_roundtrip is not None:
            d_prev = (d_prev / depth_roundtrip) * depth_roundtrip
        depth1 = torch.cat((depth0, d_prev), dim=0)
        mask1 = torch.cat((mask0.bool(), masks[:c - 1]), dim=0)
        stereo_flow1 = torch.cat((stereo_flow0, stereo_flow2[:c - 1]), dim=0)
        image1l = torch.cat((image0l, imagesl[:c - 1]), dim=0)
        g = ops.depth_backproject_warp(stereo_flow2, time_flow, baseline, intrinsics, depth1, image1l, imagesl, stereo_flow1, masks)
        if self.use_weights and not FUSED_HEADS:
            # the A/B switch selects the module route in stages(); here it would silently break forward_chunk == c forward calls
            raise RuntimeError('infer_chunk runs the fused weight heads only: set pose_net.FUSED_HEADS = True (or walk the frames with infer)')
        if self.use_weights and not (self.weight_head_2d.training or self.weight_head_3d.training):
            w2d, w3d = ops.unet_heads(g['inp1'], g['inp2'], hidden, context, pack_params(self.weight_head_2d[0]),
                                      pack_params(self.weight_head_3d[0]), self.config['image_shape'])
        elif self.use_weights:
            raise RuntimeError('infer_chunk: the weight heads must be in eval mode')
        else:
            w2d, w3d = torch.ones_like(g['depth2']), torch.ones_like(g['depth2'])
        lw = self.loss_weight.detach()[None, :].repeat(c, 1)
        problem = self.pose_head.problem
        keep, problem.partition_rows = problem.partition_rows, 1
        try:
            vec7, _ = self.pose_head(time_flow, g['pcl1'], g['pcl2w'], w2d, w3d, mask1, g['mask2w'], intrinsics, lw)
        finally:
            problem.partition_rows = keep
        return vec7[:, 0], g['depth2'], (w2d, w3d), time_flow, stereo_flow2, dict(fmap=fl[c - 1:], cnet=cn[c - 1:])

    def init_from_raft(self, raft_ckp):
        state = torch.load(raft_ckp, map_location='cpu')
        self.flow.load_state_dict({k.replace('module.', ''): v for k, v in state.items()})
        return self
