"""Frame-to-frame tracker host mirror: ``PoseEstimator(config, intrinsics, baseline, checkpoint, img_shape)``
called once per stereo frame, as scripts/infer_trajectory.py:50-51,71-77 of the reference does.

Follows core/pose/pose_estimator.py:26-48 (checkpoint + config overrides, 1/depth_clipping scale),
:50-96 (forward: failure gate ``isnan | |log| > 0.1`` -> identity, de-normalise, chain
``last_pose <- last_pose * rel^-1``) and :98-125 (get_pose_f2f), and core/utils/frame_class.py:5-50 (Frame).
Frame-to-model tracking (``frame2frame: False``, SurfelMap) is out of scope.
"""
import time
import warnings
from collections import OrderedDict

import torch

from . import ops
from .pose_net import PoseNet
from .se3 import SE3


class Frame:
    """Carrier of img / rimg / depth / mask / confidence / flow between consecutive calls (frame_class.py)."""

    def __init__(self, img, rimg=None, depth=None, mask=None, confidence=None, flow=None):
        assert img.ndim == 4
        self.img = img.contiguous()
        self.rimg = img.contiguous() if rimg is None else rimg.contiguous()
        shape, dev = self.img.shape[-2:], self.img.device
        n = self.img.shape[0]
        self.mask = (torch.ones((n, 1, *shape), dtype=torch.bool, device=dev) if mask is None else mask).bool()
        self.depth = torch.ones((n, 1, *shape), device=dev) if depth is None else depth.contiguous()
        self.confidence = torch.ones((n, 1, *shape), device=dev) if confidence is None else confidence.contiguous()
        self.flow = torch.zeros((n, 2, *shape), device=dev) if flow is None else flow.contiguous()
        assert self.rimg.shape == self.img.shape
        for t in (self.depth, self.mask, self.confidence, self.flow):
            assert t.shape[-2:] == shape

    @property
    def shape(self):
        return self.img.shape[-2:]

    @property
    def device(self):
        return self.img.device

    def to(self, d):
        for k in ('img', 'rimg', 'depth', 'mask', 'confidence'):
            setattr(self, k, getattr(self, k).to(d))
        return self


class PoseEstimator(torch.nn.Module):
    def __init__(self, config, intrinsics, baseline, checkpoint, img_shape, init_pose=None):
        """config: the ``slam`` section of the inference YAML (configuration/infer_f2f.yaml:1-11);
        img_shape = (W, H) as in the reference; checkpoint: path, ``{'state_dict','config'}`` dict or a PoseNet."""
        super().__init__()
        if not config.get('frame2frame', True):
            raise NotImplementedError('frame-to-model tracking (SurfelMap) is out of scope')
        if isinstance(checkpoint, PoseNet):
            model = checkpoint
        else:
            ckp = torch.load(checkpoint, map_location='cpu') if isinstance(checkpoint, str) else checkpoint
            mcfg = dict(ckp['config']['model'])
            mcfg['image_shape'] = (img_shape[1], img_shape[0])          # pose_estimator.py:28
            mcfg['lbgfs_iters'] = config['lbgfs_iters']
            mcfg['use_weights'] = config['conf_weighing']
            if 'solver' in config:
                mcfg['solver'] = config['solver']
            model = PoseNet(mcfg)
            state = OrderedDict((k.replace('module.', ''), v) for k, v in ckp['state_dict'].items())
            model.load_state_dict(state)
        model.eval()
        self.model = model
        self.config = config
        self.register_buffer('intrinsics', intrinsics.unsqueeze(0).float(), persistent=False)
        self.register_buffer('scale', torch.tensor(1 / config['depth_clipping'][1]), persistent=False)
        self._inv_scale = float(1 / self.scale)                      # (f32 division, once, on the host: what `1 / self.scale` gives per call in the reference, :90)
        self.register_buffer('baseline', torch.tensor(baseline).unsqueeze(0).float(), persistent=False)
        self._init_pose = SE3.Identity(1) if init_pose is None else init_pose.float()
        self.last_pose = self._init_pose
        self.frame = None
        self.last_frame = None
        # streaming: encoder outputs of the current left image, reused as image1l's on the next call (exact: both
        # encoders normalise per sample); set reuse_features=False to re-encode like the reference does
        self.reuse_features = config.get('reuse_features', True)
        self._enc_cache = None
        self._pending = []                                            # submit() / result(): frames whose encoders are already on the side stream
        self._prefetch_stream = None

    @property
    def device(self):
        return self.intrinsics.device

    def reset(self):
        """Forget the sequence (frames, chained pose, cached encoder outputs): the next call is a first frame again."""
        self.last_pose = self._init_pose
        self.frame = self.last_frame = None
        self._enc_cache = None
        self._pending = []
        return self

    @torch.no_grad()
    def submit(self, limg, rimg, mask):
        """Pipelined form of ``forward`` for a caller that has the next frame before it needs this one's pose (a video file, a camera
        queue; scripts/infer_trajectory.py:57,71-77 iterates a DataLoader): ``submit(frame t+1)`` then ``result()`` -> frame t's pose.
        submit() puts the frame's encoder work (PoseNet.encode_frame: fnet on both images, cnet on the left one -- it depends on nothing
        else) on a side stream at once; the frame waits in a queue until result() runs the rest of ``forward`` for the oldest queued
        frame.  With frame t+1 submitted before result() is called for frame t, its encoders (~1.3 ms of launches) run beside frame t's
        update loop, whose ~100 launches per iteration leave most of the chip idle at batch 1.  Same kernels, same inputs: the poses are
        bit-identical to calling forward() frame by frame (tests/test_gpu_pipeline.py).  Requires reuse_features (the default)."""
        if not self.reuse_features:
            raise RuntimeError('submit / result reuse the encoder outputs across frames: construct the estimator with reuse_features=True')
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(limg.device))          # the images are ready on the caller's stream NOW (recorded later, the
        self._pending.append([limg, rimg, mask, None, None, ready])   #  event would sit behind the running frame's whole queue)
        if len(self._pending) == 1:
            self._start_encoders(self._pending[0])            # nothing is running that it could hide behind: start at once

    def _start_encoders(self, item):
        """Frame ``item``'s encoder work onto the side stream (once)."""
        if item[3] is not None:
            return
        limg, rimg = item[0], item[1]
        dev = limg.device
        if self._prefetch_stream is None or self._prefetch_stream.device != dev:
            self._prefetch_stream = torch.cuda.Stream(device=dev)
        side = self._prefetch_stream
        with torch.cuda.stream(side):
            side.wait_event(item[5])
            item[3] = self.model.encode_frame(limg, rimg)
            item[4] = torch.cuda.Event()
            item[4].record(side)
        for t in (limg, rimg):
            t.record_stream(side)

    @torch.no_grad()
    def result(self):
        """``forward`` for the oldest submitted frame, on its prefetched encoder outputs.  The NEXT queued frame's encoders are started
        once this frame's work has been handed to the GPU and before the host waits for its success flag: they then run beside the tail
        of this frame -- the last update iterations, the weight heads and the 20-iteration solve, whose launches occupy a few
        workgroups each."""
        if not self._pending:
            raise RuntimeError('result() without a submitted frame')
        self._start_encoders(self._pending[0])
        limg, rimg, mask, enc, done, _ = self._pending.pop(0)
        cur = torch.cuda.current_stream(limg.device)
        cur.wait_event(done)
        for t in enc.values():
            t.record_stream(cur)                              # allocated on the side stream, consumed (and freed) on the caller's
        return self.forward(limg, rimg, mask, _enc=enc)

    @torch.no_grad()
    def forward(self, limg, rimg, mask, _enc=None):
        """limg, rimg: (1,3,h,w) 0..255; mask: (1,1,h,w) True = valid.  Returns (absolute pose SE3, None, flow, weights)."""
        self.last_pose = self.last_pose.to(limg.device)
        self.last_frame = self.frame
        self.frame = Frame(limg, rimg, mask=mask)
        rel_pose, ret_frame, flow, weights = self.get_pose_f2f(_enc)
        # :81-91 in one launch (ops.pose_gate_chain): the gate isnan | |log| > 0.1 -> identity, de-normalisation of the depth scaling and
        # last_pose <- last_pose * rel^-1, with ONE host synchronisation (the success flag) instead of a dozen element-wise launches and two
        rel, pose, ok = ops.pose_gate_chain(rel_pose.data.reshape(1, 7), self.last_pose.data, self._inv_scale, 1.0e-1)
        if self._pending:
            self._start_encoders(self._pending[0])            # (submit / result) the next frame's encoders, before the host waits for this one
        self.t_enqueued = time.perf_counter()             # everything of this frame has been handed to the runtime; what follows waits for the GPU
        self.success = bool(ok[0])
        if not self.success:
            warnings.warn('pose estimation not converged, skip.', RuntimeWarning)                 # :82
        self.last_rel_pose = SE3(rel)
        self.last_frame = ret_frame
        self.last_pose = SE3(pose)
        return self.last_pose, None, flow, weights

    @torch.no_grad()
    def forward_chunk(self, limgs, rimgs, masks):
        """``forward`` for c consecutive frames in ONE pass (PoseNet.infer_chunk): limgs, rimgs (c,3,h,w), masks (c,1,h,w), after at
        least one ``forward`` call (the sequence's first frame only gets its stereo depth).  Bit-identical to c single calls: same
        relative poses, same gate decisions, same chained poses, same Frame left behind.  Returns the (c,7) absolute poses;
        ``last_rel_poses`` (c,7) gated relative poses and ``successes`` (c,) bool are left on the estimator, the gate is evaluated on
        the device with ONE host synchronisation per chunk (for the warnings) instead of two per frame."""
        if self.frame is None:
            raise RuntimeError('forward_chunk: call forward() on the first frame of the sequence (it has no predecessor to pair with)')
        c = limgs.shape[0]
        prev = self.frame
        masks = masks.bool().contiguous()
        self.last_pose = self.last_pose.to(limgs.device)
        vec7, depth2, weights, flow, stereo_flow, cache = self.model.infer_chunk(
            prev.img, limgs, rimgs, self.intrinsics, self.baseline * self.scale, depth0=prev.depth * self.scale, mask0=prev.mask,
            masks=masks, stereo_flow0=prev.flow, cache0=self._enc_cache, depth_roundtrip=self.scale)
        self._enc_cache = cache if self.reuse_features else None
        rel, poses, ok = ops.pose_gate_chain(vec7.reshape(c, 7), self.last_pose.data, self._inv_scale, 1.0e-1)   # :81-91, every frame
        bad = ok == 0
        n_bad = int(bad.sum())                                             # the chunk's one host synchronisation
        for _ in range(n_bad):
            warnings.warn('pose estimation not converged, skip.', RuntimeWarning)
        self.successes = ~bad
        self.success = not bool(bad[-1]) if n_bad else True
        self.last_rel_poses = rel
        self.last_rel_pose = SE3(rel[c - 1:])
        self.last_pose = SE3(poses[c - 1:])
        self.last_frame = Frame(limgs[c - 2:c - 1], rimgs[c - 2:c - 1]) if c > 1 else prev      # (only .img of it is ever read again)
        self.frame = Frame(limgs[c - 1:], rimgs[c - 1:], depth=depth2[c - 1:] / self.scale, mask=masks[c - 1:], flow=stereo_flow[c - 1:])
        return poses, None, flow, weights

    def get_pose_f2f(self, enc=None):
        flow = None
        if self.last_frame is None:
            rel = SE3.IdentityLike(self.last_pose)
            depth, stereo_flow, valid, cache = self.model.flow2depth(self.frame.img, self.frame.rimg,
                                                                     self.baseline * self.scale, ret_cache=True, **({'enc': enc} if enc is not None else {}))
            self._enc_cache = cache if self.reuse_features else None
            self.frame.depth = depth / self.scale
            self.frame.flow = stereo_flow
            return rel, None, None, None
        rel, depth1, depth2, weights, flow, stereo_flow, cache = self.model.infer(
            self.last_frame.img, self.frame.img, self.intrinsics, self.baseline * self.scale,
            depth1=self.last_frame.depth * self.scale, image2r=self.frame.rimg, mask1=self.last_frame.mask,
            mask2=self.frame.mask, stereo_flow1=self.last_frame.flow, ret_details=True,
            cache1=self._enc_cache, ret_cache=True, **({'enc2': enc} if enc is not None and self._enc_cache is not None else {}))
        self._enc_cache = cache if self.reuse_features else None
        rel = SE3(rel.data.reshape(1, 7))
        self.frame.depth = depth2 / self.scale
        self.frame.flow = stereo_flow
        return rel, self.last_frame, flow, weights
