"""One-process-per-GPU sharding of a stereo sequence.

In frame-to-frame mode the relative pose t-1 -> t depends only on the two stereo pairs
(core/pose/pose_estimator.py:98-125); the only cross-frame state is the chained absolute pose (:91), an
associative SE(3) product.  So a sequence of F frames is cut into contiguous blocks of relative poses, one
block per rank, each with a one-frame halo (the predecessor of its first pair, whose stereo depth it needs).
There is NO collective on the data path; the single exchange is one all-gather of the (frames, 7) float32
relative poses + success flags (RCCL over xGMI with backend "nccl", gloo in the CPU tests), after which the
failure gate (:81-87) and the prefix product (:90-91) run where the trajectory is wanted.

Everything here is host logic on torch.distributed; the per-pair solve is injected, so the CPU tests can drive
it with the oracle while the GPU path drives it with PoseEstimator / the HIP kernels.
"""
import torch
import torch.distributed as dist


def block_partition(n_items, world):
    """Contiguous, near-equal blocks: returns [(start, end)] * world covering range(n_items)."""
    base, rem = divmod(n_items, world)
    out, s = [], 0
    for r in range(world):
        e = s + base + (1 if r < rem else 0)
        out.append((s, e))
        s = e
    return out


def gather_relative_poses(rel_local, ok_local, sizes, group=None):
    """all_gather of variable-length blocks: pad to the longest block, gather, strip the padding.
    rel_local (m_r,7) float32, ok_local (m_r,) bool.  Returns (rel (sum m,7), ok (sum m,)) on every rank."""
    if not dist.is_initialized():                      # no process group at all: a plain serial run
        return rel_local, ok_local
    world = dist.get_world_size(group)                 # world 1 still goes through the collective (RCCL path under test)
    mmax = max(sizes)
    dev = rel_local.device
    pad = torch.zeros(mmax, 8, dtype=torch.float32, device=dev)
    m = rel_local.shape[0]
    if m:
        pad[:m, :7] = rel_local.float()
        pad[:m, 7] = ok_local.float()
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    rel = torch.cat([b[:sz, :7] for b, sz in zip(bufs, sizes)])
    ok = torch.cat([b[:sz, 7] > 0.5 for b, sz in zip(bufs, sizes)])
    return rel, ok


def failure_gate(rel, log, thr=1.0e-1):
    """pose_estimator.py:81-87: NaN pose or any |log| > 0.1 -> failed, substitute identity."""
    bad = torch.isnan(rel).any(dim=-1) | (log.abs() > thr).any(dim=-1)
    ident = torch.zeros_like(rel)
    ident[..., 6] = 1.0
    return torch.where(bad[..., None], ident, rel), ~bad


def track_sharded(n_frames, run_block, chain_fn, rank=0, world=1, group=None, scale=None):
    """Sharded tracking of frames 0..n_frames-1.

    run_block(first_pair, last_pair) -> (rel (m,7) f32, ok (m,) bool) computes the GATED relative poses of
    pairs first_pair..last_pair-1, pair t being (frame t, frame t+1); it is responsible for the halo frame.
    chain_fn(rel (M,7), scale) -> (M,7) absolute poses: the prefix product P_k = P_{k-1} * inv(scale(rel_k)).
    ``scale`` = the depth-clipping distance the relative poses were normalised by (1 / PoseEstimator.scale,
    pose_estimator.py:40-43,90); required -- a default would silently disagree with a config's depth_clipping.
    Returns absolute poses (n_frames,7): identity for frame 0 followed by the chained poses.
    """
    if scale is None:
        raise ValueError('track_sharded: scale (the depth_clipping distance) is required')
    blocks = block_partition(max(n_frames - 1, 0), world)
    s, e = blocks[rank]
    rel_local, ok_local = run_block(s, e)
    rel, ok = gather_relative_poses(rel_local, ok_local, [b[1] - b[0] for b in blocks], group)
    ident = torch.zeros(1, 7, dtype=rel.dtype, device=rel.device)
    ident[0, 6] = 1.0
    if rel.shape[0] == 0:
        return ident, rel, ok
    return torch.cat((ident, chain_fn(rel, scale))), rel, ok


class SequenceTracker:
    """GPU driver of track_sharded on top of PoseEstimator: every rank walks its block of frames in chunks of ``chunk`` frames
    (PoseEstimator.forward_chunk: one RAFT pass over the 2 * chunk pairs of a chunk, one fused geometry pass, one n-row solve),
    starting from the halo frame.  The chunked walk is bit-identical to the reference's frame-at-a-time walk (``chunk=1``:
    core/pose/pose_estimator.py:98-125, one ``forward`` per frame); the chunk size only trades memory (a chunk of 16 640x512 frames
    = the RAFT batch 32 of bench.py) for throughput.  The estimator is built ONCE per tracker (model build / weight load / GPU
    allocation) and reset between blocks."""

    def __init__(self, make_estimator, get_frame, flow2depth=None, chain=None, chunk=16):
        """make_estimator() -> a PoseEstimator on this rank's GPU (called once, lazily); get_frame(t) -> (limg, rimg, mask).
        ``flow2depth(flow, baseline) -> (depth, valid)`` and ``chain(rel, scale) -> poses`` default to the HIP entry points
        (rpe_flow2depth, rpe_se3_chain); the CPU tests inject the oracle's so that the same driver runs without a GPU."""
        if chunk < 1:
            raise ValueError('SequenceTracker: chunk must be >= 1')
        self.make_estimator, self.get_frame = make_estimator, get_frame
        self._flow2depth, self._chain = flow2depth, chain
        self.chunk = int(chunk)
        self._est = None

    @property
    def estimator(self):
        if self._est is None:
            self._est = self.make_estimator()
        return self._est

    def run_block(self, first_pair, last_pair):
        est = self.estimator
        est.reset()
        rels, oks = [], []
        if last_pair <= first_pair:
            dev = est.device
            return torch.zeros(0, 7, device=dev), torch.zeros(0, dtype=torch.bool, device=dev)
        flow2depth = self._flow2depth
        if flow2depth is None:
            from . import ops
            flow2depth = ops.flow2depth
        l, r, m = self.get_frame(first_pair)
        est(l, r, m)                                   # halo / first frame: stereo depth only
        if first_pair > 0:
            # in the serial run this frame was the "current" frame of pair first_pair-1, whose infer() ANDed its
            # mask with the stereo validity (pose_net.py:77); reproduce that for the halo frame
            _, valid = flow2depth(est.frame.flow, est.baseline * est.scale)
            est.frame.mask &= valid
        t = first_pair
        while t < last_pair:
            k = min(self.chunk, last_pair - t)
            if k == 1:
                l, r, m = self.get_frame(t + 1)
                est(l, r, m)
                rels.append(est.last_rel_pose.data.reshape(1, 7).float())
                oks.append(torch.tensor([est.success], device=rels[-1].device))
            else:
                fr = [self.get_frame(t + 1 + i) for i in range(k)]
                est.forward_chunk(torch.cat([f[0] for f in fr]), torch.cat([f[1] for f in fr]), torch.cat([f[2] for f in fr]))
                rels.append(est.last_rel_poses.reshape(k, 7).float())
                oks.append(est.successes.reshape(k))
            t += k
        return torch.cat(rels), torch.cat(oks)

    def track(self, n_frames, rank=0, world=1, group=None, scale=None):
        """``scale`` defaults to the depth-clipping distance of the estimator this tracker builds."""
        est_scale = float(self.estimator.config['depth_clipping'][1])     # the distance PoseEstimator normalises by (pose_estimator.py:40-43)
        if scale is None:
            scale = est_scale
        elif abs(scale - est_scale) > 1e-6 * est_scale:
            raise ValueError(f'SequenceTracker.track: scale {scale} != the estimator\'s depth_clipping {est_scale}')
        chain = self._chain
        if chain is None:
            from . import ops
            chain = lambda rel, s: ops.se3_chain(rel, scale=s)
        return track_sharded(n_frames, self.run_block, chain, rank, world, group, scale)
