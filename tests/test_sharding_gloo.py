"""CPU, world_size 2, gloo: the multi-GPU host logic (block partition with halo, all-gather of relative poses,
failure gate, SE(3) prefix product) gives the serial trajectory.  The per-pair solve is a deterministic
function of the pair index here (oracle SE3 ops), so no GPU is needed."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _rel_pose(t):
    """Deterministic 'solved' relative pose of pair t (frames t, t+1); pair 5 fails the gate, pair 9 is NaN."""
    from oracle import se3
    g = torch.Generator().manual_seed(1000 + t)
    xi = torch.randn(1, 6, generator=g) * 0.02
    if t == 5:
        xi[0, 0] = 0.5
    T = se3.se3_exp(xi)
    if t == 9:
        T[0, 1] = float('nan')
    return T


def _run_block(s, e):
    from oracle import se3
    import rpe_amd.sharding as sh
    if e <= s:
        return torch.zeros(0, 7), torch.zeros(0, dtype=torch.bool)
    rel = torch.cat([_rel_pose(t) for t in range(s, e)])
    return sh.failure_gate(rel, se3.se3_log(rel))


def _chain(rel, scale):
    from oracle import tracker
    return tracker.chain(rel, scale)


def _worker(rank, world, port, n_frames, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import rpe_amd.sharding as sh
    poses, rel, ok = sh.track_sharded(n_frames, _run_block, _chain, rank, world, scale=250.0)
    out[rank] = (poses, rel, ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_frames', [14, 2, 1])
def test_two_ranks_reproduce_serial_trajectory(n_frames):
    import rpe_amd.sharding as sh
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_frames, out), nprocs=world, join=True)
    serial, rel_s, ok_s = sh.track_sharded(n_frames, _run_block, _chain, 0, 1, scale=250.0)
    for r in range(world):
        poses, rel, ok = out[r]
        assert poses.shape == (n_frames, 7)
        assert torch.equal(poses, serial) and torch.equal(ok, ok_s)
        assert torch.equal(rel, rel_s)
    if n_frames == 14:
        assert ok_s.tolist() == [t not in (5, 9) for t in range(13)]
        assert torch.equal(serial[0], torch.tensor([0, 0, 0, 0, 0, 0, 1.0]))


def test_block_partition():
    import rpe_amd.sharding as sh
    assert sh.block_partition(13, 2) == [(0, 7), (7, 13)]
    assert sh.block_partition(3, 8) == [(0, 1), (1, 2), (2, 3)] + [(3, 3)] * 5
    for n in (0, 1, 7, 100):
        for w in (1, 2, 8):
            b = sh.block_partition(n, w)
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))


def _worker_single(rank, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=0, world_size=1)
    import rpe_amd.sharding as sh
    calls = []
    real = dist.all_gather
    dist.all_gather = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    poses, rel, ok = sh.track_sharded(6, _run_block, _chain, 0, 1, scale=250.0)
    out['calls'] = len(calls)
    out['poses'] = poses
    dist.destroy_process_group()


def test_world_size_one_group_still_runs_the_collective():
    """With a process group of ONE rank the padded all-gather is executed (no world==1 shortcut): this is the path
    `bench.py --mode sequence --gpus 1` and the nccl GPU test exercise under RCCL."""
    import rpe_amd.sharding as sh
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_single, args=(_free_port(), out), nprocs=1, join=True)
    assert out['calls'] == 1
    serial, _, _ = sh.track_sharded(6, _run_block, _chain, 0, 1, scale=250.0)      # no process group: plain serial
    assert torch.equal(out['poses'], serial)


def test_scale_is_required():
    import rpe_amd.sharding as sh
    with pytest.raises(ValueError):
        sh.track_sharded(3, _run_block, _chain, 0, 1)


def test_bench_refuses_mismatched_world_size():
    """bench.py --gpus N must never silently run one process: with WORLD_SIZE set by a launcher it has to agree with
    --gpus, and without a launcher it starts the ranks itself (here: no GPU -> it must say so and exit non-zero)."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], env=env, capture_output=True, text=True)
    assert r.returncode == 2 and 'WORLD_SIZE=1' in r.stderr
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '64'], env=env, capture_output=True, text=True)
    assert r.returncode == 2 and 'GPU(s) are visible' in r.stderr


# ------------------------------------------------------------------------------------ SequenceTracker itself, world size 4
class _FrameProxy:
    def __init__(self, d):
        object.__setattr__(self, '_d', d)

    def __getattr__(self, k):
        return self._d[k]

    def __setattr__(self, k, v):
        self._d[k] = v


class _OracleEstimator:
    """oracle.tracker.PoseEstimator behind the interface SequenceTracker drives (the product PoseEstimator's): callable per
    frame, ``reset``, ``frame.flow / .mask``, ``baseline``, ``scale``, ``last_rel_pose.data``, ``success``, ``config``, ``device``."""

    def __init__(self, model, K, bf):
        self._args = (model, K, bf)
        self.config = {'depth_clipping': [1, 250.0]}
        self.device = torch.device('cpu')
        self.reset()

    def reset(self):
        from oracle import tracker
        self._t = tracker.PoseEstimator(*self._args)
        self.baseline, self.scale = self._t.baseline, self._t.scale
        return self

    def __call__(self, l, r, m):
        from types import SimpleNamespace
        self._t.forward(l, r, m)
        self.success = self._t.success[-1]
        self.last_rel_pose = SimpleNamespace(data=self._t.rel_poses[-1])

    def forward_chunk(self, L, R, M):
        """The chunk interface of the product estimator with the oracle's semantics: the c single calls, one after the other (what
        PoseEstimator.forward_chunk must reproduce bit for bit on the GPU; tests/test_gpu_chunked_tracker.py)."""
        rels, oks = [], []
        for i in range(L.shape[0]):
            self(L[i:i + 1], R[i:i + 1], M[i:i + 1])
            rels.append(self.last_rel_pose.data.reshape(1, 7))
            oks.append(self.success)
        self.last_rel_poses, self.successes = torch.cat(rels), torch.tensor(oks)
        self.chunk_calls = getattr(self, 'chunk_calls', 0) + 1

    @property
    def frame(self):
        return _FrameProxy(self._t.frame)


def _oracle_tracker(n_frames, chunk=2):
    import rpe_amd.sharding as sh
    from oracle import pose_net as opn
    from oracle import tracker, warp
    from rpe_amd import synth
    h, w = 128, 160
    cfg = synth.model_config(h, w, iters=3, lbgfs_iters=4, use_weights=False)
    torch.manual_seed(0)
    model = synth.init_synthetic_weights(opn.PoseNet(cfg)).eval()
    fr = synth.stereo_frames(123, n_frames, h, w)
    get = lambda t: (fr['image2l'][t:t + 1], fr['image2r'][t:t + 1], fr['mask2'][t:t + 1].clone())
    make = lambda: _OracleEstimator(model, fr['K'][0], 7.2 * 250.0)
    return sh.SequenceTracker(make, get, flow2depth=warp.flow2depth, chain=tracker.chain, chunk=chunk)


def _worker_seq(rank, world, port, n_frames, out):
    import sys
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1 if world > 4 else 2)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    tr = _oracle_tracker(n_frames)
    built = []
    real = tr.make_estimator
    tr.make_estimator = lambda: (built.append(1), real())[1]
    poses, rel, ok = tr.track(n_frames, rank, world)
    tr.run_block(0, 0)                                        # a second block on the same tracker: no second estimator
    out[rank] = (poses, rel, ok, len(built))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_frames', [4, 6])
def test_sequence_tracker_four_ranks_uneven_blocks(n_frames):
    """SequenceTracker (the driver bench.py --mode sequence runs) over four gloo ranks with the ORACLE as the per-pair solver:
    4 frames = 3 pairs (one rank has no pair at all), 6 frames = 5 pairs (blocks of 2, 1, 1, 1).  Every rank ends with the serial
    trajectory; the halo handling (stereo validity ANDed into the halo frame's mask) makes the relative poses those of the
    serial run; the estimator is built once per rank."""
    world = 4
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_seq, args=(world, _free_port(), n_frames, out), nprocs=world, join=True)
    torch.set_num_threads(2)
    serial, rel_s, ok_s = _oracle_tracker(n_frames, chunk=1).track(n_frames)       # the frame-at-a-time walk of the reference
    assert serial.shape == (n_frames, 7) and bool(torch.isfinite(serial).all())
    for r in range(world):
        poses, rel, ok, built = out[r]
        assert built == 1
        assert torch.equal(ok, ok_s) and float((rel - rel_s).abs().max()) <= 1e-6
        assert float((poses - serial).abs().max()) <= 1e-3 * max(1.0, float(serial.abs().max()))
        assert torch.equal(poses, out[0][0])                  # bitwise the same trajectory on every rank


def test_chunked_walk_is_the_frame_at_a_time_walk_on_the_oracle():
    """The driver's chunking itself (chunk boundaries, a trailing single frame, the halo frame in front) with the oracle as the
    estimator: chunk = 3 over 8 frames = 7 pairs walks (halo) 3 + 3 + 1 and must hand back exactly what chunk = 1 does."""
    torch.set_num_threads(4)
    one = _oracle_tracker(8, chunk=1)
    three = _oracle_tracker(8, chunk=3)
    r1, ok1 = one.run_block(0, 7)
    r3, ok3 = three.run_block(0, 7)
    assert torch.equal(r1, r3) and torch.equal(ok1, ok3) and three.estimator.chunk_calls == 2
    a, oka = three.run_block(2, 6)                            # a block in the middle: halo frame 2, chunks of 3 + 1
    assert float((a - r1[2:6]).abs().max()) <= 1e-6 and torch.equal(oka, ok1[2:6])
    with pytest.raises(ValueError):
        _oracle_tracker(3, chunk=0)


def test_sequence_tracker_eight_ranks_chunked():
    """World size 8 (one node), 14 frames = 13 pairs -> blocks of 2,2,2,2,2,1,1,1: the five two-pair blocks run through
    forward_chunk, the three single-pair ones through forward; all ranks end with the serial trajectory bit for bit."""
    import rpe_amd.sharding as sh
    world, n_frames = 8, 14
    assert [e - s for s, e in sh.block_partition(n_frames - 1, world)] == [2, 2, 2, 2, 2, 1, 1, 1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_seq, args=(world, _free_port(), n_frames, out), nprocs=world, join=True)
    torch.set_num_threads(4)
    serial, rel_s, ok_s = _oracle_tracker(n_frames, chunk=1).track(n_frames)
    for r in range(world):
        poses, rel, ok, built = out[r]
        assert built == 1 and torch.equal(ok, ok_s) and float((rel - rel_s).abs().max()) <= 1e-6
        assert float((poses - serial).abs().max()) <= 1e-3 * max(1.0, float(serial.abs().max()))
        assert torch.equal(poses, out[0][0])


def _worker_bench_pass(rank, world, port, n_frames, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bench
    import rpe_amd.sharding as sh
    track = lambda: sh.track_sharded(n_frames, _run_block, _chain, rank, world, scale=250.0)
    sp = bench.sequence_pass(track, n_frames, rank, steps=2, warmup=1, dev='cpu', dist=dist)
    out[rank] = (sp['poses'], sp['ok'], sp['same'], sp['frames_per_s'], sp['checksum'])
    dist.barrier()
    dist.destroy_process_group()


def test_bench_sequence_pass_on_two_ranks():
    """bench.py's sequence pass (what ``--mode sequence`` times and what ``--gpus N`` adds to the batch line as sequence_frames_per_s when
    N > 1): K timed walks, the trajectory checksum compared across ranks with MIN / MAX all-reduces.  Driven here on the CPU oracle
    under gloo, world size 2: both ranks report the serial trajectory, the same checksum, and a positive rate."""
    import rpe_amd.sharding as sh
    world, n_frames = 2, 14
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_bench_pass, args=(world, _free_port(), n_frames, out), nprocs=world, join=True)
    serial, _, ok_s = sh.track_sharded(n_frames, _run_block, _chain, 0, 1, scale=250.0)
    for r in range(world):
        poses, ok, same, fps, chk = out[r]
        assert same and fps > 0.0
        assert torch.equal(poses, serial) and torch.equal(ok, ok_s)
    assert out[0][4] == out[1][4]
