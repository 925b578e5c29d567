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
    poses, rel, ok = sh.track_sharded(n_frames, _run_block, _chain, rank, world)
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
    serial, rel_s, ok_s = sh.track_sharded(n_frames, _run_block, _chain, 0, 1)
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
