"""GPU: the chunked block walk of BASELINE config 4 (sharding.SequenceTracker(chunk=c) -> PoseEstimator.forward_chunk ->
PoseNet.infer_chunk: ONE RAFT pass over a chunk's 2c pairs, one fused geometry pass with depth1 / mask1 / stereo_flow1 shifted by
one row inside the batch, one c-row solve) against the frame-at-a-time walk the reference performs
(core/pose/pose_estimator.py:98-125, core/pose/pose_net.py:63-79, one ``forward`` per frame) -- BIT-IDENTICAL: relative poses,
gate decisions, chained poses, and the Frame (depth, mask, stereo flow) left behind.

That rests on every kernel computing a row independently of the batch it is launched in (tile classes that leave the same
products in the same order and the same statistics records; rpe_solve_opts.partition_rows = 1 for the solve), which the
kernel-level tests check one by one; here the whole path is compared with torch.equal."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu
H, W = 352, 384
CFG = dict(frame2frame=True, depth_clipping=[1, 250], lbgfs_iters=8, conf_weighing=True)


@pytest.fixture(scope='module')
def seq(rpe):
    from rpe_amd import pose_net, synth
    cfg = synth.model_config(H, W, iters=12, lbgfs_iters=8)
    model = synth.init_synthetic_weights(pose_net.PoseNet(cfg)).eval().cuda()
    fr = synth.stereo_frames(21, 10, H, W)                     # the ten "second views" as a 10-frame stereo sequence
    L, R, M = fr['image2l'].clone(), fr['image2r'].clone(), fr['mask2'].clone()
    L[5, :, 100:116, 200:232] = float('nan')                   # frame 5 is broken: pairs 4 and 5 come back NaN and must be gated
    return model, fr['K'][0], (L.cuda(), R.cuda(), M.cuda())


def _tracker(seq, chunk):
    from rpe_amd import pose_estimator, sharding
    model, K, (L, R, M) = seq
    make = lambda: pose_estimator.PoseEstimator(CFG, K, 7.2 * 250.0, model, (W, H)).cuda()
    get = lambda t: (L[t:t + 1], R[t:t + 1], M[t:t + 1].clone())
    return sharding.SequenceTracker(make, get, chunk=chunk)


def _walk(seq, chunk, first, last):
    tr = _tracker(seq, chunk)
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter('always')
        rel, ok = tr.run_block(first, last)
    est = tr.estimator
    n_warn = len([w for w in wl if issubclass(w.category, RuntimeWarning)])
    return dict(rel=rel, ok=ok, depth=est.frame.depth.clone(), mask=est.frame.mask.clone(), flow=est.frame.flow.clone(),
                pose=est.last_pose.data.clone(), warnings=n_warn)


def test_prefetched_encoders_are_bit_identical_to_frame_at_a_time(seq):
    """PoseEstimator.submit / result: frame t+1's encoders run on a side stream while frame t's update loop is on the GPU (the reference's
    frame-at-a-time deployment, scripts/infer_trajectory.py:57,71-77, with the next frame read one iteration early).  Same kernels on the
    same inputs: every absolute pose, gate decision, relative pose and the Frame left behind equal the plain forward() walk bit for
    bit -- on the ten-frame sequence with the broken frame 5 (its NaNs reach pairs 4 and 5, both gated)."""
    from rpe_amd import pose_estimator
    model, K, (L, R, M) = seq
    make = lambda: pose_estimator.PoseEstimator(CFG, K, 7.2 * 250.0, model, (W, H)).cuda()
    get = lambda t: (L[t:t + 1], R[t:t + 1], M[t:t + 1].clone())
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        plain = make()
        ref = []
        for t in range(10):
            P, _, _, _ = plain(*get(t))
            ref.append((P.data.clone(), plain.success, plain.last_rel_pose.data.clone() if t else None))
        piped = make()
        got = []
        piped.submit(*get(0))
        for t in range(10):
            if t + 1 < 10:
                piped.submit(*get(t + 1))                       # the next frame's encoders start before this frame's pose is asked for
            P, _, _, _ = piped.result()
            got.append((P.data.clone(), piped.success, piped.last_rel_pose.data.clone() if t else None))
    assert [g[1] for g in got] == [r[1] for r in ref] and [r[1] for r in ref][5:7] == [False, False]
    for t, (g, r) in enumerate(zip(got, ref)):
        assert torch.equal(g[0], r[0]), t
        assert t == 0 or torch.equal(g[2], r[2]), t
    for k in ('depth', 'mask', 'flow'):
        assert torch.equal(getattr(piped.frame, k), getattr(plain.frame, k)), k
    with pytest.raises(RuntimeError):
        piped.result()                                          # nothing submitted


def test_chunked_block_walk_is_bit_identical_to_frame_at_a_time(seq):
    base = _walk(seq, 1, 0, 9)
    assert base['ok'].tolist()[4:6] == [False, False] and bool(base['ok'][:4].all())     # both gated pairs, and converged ones
    assert base['warnings'] == int((~base['ok']).sum())
    assert bool(torch.isfinite(base['rel']).all())
    for chunk in (4, 3, 9, 16):                                # 4+4+1, 3+3+3, one chunk, a chunk larger than the block
        got = _walk(seq, chunk, 0, 9)
        for k in ('rel', 'ok', 'depth', 'mask', 'flow', 'pose'):
            assert torch.equal(got[k], base[k]), (chunk, k)
        assert got['warnings'] == base['warnings']


def test_chunked_block_with_a_halo_frame(seq):
    """A block in the middle of the sequence (what rank r > 0 walks): the halo frame's mask is ANDed with its stereo validity,
    then chunks of 4 + 2; bitwise what the frame-at-a-time walk of the same block gives, and within float32 round-off what the
    serial run from frame 0 gives for those pairs."""
    base = _walk(seq, 1, 3, 9)
    got = _walk(seq, 4, 3, 9)
    for k in ('rel', 'ok', 'depth', 'mask', 'flow'):
        assert torch.equal(got[k], base[k]), k
    serial = _walk(seq, 16, 0, 9)
    assert torch.equal(got['ok'], serial['ok'][3:]) and float((got['rel'] - serial['rel'][3:]).abs().max()) < 1e-5


def test_forward_chunk_returns_the_chained_poses_of_single_calls(seq):
    from rpe_amd import pose_estimator
    model, K, (L, R, M) = seq
    a = pose_estimator.PoseEstimator(CFG, K, 7.2 * 250.0, model, (W, H)).cuda()
    b = pose_estimator.PoseEstimator(CFG, K, 7.2 * 250.0, model, (W, H)).cuda()
    single = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for t in range(8):
            P, _, flow, weights = a(L[t:t + 1], R[t:t + 1], M[t:t + 1].clone())
            single.append(P.data.reshape(1, 7).clone())
        b(L[:1], R[:1], M[:1].clone())
        with pytest.raises(RuntimeError):
            pose_estimator.PoseEstimator(CFG, K, 7.2 * 250.0, model, (W, H)).cuda().forward_chunk(L[:2], R[:2], M[:2].clone())
        P1, _, _, _ = b.forward_chunk(L[1:4], R[1:4], M[1:4].clone())
        P2, _, flow_c, weights_c = b.forward_chunk(L[4:8], R[4:8], M[4:8].clone())
    assert torch.equal(torch.cat((P1, P2)), torch.cat(single[1:]))
    assert torch.equal(flow_c[-1:], flow) and torch.equal(weights_c[0][-1:], weights[0]) and torch.equal(weights_c[1][-1:], weights[1])
    assert a.success == b.success and torch.equal(a.last_rel_pose.data, b.last_rel_pose.data)
    assert torch.equal(a.frame.depth, b.frame.depth) and torch.equal(a.frame.mask, b.frame.mask) and torch.equal(a.last_frame.img, b.last_frame.img)


def test_forward_chunk_without_feature_reuse(seq):
    """reuse_features=False (re-encode the previous left image like the reference does): the chunk encodes image0l itself; same bits."""
    from rpe_amd import pose_estimator
    model, K, (L, R, M) = seq
    cfg = dict(CFG, reuse_features=False)
    a = pose_estimator.PoseEstimator(cfg, K, 7.2 * 250.0, model, (W, H)).cuda()
    b = pose_estimator.PoseEstimator(cfg, K, 7.2 * 250.0, model, (W, H)).cuda()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        single = [a(L[t:t + 1], R[t:t + 1], M[t:t + 1].clone())[0].data.reshape(1, 7).clone() for t in range(5)]
        b(L[:1], R[:1], M[:1].clone())
        P = b.forward_chunk(L[1:5], R[1:5], M[1:5].clone())[0]
    assert torch.equal(P, torch.cat(single[1:])) and torch.equal(a.frame.depth, b.frame.depth) and b._enc_cache is None


def test_track_sequence_in_chunks_writes_the_same_trajectory(seq):
    """trajectory.track_sequence (the loop of scripts/infer_trajectory.py:70-91) with chunk = 4: the same poses and stamps, bit for bit."""
    from rpe_amd import pose_estimator, trajectory
    model, K, (L, R, M) = seq
    out = {}
    for chunk in (1, 4):
        est = pose_estimator.PoseEstimator(CFG, K, 7.2 * 250.0, model, (W, H)).cuda()
        frames = ((L[t:t + 1], R[t:t + 1], M[t:t + 1].clone(), 100 + t) for t in range(10))
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            out[chunk] = trajectory.track_sequence(est, frames, start_stamp=99, chunk=chunk)
    assert [t['timestamp'] for t in out[4]] == [t['timestamp'] for t in out[1]] == [99] + list(range(100, 110))
    assert all(torch.equal(a['camera-pose'], b['camera-pose']) for a, b in zip(out[1], out[4]))


def test_solve_rows_do_not_depend_on_the_batch_with_partition_rows_1(rpe):
    """rpe_solve_opts.partition_rows = 1: a row's float64 sums are grouped as if it were solved alone."""
    from rpe_amd import ops, synth
    n, h, w = 6, 192, 256
    g = torch.Generator().manual_seed(5)
    flow = torch.randn(n, 2, h, w, generator=g)
    pcl1 = torch.rand(n, 3, h, w, generator=g) + torch.tensor([-0.5, -0.5, 0.5])[None, :, None, None]
    pcl2 = pcl1 + 0.01 * torch.randn(n, 3, h, w, generator=g)
    w1, w2 = torch.rand(n, 1, h, w, generator=g), torch.rand(n, 1, h, w, generator=g)
    m1, m2 = torch.rand(n, 1, h, w, generator=g) > 0.1, torch.rand(n, 1, h, w, generator=g) > 0.1
    K = synth.intrinsics(h, w)[None].repeat(n, 1, 1)
    args = [t.cuda() for t in (flow, pcl1, pcl2, w1, w2, m1, m2, K, torch.ones(n, 2))]
    for mode in (ops.SOLVER_LBFGS, ops.SOLVER_GN):
        T_all, v_all, l_all, i_all = ops.pose_solve(*args, iters=6, mode=mode, partition_rows=1)
        T_def = ops.pose_solve(*args, iters=6, mode=mode)[0]
        assert float((T_all - T_def).abs().max()) < 1e-9                # the default partition: same numbers to f64 round-off
        for i in range(n):
            Ti, vi, li, ii = ops.pose_solve(*[a[i:i + 1].contiguous() for a in args], iters=6, mode=mode)
            assert torch.equal(Ti, T_all[i:i + 1]) and torch.equal(vi, v_all[i:i + 1]) and torch.equal(ii, i_all[i:i + 1])


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_gate_chain_kernel_equals_the_step_by_step_bookkeeping(rpe, dtype):
    """rpe_pose_gate_chain (one launch per frame / chunk) against PoseEstimator.forward's bookkeeping written out with the SE3 operations
    it replaced (core/pose/pose_estimator.py:81-91): isnan | |log| > 0.1 -> identity, scale, inverse, chain.  Bit for bit, on rows just
    below / above the threshold in a translation and in a rotation component, a NaN row and ordinary rows."""
    from rpe_amd import ops
    from rpe_amd.se3 import SE3
    g = torch.Generator().manual_seed(5)
    xi = torch.randn(12, 6, generator=g, dtype=torch.float64) * torch.tensor([0.02, 0.02, 0.02, 0.01, 0.01, 0.01], dtype=torch.float64)
    xi[2, 0], xi[3, 0] = 0.0999, 0.1001
    xi[4, 4], xi[5, 4] = -0.0999, -0.1001
    xi[9] = 0.0
    rel = SE3.exp(xi.to(dtype).cuda()).data.clone()
    rel[7, 5] = float('nan')
    init = SE3.exp((torch.randn(1, 6, generator=g, dtype=torch.float64) * 0.3).to(dtype).cuda()).data
    s = float(1 / torch.tensor(1 / 250.0))
    got_rel, got_abs, ok = ops.pose_gate_chain(rel, init, s, 1.0e-1)
    pose, want_rel, want_abs, want_ok = SE3(init), [], [], []
    for k in range(rel.shape[0]):
        r = SE3(rel[k:k + 1])
        bad = bool(torch.isnan(r.vec()).any()) or bool((torch.abs(r.log()) > 1.0e-1).any())
        if bad:
            r = SE3.IdentityLike(pose)
        want_ok.append(0 if bad else 1)
        want_rel.append(r.data)
        pose = pose * r.scale(s).inv()
        want_abs.append(pose.data)
    assert ok.tolist() == want_ok and want_ok == [1, 1, 1, 0, 1, 0, 1, 0, 1, 1, 1, 1]
    assert torch.equal(got_rel, torch.cat(want_rel)) and torch.equal(got_abs, torch.cat(want_abs))
