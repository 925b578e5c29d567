"""GPU: the HIP pose layer (rpe_pose_reduce / rpe_pose_solve / rpe_se3_*) through the C ABI, against the
golden vectors of the reference and against the CPU oracle on seeded inputs.

Tolerances: the arithmetic is float64 on both sides; differences come from summation order only.
  objective / gradient : 1e-11 relative
  L-BFGS iterates      : 1e-8 absolute on the 7-vector (north-star bar: 1e-4)
"""
import pytest
import torch

from conftest import SOLVER_KEYS, load_golden
from oracle import pose_head, se3, synth

pytestmark = pytest.mark.gpu


def dev(args):
    return [a.cuda() for a in args]


@pytest.mark.parametrize('name', ['solver_a', 'solver_b', 'solver_c', 'solver_d'])
def test_reduce_matches_reference_golden(rpe, name):
    from rpe_amd import ops
    g = load_golden(name + '.npz')
    args = dev([g[k] for k in SOLVER_KEYS])
    for tag in ('id', 'rnd'):
        out = ops.pose_reduce(*args, g['T_' + tag].cuda(), need_hessian=True)
        f, gr = out['f'].cpu(), out['g'].cpu()
        scale = max(1.0, float(g['graw_' + tag].abs().max()))
        assert torch.allclose(f, g['f_' + tag], rtol=1e-11, atol=1e-300), (name, tag)
        assert float((gr - g['graw_' + tag]).abs().max()) <= 1e-11 * scale, (name, tag)
        P = pose_head._prep(*[g[k] for k in SOLVER_KEYS])
        H = pose_head.evaluate(P, g['T_' + tag], need_hessian=True)['H']
        assert float((out['H'].cpu() - H).abs().max()) <= 1e-10 * max(1.0, float(H.abs().max())), (name, tag)


@pytest.mark.parametrize('name,ks', [('solver_a', (1, 2, 3, 8, 20, 100)), ('solver_c', (1, 3, 8, 20, 100)),
                                     ('solver_d', (1,))])
def test_lbfgs_iterates_match_reference_golden(rpe, name, ks):
    """n == 1 rows of the reference (its inference configuration), iterate for iterate."""
    from rpe_amd import ops
    g = load_golden(name + '.npz')
    args = [g[k] for k in SOLVER_KEYS]
    n = args[0].shape[0]
    for k in ks:
        T, v7, l6, info = ops.pose_solve(*dev(args), iters=k, mode=ops.SOLVER_LBFGS)
        ref = g['T_k%d' % k] if n == 1 else g.get('Tind_k%d' % k)
        if ref is None:
            ref, _ = pose_head.lbfgs_solve(*args, iters=k, coupled=False)
        assert float((T.cpu() - ref).abs().max()) < 1e-8, (name, k, info.cpu())
        if n == 1:
            assert torch.allclose(v7.cpu(), g['vec7_k%d' % k].reshape(n, 7), atol=1e-6)
            assert torch.allclose(l6.cpu(), g['log6_k%d' % k].reshape(n, 6), atol=1e-6)


def test_batched_rows_are_independent_solves(rpe):
    from rpe_amd import ops
    g = load_golden('solver_b.npz')
    args = [g[k] for k in SOLVER_KEYS]
    for k in (3, 8, 20):
        T, _, _, info = ops.pose_solve(*dev(args), iters=k)
        assert float((T.cpu() - g['Tind_k%d' % k]).abs().max()) < 1e-8
        To, io = pose_head.lbfgs_solve(*args, iters=k, coupled=False)
        assert info[:, 0].cpu().tolist() == io['n_iter'].tolist()
        assert info[:, 2].cpu().tolist() == io['stop'].tolist()


def test_stop_reasons_and_counts_match_oracle(rpe):
    from rpe_amd import ops
    for seed in (3, 4):
        c = synth.solver_case(seed, 2, 64, 96)
        args = synth.solver_args(c)
        for k in (0, 1, 2, 5, 40):
            T, _, _, info = ops.pose_solve(*dev(args), iters=k)
            To, io = pose_head.lbfgs_solve(*args, iters=k)
            assert float((T.cpu() - To).abs().max()) < 1e-8
            assert info[:, 0].cpu().tolist() == io['n_iter'].tolist(), k
            assert info[:, 1].cpu().tolist() == io['evals'].tolist(), k
            assert info[:, 2].cpu().tolist() == [s if s else 3 for s in io['stop'].tolist()], k


def test_gauss_newton_matches_oracle(rpe):
    from rpe_amd import ops
    c = synth.solver_case(7, 3, 64, 80)
    args = synth.solver_args(c)
    for k in (1, 2, 8):
        T, _, _, info = ops.pose_solve(*dev(args), iters=k, mode=ops.SOLVER_GN)
        To, io = pose_head.gn_solve(*args, iters=k)
        assert float((T.cpu() - To).abs().max()) < 1e-9
        assert info[:, 2].cpu().tolist() == io['stop'].tolist()
    # converged GN sits within 1e-4 of the ground-truth pose (noise-limited)
    err = (se3.se3_log(T.cpu()) - c['xi_gt']).abs().max()
    assert float(err) < 5e-3


def test_all_masked_rows_stop_at_start(rpe):
    from rpe_amd import ops
    c = synth.solver_case(5, 2, 32, 48)
    c['mask1'][1] = False
    args = synth.solver_args(c)
    T, v7, l6, info = ops.pose_solve(*dev(args), iters=8)
    assert info[1, 2].item() == 1 and torch.equal(T[1].cpu(), torch.tensor([0, 0, 0, 0, 0, 0, 1.0], dtype=torch.float64))
    To, _ = pose_head.lbfgs_solve(*args, iters=8)
    assert float((T.cpu() - To).abs().max()) < 1e-8


def test_nan_poisons_pose_like_reference(rpe):
    from rpe_amd import ops
    g = load_golden('solver_nan.npz')
    args = dev([g[k] for k in SOLVER_KEYS])
    out = ops.pose_reduce(*args, torch.tensor([[0, 0, 0, 0, 0, 0, 1.0]], dtype=torch.float64).cuda())
    assert torch.allclose(out['f'].cpu(), g['f_id'], rtol=1e-11)
    assert bool(torch.isnan(out['g']).all())
    T, v7, _, _ = ops.pose_solve(*args, iters=3)
    assert bool(torch.isnan(T).all()) and bool(torch.isnan(v7).all())


def test_odd_sizes_take_the_scalar_path(rpe):
    from rpe_amd import ops
    c = synth.solver_case(9, 2, 37, 53, outliers=False)
    args = synth.solver_args(c)
    T, _, _, _ = ops.pose_solve(*dev(args), iters=8)
    To, _ = pose_head.lbfgs_solve(*args, iters=8)
    assert float((T.cpu() - To).abs().max()) < 1e-8


def test_reference_known_answer_test_on_hip(rpe):
    """tests/unit_test_pose_head.py:38-50 thresholds, per-frame solves, 180x180, 100 iterations."""
    from rpe_amd import ops
    c = synth.solver_case(12345, 5, 180, 180, sigma_t=0.01, sigma_r=0.01, noise=0.0, unit_weights=True,
                          full_masks=True, outliers=False)
    c['loss_weight'] = torch.tensor([[0.001, 1.0]]).repeat(5, 1)
    args = synth.solver_args(c)
    T, _, l6, _ = ops.pose_solve(*dev(args), iters=100)
    assert float(pose_head.objective(*args, T.cpu()).max()) <= 1e-5
    sup = (l6.cpu().double() - c['xi_gt']).abs().sum() / 5
    assert float(sup) <= 0.05


def test_full_size_properties(rpe):
    """640x512, batch 4: (a) gradient at the noise-free ground truth is ~0; (b) the solve recovers the pose;
    (c) a row permutation of the batch permutes the result (rows independent, bit for bit)."""
    from rpe_amd import ops
    c = synth.solver_case(21, 4, 512, 640, noise=0.0, outliers=False, full_masks=True)
    args = synth.solver_args(c)
    d = dev(args)
    Tgt = se3.se3_exp(c['xi_gt'])
    out = ops.pose_reduce(*d, Tgt.cuda())
    assert float(out['g'].abs().max()) < 1e-6
    T, _, l6, info = ops.pose_solve(*d, iters=20)
    assert float((l6.cpu().double() - c['xi_gt']).abs().max()) < 2e-3
    perm = [2, 0, 3, 1]
    Tp, _, _, _ = ops.pose_solve(*[a[perm].contiguous() for a in d], iters=20)
    assert torch.equal(Tp, T[perm])


def test_se3_kernels_match_oracle(rpe):
    from rpe_amd import ops
    torch.manual_seed(0)
    for dt, tol in ((torch.float64, 1e-13), (torch.float32, 2e-6)):
        xi = torch.cat((torch.randn(200, 6, dtype=dt), torch.randn(56, 6, dtype=dt) * 1e-4))     # incl. Taylor branch
        T = ops.se3_exp(xi.cuda()).cpu()
        assert torch.allclose(T, se3.se3_exp(xi), atol=tol)
        assert torch.allclose(ops.se3_log(T.cuda()).cpu(), se3.se3_log(T), atol=tol * 10)
        assert torch.allclose(ops.se3_inv(T.cuda()).cpu(), se3.se3_inv(T), atol=tol * 10)
        B = se3.se3_exp(torch.randn(256, 6, dtype=dt))
        assert torch.allclose(ops.se3_mul(T.cuda(), B.cuda()).cpu(), se3.se3_mul(T, B), atol=tol * 10)
        pts = torch.randn(256, 33, 3, dtype=dt)
        assert torch.allclose(ops.se3_act(T.cuda(), pts.cuda()).cpu(), se3.se3_act(T[:, None], pts), atol=tol * 10)
    # reference tolerances (tests/unit_test_pinhole_transforms.py:24-33)
    pcl = torch.clamp(torch.rand(20, 900, 3), 0.0001, 1)
    T = se3.se3_exp(torch.randn(20, 6))
    fwd = ops.se3_act(T.cuda(), pcl.cuda())
    back = ops.se3_act(ops.se3_inv(T.cuda()), fwd).cpu()
    assert torch.allclose(back, pcl, rtol=1e-3, atol=1e-6)


def test_chain_matches_tracker_recurrence(rpe):
    from rpe_amd import ops
    from oracle import tracker
    torch.manual_seed(1)
    rel = se3.se3_exp(torch.randn(300, 6, dtype=torch.float64) * 0.01)
    out = ops.se3_chain(rel.cuda(), scale=250.0).cpu()
    assert torch.allclose(out, tracker.chain(rel, 250.0), atol=1e-9)


def test_tartanair_ground_truth_pose_recovered_on_hip(rpe):
    """Real data with ground truth (the reference's TartanAir fixture): HIP solve == oracle solve, both == GT pose."""
    from rpe_amd import ops
    from oracle import warp
    g = load_golden('tartanair_crop.npz')
    pcl1 = warp.backproject(g['depth0'], g['K'])
    ones = torch.ones_like(g['depth0'])
    args = (g['flow'], pcl1, g['pcl2w'], ones, ones, g['valid'], torch.ones_like(g['valid']), g['K'], torch.ones(1, 2))
    for mode, solve in ((ops.SOLVER_LBFGS, pose_head.lbfgs_solve), (ops.SOLVER_GN, pose_head.gn_solve)):
        T, _, _, info = ops.pose_solve(*dev(args), iters=20, mode=mode)
        To, _ = solve(*args, iters=20)
        assert float((T.cpu() - To).abs().max()) < 1e-8
        assert float((se3.se3_matrix(T.cpu())[0] - g['rel_matrix']).abs().max()) < 2e-3


def test_history_shift_and_tolerances(rpe):
    """torch.optim.LBFGS keeps at most history_size (y, s) pairs and drops the oldest; with the stopping tests
    disabled (tolerances 0) and a short history the shift runs many times.  Also the default-tolerance run."""
    from rpe_amd import ops
    c = synth.solver_case(31, 2, 24, 32, noise=0.3, sigma_t=0.05, sigma_r=0.05)
    args = synth.solver_args(c)
    for hist in (3, 100):
        To, io = pose_head.lbfgs_solve(*args, iters=30, tolerance_change=0.0, tolerance_grad=0.0, history_size=hist)
        T, _, _, info = ops.pose_solve(*dev(args), iters=30, tolerance_change=0.0, tolerance_grad=0.0, history_size=hist)
        assert info[:, 0].cpu().tolist() == io['n_iter'].tolist() == [30, 30]
        assert float((T.cpu() - To).abs().max()) < 1e-6, hist
    Td, iod = pose_head.lbfgs_solve(*args, iters=140)
    T, _, _, info = ops.pose_solve(*dev(args), iters=140)
    assert info[:, 0].cpu().tolist() == iod['n_iter'].tolist() and info[:, 2].cpu().tolist() == iod['stop'].tolist()
    assert float((T.cpu() - Td).abs().max()) < 1e-7


def test_solve_is_graph_capturable(rpe):
    """A whole solve is 2N + 2 launches on one stream with no host synchronisation (the reference synchronises every iteration
    through float(loss)): it can be captured into a HIP graph and replayed on new inputs in the same buffers."""
    from rpe_amd import ops
    c = synth.solver_case(8, 2, 64, 80)
    args = [a.clone() for a in dev(synth.solver_args(c))]
    eager = ops.pose_solve(*args, iters=8)[0].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.pose_solve(*args, iters=8)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        T, vec7, log6, info = ops.pose_solve(*args, iters=8)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(T, eager)
    c2 = synth.solver_case(9, 2, 64, 80)
    for dst, src in zip(args, dev(synth.solver_args(c2))):
        dst.copy_(src)
    graph.replay()
    torch.cuda.synchronize()
    want = ops.pose_solve(*dev(synth.solver_args(c2)), iters=8)[0]
    assert torch.equal(T, want) and info[:, 0].tolist() == [8, 8]


@pytest.mark.parametrize('mode', [0, 1])
def test_one_launch_solve_equals_a_launch_per_evaluation(rpe, mode):
    """The whole solve as ONE persistent launch (a row's workgroups wait for the row's tail between evaluations) against one launch per
    evaluation (RPE_SOLVE_LAUNCH_PER_EVALUATION): the same block partition, the same fixed-order sums, so every output -- poses, float32
    views, iteration / evaluation counts, stop reasons -- must be equal bit for bit.  Rows that stop early (masked out, NaN, converged)
    sit next to rows that run all iterations; sizes from one workgroup per row to the full 640x512 frame; repeated (stale tickets /
    epochs of a previous solve in a reused workspace would show)."""
    from rpe_amd import ops
    cases = [synth.solver_args(synth.solver_case(11, 5, 64, 96)), synth.solver_args(synth.solver_case(12, 2, 512, 640)),
             synth.solver_args(synth.solver_case(13, 16, 128, 160)), synth.solver_args(synth.solver_case(14, 1, 33, 57))]
    a0 = [t.clone() for t in cases[0]]
    a0[5][1] = False                                    # row 1: mask1 all false -> optimal at start
    a0[0][3, 0, 5, 5] = float('nan')                    # row 3: a NaN flow -> NaN pose, runs to the evaluation limit like the reference
    cases.append(a0)
    for args in cases:
        for k in (1, 2, 8, 20):
            a = rpe_solve(ops, args, k, mode, True)
            for _ in range(2):
                b = rpe_solve(ops, args, k, mode, False)
                c = rpe_solve(ops, args, k, mode, True)
                for x, y, z in zip(a, b, c):
                    assert torch.equal(x, y, ) or (torch.isnan(x) == torch.isnan(y)).all() and torch.equal(torch.nan_to_num(x), torch.nan_to_num(y))
                    assert torch.equal(torch.nan_to_num(x), torch.nan_to_num(z)) and (torch.isnan(x) == torch.isnan(z)).all()


def rpe_solve(ops, args, k, mode, persistent):
    return [t.cpu() for t in ops.pose_solve(*dev(args), iters=k, mode=mode, persistent=persistent)]
