"""GPU: backward of the declarative pose layer (rpe_pose_backward_moments / rpe_pose_backward_grads through the C ABI
and the autograd wiring of rpe_amd.pose_head.DeclarativeLayerLie) against the gradients the reference's own
DeclarativeNodeLie.gradient produced with autograd (tests/golden/backward_*.npz, oracle/gen_golden.py::gen_backward)
and against the closed-form CPU oracle on a larger seeded case.  float64 arithmetic on both sides (gradients are
returned as float32, the inputs' dtype): 1e-6 relative to each gradient's scale."""
import warnings

import pytest
import torch

from conftest import SOLVER_KEYS, load_golden
from oracle import pose_grad, pose_head, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['backward_a', 'backward_b'])
def test_moments_and_grads_match_reference_golden(rpe, name):
    from rpe_amd import ops
    g = load_golden(name + '.npz')
    args = [g[k].cuda() for k in SOLVER_KEYS]
    n = args[0].shape[0]
    T = g['vec7'].reshape(n, 7).double().cuda()
    g2u, g3u, H = ops.pose_backward_moments(*args, T)
    lw = g['loss_weight'].double()
    fY = lw[:, 1:2] * g2u.cpu() + lw[:, 0:1] * g3u.cpu()
    assert float((fY - g['fY_f64']).abs().max()) < 1e-8
    Href = 0.5 * (g['fYY_f64'] + g['fYY_f64'].transpose(1, 2))
    assert float((H.cpu() - Href).abs().max()) < 1e-7 * float(Href.abs().max())
    u = torch.linalg.solve(Href, -g['v'].reshape(n, 6, 1).double())[..., 0]
    got = ops.pose_backward_grads(*args, T, u.cuda(), ['flow', 'pcl1', 'pcl2', 'w1', 'w2'])
    for k, v in got.items():
        ref = g[f'g_{k}_f64']
        assert v.dtype == torch.float32 and v.shape == ref.shape
        assert float((v.cpu().double() - ref).abs().max()) < 1e-6 * float(ref.abs().max()), k
    only = ops.pose_backward_grads(*args, T, u.cuda(), ['w2'])
    assert list(only) == ['w2'] and torch.equal(only['w2'], got['w2'])


def _layer_grads(g, iters=100):
    from rpe_amd import pose_head as ph
    xs = []
    for i, k in enumerate(SOLVER_KEYS):
        t = g[k].cuda()
        if t.is_floating_point() and i in (0, 1, 2, 3, 4, 8):
            t.requires_grad_(True)
        xs.append(t)
    layer = ph.DeclarativeLayerLie(ph.DPoseSE3Head(None, lbgfs_iters=iters))
    vec7, log6 = layer(*xs)
    (log6 * g['v'].float().cuda()).sum().backward()                              # dL/dlog6 = v
    return xs, vec7, log6


def test_layer_autograd_matches_reference_golden(rpe):
    """The whole path a training step takes: layer forward (L-BFGS 100), a tangent-space loss, .backward() -- against the
    reference's own run of the same thing (n = 1: the reference's batched forward shares one L-BFGS history between rows)."""
    g = load_golden('backward_b.npz')
    xs, vec7, log6 = _layer_grads(g)
    assert vec7.shape == (1, 1, 7) and log6.shape == (1, 1, 6) and log6.requires_grad
    assert float((vec7.detach().cpu() - g['vec7']).abs().max()) < 1e-6          # same solution as the reference's forward
    for i, k in ((0, 'flow'), (1, 'pcl1'), (2, 'pcl2'), (3, 'w1'), (4, 'w2'), (8, 'loss_weight')):
        ref = g[f'g_{k}_f64']
        got = xs[i].grad
        assert got is not None and got.shape == ref.shape, k
        # the pose the backward is evaluated at is the GPU solve's float32 vec7 (~1e-7 from the reference's): 1e-4 relative
        assert float((got.cpu().double() - ref).abs().max()) < 1e-4 * float(ref.abs().max()), k
    assert xs[7].grad is None


def test_layer_autograd_batch_rows_match_oracle_at_their_own_pose(rpe):
    """n = 2 (independent rows on the GPU): gradients vs the closed-form oracle evaluated at the poses the GPU forward returned."""
    g = load_golden('backward_a.npz')
    xs, vec7, _ = _layer_grads(g)
    out, _, _ = pose_grad.layer_backward(*[g[k] for k in SOLVER_KEYS], vec7.detach().cpu(), g['v'])
    for i, k in ((0, 'flow'), (1, 'pcl1'), (2, 'pcl2'), (3, 'w1'), (4, 'w2'), (8, 'loss_weight')):
        assert float((xs[i].grad.cpu().double() - out[k]).abs().max()) < 1e-5 * float(out[k].abs().max()), k


def test_backward_larger_case_matches_oracle_and_is_linear_in_v(rpe):
    from rpe_amd import ops
    c = synth.solver_case(71, 2, 96, 128)
    c['loss_weight'] = torch.tensor([[1.0, 1.0], [0.3, 2.0]])
    args = synth.solver_args(c)
    To, _ = pose_head.lbfgs_solve(*args, iters=100, coupled=False)
    vec7 = To.float()
    v = torch.randn(2, 6, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    out, fY, fYY = pose_grad.layer_backward(*args, vec7, v)
    dargs = [a.cuda() for a in args]
    T = vec7.double().cuda()
    g2u, g3u, H = ops.pose_backward_moments(*dargs, T)
    Hs = 0.5 * (fYY + fYY.transpose(1, 2))
    assert float((H.cpu() - Hs).abs().max()) < 1e-9 * float(Hs.abs().max())
    u = torch.cholesky_solve(-v.reshape(2, 6, 1), torch.linalg.cholesky(Hs))[..., 0]
    got = ops.pose_backward_grads(*dargs, T, u.cuda(), ['flow', 'pcl1', 'pcl2', 'w1', 'w2'])
    for k, gv in got.items():
        assert float((gv.cpu().double() - out[k]).abs().max()) < 1e-6 * float(out[k].abs().max()), k
    twice = ops.pose_backward_grads(*dargs, T, (2 * u).cuda(), ['pcl1'])['pcl1']
    assert torch.equal(twice, 2 * got['pcl1'])                                   # exactly linear in u (power of two)


def test_backward_returns_zeros_when_solver_has_not_converged(rpe):
    """declerative_node_lie.py:43-47 (the reference's 'more error-handling'): |fY| > 1e-3 -> warning, zero gradients."""
    from rpe_amd import pose_head as ph
    c = synth.solver_case(72, 1, 48, 64, sigma_t=0.05, sigma_r=0.1)
    c['loss_weight'] = torch.tensor([[200.0, 200.0]])                            # large gradient at the 1-iteration iterate
    xs = [a.cuda() for a in synth.solver_args(c)]
    xs[3].requires_grad_(True)
    layer = ph.DeclarativeLayerLie(ph.DPoseSE3Head(None, lbgfs_iters=1))
    _, log6 = layer(*xs)
    with pytest.warns(UserWarning, match='Non-zero objective'):
        log6.sum().backward()
    assert float(xs[3].grad.abs().max()) == 0.0


def test_reference_backward_smoke(rpe):
    """tests/unit_test_pose_head.py:55-67 of the reference restated: gradient of a supervised tangent loss w.r.t. the loss weights."""
    from rpe_amd import pose_head as ph
    from oracle import se3
    c = synth.solver_case(12345, 5, 180, 180, sigma_t=0.01, sigma_r=0.01, noise=0.0, unit_weights=True, full_masks=True, outliers=False)
    lw = torch.nn.Parameter(torch.tensor([[0.01, 1.0]]).repeat(5, 1).cuda())
    c['loss_weight'] = lw
    xs = [a if a is lw else a.cuda() for a in synth.solver_args(c)]
    layer = ph.DeclarativeLayerLie(ph.DPoseSE3Head(None, lbgfs_iters=100))
    poses = layer(*xs)[1]
    target = se3.se3_log(se3.se3_exp(c['xi_gt'])).float().cuda()
    loss = (poses[:, 0] - target).abs().sum() / 5
    with warnings.catch_warnings():
        warnings.simplefilter('error')                                           # converged: no optimality warning
        grad_x, = torch.autograd.grad(loss, lw)
    assert grad_x.shape == (5, 2) and bool(torch.isfinite(grad_x).all())


def test_posenet_training_forward_and_backward(rpe):
    """The reference's training step (core/pose/pose_net.py:28-58 + scripts/train_posenet.py:97-136) with the flow network
    frozen: PoseNet.forward -> tangent pose -> L1 loss -> backward reaches both weight heads and loss_weight."""
    from rpe_amd import pose_net, synth
    from rpe_amd.se3 import SE3
    h, w = 352, 384
    cfg = synth.model_config(h, w, iters=12, lbgfs_iters=100)
    model = synth.init_synthetic_weights(pose_net.PoseNet(cfg)).cuda()
    fr = synth.stereo_frames(3, 2, h, w)
    a = {k: v.cuda() for k, v in fr.items()}
    # (1) eval mode: the differentiable head path equals the fused inference path, and forward() equals infer()
    model.eval()
    x = torch.randn(2, 264, h // 8, w // 8, device='cuda')
    with torch.no_grad():
        ref_map = model.weight_head_2d(x)
    got_map = model.weight_head_2d(x.requires_grad_(True))
    assert got_map.requires_grad and float((got_map - ref_map).abs().max()) < 1e-5
    depth1, sflow1, valid1 = model.flow2depth(a['image1l'], a['image2r'], a['baseline'])      # any right image: only the plumbing matters
    pose_tan, d1, d2, maps = model(a['image1l'], a['image2l'], a['K'], a['baseline'], a['image2r'], a['image2r'],
                                   mask1=a['mask1'], mask2=a['mask2'], ret_confmap=True)
    assert pose_tan.shape == (2, 6) and pose_tan.requires_grad and torch.equal(d1, depth1)
    m1 = a['mask1'] & valid1
    inf = model.infer(a['image1l'], a['image2l'], a['K'], a['baseline'], depth1, a['image2r'], m1, a['mask2'].clone(), sflow1)
    assert float((SE3(inf.data).log() - pose_tan.detach()).abs().max()) < 1e-5
    # (2) train mode (batch-norm statistics of the batch, as the reference trains): gradients arrive
    model.train().freeze_flow(True)
    assert not model.flow.training and not any(p.requires_grad for p in model.flow.parameters())
    pose_tan, _, _ = model(a['image1l'], a['image2l'], a['K'], a['baseline'], a['image2r'], a['image2r'], mask1=a['mask1'], mask2=a['mask2'])
    gt = torch.zeros_like(pose_tan)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        (pose_tan - gt).abs().sum().backward()
    heads = [p for n_, p in model.named_parameters() if n_.startswith('weight_head')]
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in heads)
    assert sum(float(p.grad.abs().sum()) for p in heads) > 0.0
    assert model.loss_weight.grad is not None and bool(torch.isfinite(model.loss_weight.grad).all()) and float(model.loss_weight.grad.abs().sum()) > 0
    with pytest.raises(NotImplementedError):
        model.freeze_flow(False)


def test_backward_falls_back_per_row_when_h_is_not_positive_definite(rpe, monkeypatch):
    """ddn's _solve_linear_system (behind declerative_node_lie.py:58) retries the rows a batched Cholesky fails on and falls back to
    an LU solve for them; the reference zeroes the batch's gradients only when that fails too.  H is forced here: row 1 indefinite
    but regular -> no warning, its gradient is that of the LU solution; row 1 singular -> the reference's warning and zeros."""
    from rpe_amd import ops
    from rpe_amd import pose_head as ph
    c = synth.solver_case(73, 2, 32, 48, outliers=False)
    xs = [a.cuda() for a in synth.solver_args(c)]
    head = ph.DPoseSE3Head(None, lbgfs_iters=100)
    T, _ = head.solve(*xs)
    y = T.data.reshape(2, 7).float()
    v = torch.randn(2, 1, 6, device='cuda', dtype=torch.float64)
    real = ops.pose_backward_moments
    needs = [False, False, False, True, True, False, False, False, True]

    def forced(kind):
        def f(*a):
            g2u, g3u, H = real(*a)
            H = H.clone()
            if kind == 'indefinite':
                H[1] = torch.diag(torch.tensor([2.0, 1.0, -3.0, 1.5, 2.5, 1.0], dtype=H.dtype, device=H.device)) * H[0].abs().max()
            else:
                H[1] = 0.0
            return g2u, g3u, H
        return f
    monkeypatch.setattr(ops, 'pose_backward_moments', forced('indefinite'))
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        g = head.gradient(*xs, y=y, v=v, needs=needs)
    assert float(g[3][0].abs().max()) > 0 and float(g[3][1].abs().max()) > 0 and bool(torch.isfinite(g[8]).all())
    monkeypatch.setattr(ops, 'pose_backward_moments', real)
    want0 = head.gradient(*xs, y=y, v=v, needs=needs)[3][0]
    assert torch.equal(g[3][0], want0)                                          # the regular row is untouched by the other row's fallback
    monkeypatch.setattr(ops, 'pose_backward_moments', forced('singular'))
    with pytest.warns(UserWarning, match='positive definite'):
        g = head.gradient(*xs, y=y, v=v, needs=needs)
    assert float(g[3].abs().max()) == 0.0 and float(g[8].abs().max()) == 0.0
