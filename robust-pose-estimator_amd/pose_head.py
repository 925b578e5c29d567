"""Pose layer host mirror (boundary B2): ``DeclarativeLayerLie(DPoseSE3Head(img_coords, lbgfs_iters))(flow, pcl1,
pcl2, w1, w2, mask1, mask2, K, loss_weight) -> (vec7 (n,1,7) f32, log6 (n,1,6) f32)`` exactly as the reference
calls it (core/pose/pose_net.py:23,55,84; core/optimization/declerative_node_lie.py:223-247,283-284;
core/pose/pose_head.py:60-79), executed by ``rpe_pose_solve`` on the device with no host synchronisation.

Differences from the reference, all documented in DESIGN.md:
  * the backward (training) evaluates the implicit-differentiation formulas of
    declerative_node_lie.py:13-82 in closed form on the device (csrc/pose_backward.hip) instead of differentiating the
    objective twice with autograd; same gradients (golden: the reference's own backward), float64 arithmetic
  * a batch of n rows is n independent solves (what the reference does per frame; its n>1 coupling through a
    shared L-BFGS history only exists in training)
  * ``solver='gn'`` selects the Gauss-Newton mode of the same kernel
"""
import torch

from . import ops
from .se3 import SE3


def create_img_coords_t(y, x, b=1, device='cpu'):
    """core/geometry/pinhole_transforms.py:7-19 -- (3, y*x) pixel centres [x+.5, y+.5, 1], x fastest."""
    xs = torch.arange(x, dtype=torch.float32, device=device) + 0.5
    ys = torch.arange(y, dtype=torch.float32, device=device) + 0.5
    xm = xs[None, :].expand(y, x).reshape(-1)
    ym = ys[:, None].expand(y, x).reshape(-1)
    return torch.stack((xm, ym, torch.ones_like(xm)), dim=0)


class DPoseSE3Head:
    """Holds the solver configuration; the pixel grid is implicit in the kernel (x+.5, y+.5)."""

    def __init__(self, img_coordinates=None, lbgfs_iters=100, dbg=False, solver='lbfgs'):
        self.img_coordinates = img_coordinates
        self.lbgfs_iters = lbgfs_iters
        self.solver = solver
        self.losses = []
        self.last_info = None
        # 1: every row is reduced with the block partition it would get alone (rpe_solve_opts.partition_rows), so a row's pose does not
        # depend on the batch it is solved in, bit for bit; 0: the partition follows the batch (one resident round of workgroups)
        self.partition_rows = 0

    @property
    def mode(self):
        return ops.SOLVER_GN if self.solver == 'gn' else ops.SOLVER_LBFGS

    def objective(self, *xs, y):
        """pose_head.py:53-58 at pose y (SE3 or (n,[1,]7) tensor): returns the (n,) f64 objective."""
        T = y[0] if isinstance(y, (tuple, list)) else y
        T = T.data if isinstance(T, SE3) else T
        n = xs[0].shape[0]
        return ops.pose_reduce(*xs, T.reshape(n, 7).double())['f']

    def solve(self, *xs):
        xs = [x.detach() if isinstance(x, torch.Tensor) else x for x in xs]
        T, vec7, log6, info = ops.pose_solve(*xs, iters=self.lbgfs_iters, mode=self.mode, partition_rows=self.partition_rows)
        self.last_info = info
        return SE3(T[:, None]), (vec7, log6)

    def gradient(self, *xs, y, v, needs=None, eps=1e-3):
        """DeclarativeNodeLie.gradient (declerative_node_lie.py:13-82) at the layer's float32 output pose ``y`` (n,7) for
        the incoming tangent gradient ``v`` (n,[1,]6): one gradient (or None) per input.  As the reference: all zeros
        with a warning when the solver did not reach |fY| <= eps (:43-47) or the 6x6 system is not positive definite
        (:59-63); NaNs in u and in the result are zeroed (:67,76)."""
        import warnings
        flow, pcl1, pcl2, w1, w2, mask1, mask2, K, lw = [x.detach() for x in xs]
        n = flow.shape[0]
        needs = [True] * 9 if needs is None else list(needs)
        T = y.reshape(n, 7).double()
        g2u, g3u, H = ops.pose_backward_moments(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, lw, T)
        lwd = lw.double()
        fY = lwd[:, 1:2] * g2u + lwd[:, 0:1] * g3u

        def zeros():
            return tuple(torch.zeros_like(x) if nd and x.is_floating_point() else None for x, nd in zip(xs, needs))
        if not bool((fY.abs() <= eps).all()):
            warnings.warn('Non-zero objective function gradient at y:\n{}'.format(fY.detach().squeeze().cpu().numpy()))
            return zeros()
        # ddn's _solve_linear_system (anucvml/ddn node.py, called at declerative_node_lie.py:58): batched Cholesky; rows it fails on
        # are retried one by one and fall back to an LU solve; only when that fails too does the reference's ``except`` zero the
        # gradients of the whole batch (:59-63)
        rhs = -v.reshape(n, 6, 1).double()
        L, info = torch.linalg.cholesky_ex(H)
        bad = info != 0
        u = torch.cholesky_solve(rhs, torch.where(bad[:, None, None], torch.eye(6, dtype=H.dtype, device=H.device), L))
        if bool(bad.any()):
            lu, lu_info = torch.linalg.solve_ex(H[bad], rhs[bad])
            if bool((lu_info != 0).any()) or not bool(torch.isfinite(lu).all()):
                warnings.warn('linear system is not positive definite ')
                return zeros()
            u[bad] = lu
        u = u[..., 0]
        u = torch.where(torch.isnan(u), torch.zeros_like(u), u)
        names = ('flow', 'pcl1', 'pcl2', 'w1', 'w2')
        want = [nm for nm, nd in zip(names, needs[:5]) if nd]
        g = ops.pose_backward_grads(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, lw, T, u, want) if want else {}
        out = [g.get(nm) for nm in names] + [None, None, None, None]
        if needs[8]:
            glw = torch.stack(((u * g3u).sum(-1), (u * g2u).sum(-1)), dim=-1)
            out[8] = torch.where(torch.isnan(glw), torch.zeros_like(glw), glw).to(lw.dtype)
        return tuple(out)


class DeclarativeFunctionLie(torch.autograd.Function):
    """declerative_node_lie.py:211-267: forward = solve, outputs (vec7, log6) float32; backward = implicit
    differentiation in tangent space, driven by the gradient that arrives on log6 (the vec7 gradient is ignored, :264)."""

    @staticmethod
    def forward(ctx, problem, *inputs):
        with torch.no_grad():
            _, (vec7, log6) = problem.solve(*inputs)
        ctx.problem = problem
        ctx.save_for_backward(vec7, *inputs)
        return vec7[:, None].clone(), log6[:, None]   # (n,1,7), (n,1,6) float32 (:233-234)

    @staticmethod
    def backward(ctx, grad_vec7, grad_log6):
        vec7, *inputs = ctx.saved_tensors
        grads = ctx.problem.gradient(*inputs, y=vec7, v=grad_log6, needs=ctx.needs_input_grad[1:])
        return (None, *grads)


class DeclarativeLayerLie(torch.nn.Module):
    def __init__(self, problem):
        super().__init__()
        self.problem = problem

    def forward(self, *inputs):
        if torch.is_grad_enabled() and any(isinstance(x, torch.Tensor) and x.requires_grad for x in inputs):
            return DeclarativeFunctionLie.apply(self.problem, *inputs)
        with torch.no_grad():
            _, (vec7, log6) = self.problem.solve(*inputs)
        return vec7[:, None], log6[:, None]          # (n,1,7), (n,1,6) float32 (declerative_node_lie.py:233-234)
