"""Pose layer host mirror (boundary B2): ``DeclarativeLayerLie(DPoseSE3Head(img_coords, lbgfs_iters))(flow, pcl1,
pcl2, w1, w2, mask1, mask2, K, loss_weight) -> (vec7 (n,1,7) f32, log6 (n,1,6) f32)`` exactly as the reference
calls it (core/pose/pose_net.py:23,55,84; core/optimization/declerative_node_lie.py:223-247,283-284;
core/pose/pose_head.py:60-79), executed by ``rpe_pose_solve`` on the device with no host synchronisation.

Differences from the reference, all documented in DESIGN.md:
  * forward / inference only (the implicit-differentiation backward is out of scope)
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
        T, vec7, log6, info = ops.pose_solve(*xs, iters=self.lbgfs_iters, mode=self.mode)
        self.last_info = info
        return SE3(T[:, None]), (vec7, log6)


class DeclarativeLayerLie(torch.nn.Module):
    def __init__(self, problem):
        super().__init__()
        self.problem = problem

    @torch.no_grad()
    def forward(self, *inputs):
        _, (vec7, log6) = self.problem.solve(*inputs)
        return vec7[:, None], log6[:, None]          # (n,1,7), (n,1,6) float32 (declerative_node_lie.py:233-234)
