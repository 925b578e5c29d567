"""lietorch-style ``SE3`` value type on the rpe_se3_* HIP kernels (boundary B4 of SURVEY.md section 8b).

Covers the call surface the reference uses on the inference path: ``Identity / IdentityLike / InitFromVec /
exp / Random``, ``vec / log / inv / matrix / scale / __mul__ (group or points) / __getitem__``
(core/pose/pose_estimator.py:81-91,105; core/pose/pose_net.py:96; core/geometry/pinhole_transforms.py:29,51;
core/utils/trajectory.py:14,48,58).  Data layout ``[tx ty tz qx qy qz qw]``, tangent ``[tau phi]``.
Tensors must be on the GPU; every group operation runs in librpe_hip.so (no CPU fallback).
"""
import torch

from . import ops


class SE3:
    manifold_dim = 6
    embedded_dim = 7

    def __init__(self, data):
        self.data = data.data if isinstance(data, SE3) else data

    @classmethod
    def Identity(cls, *batch_shape, **kwargs):
        data = torch.zeros(*batch_shape, 7, **kwargs)
        data[..., 6] = 1.0
        return cls(data)

    @classmethod
    def IdentityLike(cls, G):
        return cls.Identity(*G.shape, dtype=G.dtype, device=G.device)

    @classmethod
    def InitFromVec(cls, vec):
        return cls(vec)

    @classmethod
    def exp(cls, xi):
        return cls(ops.se3_exp(xi))

    @classmethod
    def Random(cls, *batch_shape, sigma=1.0, **kwargs):
        return cls.exp(sigma * torch.randn(*batch_shape, 6, **kwargs))

    @property
    def shape(self):
        return self.data.shape[:-1]

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def device(self):
        return self.data.device

    def to(self, *a, **k):
        return SE3(self.data.to(*a, **k))

    def float(self, *_):
        return SE3(self.data.float())

    def double(self):
        return SE3(self.data.double())

    def __getitem__(self, idx):
        return SE3(self.data[idx])

    def vec(self):
        return self.data

    def log(self):
        return ops.se3_log(self.data)

    def inv(self):
        return SE3(ops.se3_inv(self.data))

    def scale(self, s):
        s = torch.as_tensor(s, dtype=self.data.dtype, device=self.data.device)
        return SE3(torch.cat((self.data[..., :3] * s, self.data[..., 3:]), dim=-1))

    def matrix(self):
        """4x4 homogeneous matrices: columns = images of the basis vectors and of the origin."""
        flat = self.data.reshape(-1, 7).contiguous()
        basis = torch.zeros(flat.shape[0], 4, 3, dtype=flat.dtype, device=flat.device)
        basis[:, 0, 0] = basis[:, 1, 1] = basis[:, 2, 2] = 1.0
        img = ops.se3_act(flat, basis)                       # rows: R e_i + t (i < 3), t
        t = img[:, 3]
        R = (img[:, :3] - t[:, None]).transpose(1, 2)
        M = torch.zeros(flat.shape[0], 4, 4, dtype=flat.dtype, device=flat.device)
        M[:, :3, :3], M[:, :3, 3], M[:, 3, 3] = R, t, 1.0
        return M.reshape(*self.shape, 4, 4)

    def __mul__(self, other):
        if isinstance(other, SE3):
            return SE3(ops.se3_mul(self.data, other.data))
        pts = other                                          # (n, m, 3) points, pose (n,7) or (n,1,7) broadcast
        return ops.se3_act(self.data.reshape(pts.shape[0], 7), pts.contiguous())

    def __repr__(self):
        return f'SE3({self.data})'
