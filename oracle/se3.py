"""Oracle (test infrastructure): SE(3) value type with the lietorch call surface the path uses.

lietorch (princeton-vl/lietorch, unpinned HEAD, reference README.md:37) is not installed and its
source is not under /root/reference, so this restates its published algorithm:

  * data layout  : 7-vector ``[tx, ty, tz, qx, qy, qz, qw]``; tangent ``[tau(3), phi(3)]``
                   (translation first -- reference core/geometry/pinhole_transforms.py:39-42 builds
                   the Jacobian as ``[I | -[X]x]``, scripts/train_posenet.py:46-47 slices ``[:, :3]``
                   as translation).
  * exp / log    : unit-quaternion exponential with Taylor guards at ``theta^2 < EPS`` (EPS = 1e-6),
                   translation through the SO(3) left Jacobian / its inverse.
  * mul / inv/act: ``(t1 + R1 t2, q1 q2)``, ``(-R^T t, q*)``, ``R p + t`` with the rotation applied
                   as ``p + w*uv + v x uv``, ``uv = 2 v x p``.
  * LieGroupParameter: tangent-zero tensor subclass whose ``add_`` is the left retraction
                   ``group <- exp(alpha*u) * group`` and whose use in an expression is
                   ``exp(self) * group`` (so autograd yields left-perturbation gradients).

Call sites that fix the surface: pose_head.py:68, pinhole_transforms.py:29,51,
declerative_node_lie.py:233-234, pose_net.py:96, pose_estimator.py:81-91,105,
utils/trajectory.py:14,48,58 (all paths relative to /root/reference).

Everything is written with differentiable torch ops so the reference's own files can run on top
of it when ``oracle/gen_golden.py`` seeds it into ``sys.modules['lietorch']``.
"""
import math
import torch

EPS = 1e-6


# ----------------------------------------------------------------------------- quaternion helpers
def _cross(a, b):
    ax, ay, az = a.unbind(-1)
    bx, by, bz = b.unbind(-1)
    return torch.stack((ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx), dim=-1)


def quat_rotate(q, p):
    """Rotate points p (...,3) by unit quaternion q (...,4) = [x,y,z,w]."""
    v, w = q[..., :3], q[..., 3:4]
    uv = 2.0 * _cross(v, p)
    return p + w * uv + _cross(v, uv)


def quat_mul(q1, q2):
    x1, y1, z1, w1 = q1.unbind(-1)
    x2, y2, z2, w2 = q2.unbind(-1)
    return torch.stack((
        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
        w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2), dim=-1)


def quat_conj(q):
    return torch.cat((-q[..., :3], q[..., 3:4]), dim=-1)


def quat_normalize(q):
    return q / torch.linalg.norm(q, dim=-1, keepdim=True)


def hat(v):
    """so(3) hat map; same matrix as reference core/utils/pytorch.py:144-161 (skewmat)."""
    x, y, z = v.unbind(-1)
    o = torch.zeros_like(x)
    return torch.stack((torch.stack((o, -z, y), -1),
                        torch.stack((z, o, -x), -1),
                        torch.stack((-y, x, o), -1)), dim=-2)


# ----------------------------------------------------------------------------- so(3) pieces
def so3_exp(phi):
    theta_sq = (phi * phi).sum(-1, keepdim=True)
    small = theta_sq < EPS
    theta_p4 = theta_sq * theta_sq
    safe_sq = torch.where(small, torch.ones_like(theta_sq), theta_sq)
    theta = torch.sqrt(safe_sq)
    imag = torch.where(small, 0.5 - theta_sq / 48.0 + theta_p4 / 3840.0, torch.sin(0.5 * theta) / theta)
    real = torch.where(small, 1.0 - theta_sq / 8.0 + theta_p4 / 384.0, torch.cos(0.5 * theta))
    return torch.cat((imag * phi, real), dim=-1)


def so3_log(q):
    v, w = q[..., :3], q[..., 3:4]
    sq_n = (v * v).sum(-1, keepdim=True)
    small = sq_n < EPS * EPS
    safe_sq = torch.where(small, torch.ones_like(sq_n), sq_n)
    n = torch.sqrt(safe_sq)
    w_small = w.abs() < EPS
    safe_w = torch.where(w_small, torch.ones_like(w), w)
    taylor = 2.0 / safe_w - (2.0 / 3.0) * sq_n / (safe_w * safe_w * safe_w)
    pi_branch = torch.where(w > 0, math.pi / n, -math.pi / n)
    reg = 2.0 * torch.atan(n / safe_w) / n
    coef = torch.where(small, taylor, torch.where(w_small, pi_branch, reg))
    return coef * v


def so3_left_jacobian(phi):
    theta_sq = (phi * phi).sum(-1, keepdim=True)
    small = theta_sq < EPS
    safe_sq = torch.where(small, torch.ones_like(theta_sq), theta_sq)
    theta = torch.sqrt(safe_sq)
    c1 = torch.where(small, 0.5 - theta_sq / 24.0, (1.0 - torch.cos(theta)) / safe_sq)
    c2 = torch.where(small, 1.0 / 6.0 - theta_sq / 120.0, (theta - torch.sin(theta)) / (safe_sq * theta))
    Phi = hat(phi)
    eye = torch.eye(3, dtype=phi.dtype, device=phi.device).expand(Phi.shape)
    return eye + c1[..., None] * Phi + c2[..., None] * (Phi @ Phi)


def so3_left_jacobian_inv(phi):
    theta_sq = (phi * phi).sum(-1, keepdim=True)
    small = theta_sq < EPS
    safe_sq = torch.where(small, torch.ones_like(theta_sq), theta_sq)
    theta = torch.sqrt(safe_sq)
    half = 0.5 * theta
    c2 = torch.where(small, torch.full_like(theta_sq, 1.0 / 12.0) + theta_sq / 720.0,
                     (1.0 - theta * torch.cos(half) / (2.0 * torch.sin(half))) / safe_sq)
    Phi = hat(phi)
    eye = torch.eye(3, dtype=phi.dtype, device=phi.device).expand(Phi.shape)
    return eye - 0.5 * Phi + c2[..., None] * (Phi @ Phi)


# ----------------------------------------------------------------------------- se(3) on raw tensors
def se3_exp(xi):
    tau, phi = xi[..., :3], xi[..., 3:]
    q = so3_exp(phi)
    t = (so3_left_jacobian(phi) @ tau[..., None])[..., 0]
    return torch.cat((t, q), dim=-1)


def se3_log(T):
    t, q = T[..., :3], T[..., 3:]
    phi = so3_log(q)
    tau = (so3_left_jacobian_inv(phi) @ t[..., None])[..., 0]
    return torch.cat((tau, phi), dim=-1)


def se3_mul(A, B):
    ta, qa = A[..., :3], A[..., 3:]
    tb, qb = B[..., :3], B[..., 3:]
    return torch.cat((ta + quat_rotate(qa, tb), quat_normalize(quat_mul(qa, qb))), dim=-1)


def se3_inv(T):
    t, q = T[..., :3], T[..., 3:]
    qi = quat_conj(q)
    return torch.cat((-quat_rotate(qi, t), qi), dim=-1)


def se3_act(T, p):
    return quat_rotate(T[..., 3:], p) + T[..., :3]


def se3_matrix(T):
    t, q = T[..., :3], T[..., 3:]
    eye = torch.eye(3, dtype=T.dtype, device=T.device).expand(*T.shape[:-1], 3, 3)
    R = quat_rotate(q[..., None, :], eye)          # rows = rotated basis vectors
    R = R.transpose(-1, -2)                        # columns = R e_i
    top = torch.cat((R, t[..., None]), dim=-1)
    bot = torch.zeros(*T.shape[:-1], 1, 4, dtype=T.dtype, device=T.device)
    bot[..., 0, 3] = 1.0
    return torch.cat((top, bot), dim=-2)


# ----------------------------------------------------------------------------- lietorch-like types
class SE3:
    manifold_dim = 6
    embedded_dim = 7

    def __init__(self, data):
        if isinstance(data, SE3):
            data = data.data
        self.data = data

    # -- constructors
    @classmethod
    def Identity(cls, *batch_shape, **kwargs):
        kwargs.pop('requires_grad', None)
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
    def Random(cls, *batch_shape, sigma=1.0, **kwargs):
        xi = sigma * torch.randn(*batch_shape, 6, **kwargs)
        return cls.exp(xi)

    @classmethod
    def exp(cls, xi):
        return cls(se3_exp(xi))

    # -- attributes
    @property
    def shape(self):
        return self.data.shape[:-1]

    @property
    def tangent_shape(self):
        return self.data.shape[:-1] + (6,)

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def device(self):
        return self.data.device

    @property
    def requires_grad(self):
        return self.data.requires_grad

    def to(self, *args, **kwargs):
        return SE3(self.data.to(*args, **kwargs))

    def float(self, *_):
        return SE3(self.data.float())

    def double(self):
        return SE3(self.data.double())

    def detach(self):
        return SE3(self.data.detach())

    def clone(self):
        return SE3(self.data.clone())

    def view(self, *shape):
        return SE3(self.data.view(*shape, 7))

    def squeeze(self, dim=None):
        return SE3(self.data.squeeze(dim) if dim is not None else self.data.squeeze())

    def __getitem__(self, index):
        return SE3(self.data[index])

    def __len__(self):
        return self.data.shape[0]

    # -- group ops
    def vec(self):
        return self.data

    def log(self):
        return se3_log(self.data)

    def inv(self):
        return SE3(se3_inv(self.data))

    def matrix(self):
        return se3_matrix(self.data)

    def scale(self, s):
        s = torch.as_tensor(s, dtype=self.data.dtype, device=self.data.device)
        return SE3(torch.cat((self.data[..., :3] * s, self.data[..., 3:]), dim=-1))

    def act(self, p):
        return se3_act(self.data, p)

    def retr(self, a):
        return SE3(se3_mul(se3_exp(a), self.data))

    def __mul__(self, other):
        if isinstance(other, LieGroupParameter):
            other = other.retr()
        if isinstance(other, SE3):
            return SE3(se3_mul(self.data, other.data))
        return se3_act(self.data, other)

    def __repr__(self):
        return "SE3({})".format(self.data)


class LieGroupParameter(torch.Tensor):
    """Tangent-zero parameter carrying a group element; ``add_`` is the left retraction."""

    @staticmethod
    def __new__(cls, group, requires_grad=True):
        data = torch.zeros(group.tangent_shape, dtype=group.dtype, device=group.device)
        return torch.Tensor._make_subclass(cls, data, requires_grad)

    def __init__(self, group, requires_grad=True):
        self.group = group.detach()

    def retr(self):
        return self.group.retr(self.as_subclass(torch.Tensor))

    def log(self):
        return self.retr().log()

    def inv(self):
        return self.retr().inv()

    def __mul__(self, other):
        if isinstance(other, LieGroupParameter):
            return self.retr() * other.retr()
        return self.retr() * other

    def add_(self, update, alpha=1.0):
        with torch.no_grad():
            upd = update.as_subclass(torch.Tensor) if isinstance(update, torch.Tensor) else update
            self.group = SE3(se3_mul(se3_exp(alpha * upd), self.group.data))
        return self

    def __getitem__(self, index):
        return self.retr().__getitem__(index)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        return super().__torch_function__(func, types, args, kwargs or {})
