"""Oracle (test infrastructure): backward of the declarative pose layer in closed form, PyTorch-CPU float64.

Restates what the reference computes with autograd in ``DeclarativeNodeLie.gradient`` /
``_get_objective_derivatives`` (core/optimization/declerative_node_lie.py:13-82,106-126) for the objective of
core/pose/pose_head.py:12-58 evaluated through the double-backward ``Transform`` of
core/geometry/pinhole_transforms.py:33-76.  PINNED: ``tests/golden/backward_{a,b}.npz`` are gradients, fY and fYY
produced by the reference's own code (oracle/gen_golden.py::gen_backward); ``tests/test_oracle_golden.py`` checks this
file against them.

Notation: X_p = R p1 + t, J_p = [I | -[X_p]x] (tangent Jacobian of the left perturbation), g_p = dl_p/dX and
M_p = d2l_p/dX2 of the per-pixel loss.  ``Transform.backward`` builds grad_T from its SAVED OUTPUT, which autograd
re-attaches to the graph, so J_p is differentiated too (that is why the reference has to symmetrise fYY):
    fY       = sum_p J_p^T g_p
    fYY[i,j] = sum_p J_i . M_p J_j  +  [i >= 3] sum_p (J_j x g_p)_(i-3)          (J_i: i-th column of J_p)
    H        = (fYY + fYY^T) / 2 ,   u = -H^-1 v
    grad_x   = sum_p (d fY / d x)^T u   (closed forms in ``layer_backward``)
"""
import warnings

import torch

from . import se3 as _se3
from .pose_head import _prep

F64 = torch.float64


def _cross(a, b):
    return torch.stack((a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                        a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]), dim=-1)


def pixel_terms(P, T):
    """Per-pixel quantities at pose T (n,7) f64: X (n,HW,3), g2/g3 (unit loss weight and unit per-pixel weight),
    M (n,HW,3,3) with all weights, dpi (n,HW,2,3) = d pi_c/d(KX), s2, s3 (n,HW) gate * 2 * normalisation."""
    n, h, w = P['n'], P['h'], P['w']
    hw = h * w
    X = _se3.se3_act(T.reshape(n, 1, 7), P['pcl1'].permute(0, 2, 1))
    K = P['K']
    a = torch.einsum('nij,npj->npi', K, X)
    d = torch.clamp(a[..., 2], 1e-12, None)
    s = (a[..., 2] >= 1e-12).to(F64)
    fx = P['pix'][0][None] + P['flow'][:, 0]
    fy = P['pix'][1][None] + P['flow'][:, 1]
    r0, r1 = fx - a[..., 0] / d, fy - a[..., 1] / d
    res = (r0 * r0 + r1 * r1) * P['w1']
    inimg = (fx > 0) & (fy > 0) & (fx < w) & (fy < h)
    bad = torch.isinf(res) | torch.isnan(res) | ~inimg | ~P['m1']
    s2 = torch.where(bad, torch.zeros_like(res), torch.full_like(res, 2.0 / hw / hw))
    s3 = (P['m1'] & P['m2']).to(F64) * (2.0 / hw)
    z = torch.zeros_like(d)
    dpi = torch.stack((torch.stack((1 / d, z, -s * a[..., 0] / d ** 2), -1), torch.stack((z, 1 / d, -s * a[..., 1] / d ** 2), -1)), -2)
    ga = -s2[..., None] * (r0[..., None] * dpi[..., 0, :] + r1[..., None] * dpi[..., 1, :])
    ga = torch.where(bad[..., None], torch.zeros_like(ga), ga)
    g2 = torch.einsum('nji,npj->npi', K, ga)                                   # K^T ga
    g3 = s3[..., None] * (X - P['pcl2'].permute(0, 2, 1))
    Ha = torch.einsum('npci,npcj->npij', dpi, dpi)
    cr = s / d ** 2
    Ha[..., 0, 2] += r0 * cr; Ha[..., 2, 0] += r0 * cr
    Ha[..., 1, 2] += r1 * cr; Ha[..., 2, 1] += r1 * cr
    Ha[..., 2, 2] -= 2 * s * (r0 * a[..., 0] + r1 * a[..., 1]) / d ** 3
    k2 = s2 * P['w1'] * P['lw'][:, 1:2]
    k2 = torch.where(bad, torch.zeros_like(k2), k2)
    M = k2[..., None, None] * torch.einsum('nki,npkl,nlj->npij', K, Ha, K)
    M = torch.where(bad[..., None, None], torch.zeros_like(M), M)
    M = M + (s3 * P['w2'] * P['lw'][:, 0:1])[..., None, None] * torch.eye(3, dtype=F64)
    return dict(X=X, g2=g2, g3=g3, M=M, dpi=dpi, s2=s2, s3=s3)


def _jcols(X):
    """(n,HW,6,3): the six columns of J = [I | -[X]x] (e_0, e_1, e_2, e_0 x X, e_1 x X, e_2 x X)."""
    eye = torch.eye(3, dtype=F64).expand(*X.shape[:-1], 3, 3)
    rot = torch.stack([_cross(eye[..., k, :], X) for k in range(3)], dim=-2)
    return torch.cat((eye, rot), dim=-2)


def objective_derivatives(P, T):
    """fY (n,6), fYY (n,6,6) (NOT symmetrised, as _get_objective_derivatives returns it), and the per-term tangent
    gradients with unit loss weight g2u, g3u (n,6)."""
    t = pixel_terms(P, T)
    J = _jcols(t['X'])
    g2w, g3w = t['g2'] * P['w1'][..., None], t['g3'] * P['w2'][..., None]
    g2u = torch.einsum('npic,npc->ni', J, g2w)
    g3u = torch.einsum('npic,npc->ni', J, g3w)
    lw = P['lw']
    fY = lw[:, 1:2] * g2u + lw[:, 0:1] * g3u
    g = lw[:, 1:2, None] * g2w + lw[:, 0:1, None] * g3w
    fYY = torch.einsum('npic,npcd,npjd->nij', J, t['M'], J)
    extra = torch.stack([_cross(J[..., j, :], g).sum(dim=1) for j in range(6)], dim=-1)     # (n, 3 (k), 6 (j)) = sum_p (J_j x g)_k
    fYY[:, 3:, :] += extra
    return fY, fYY, g2u, g3u, t, g


def layer_backward(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight, vec7, v, eps=1e-3):
    """Gradients the layer's backward returns for (flow, pcl1, pcl2, w1, w2, loss_weight), float64, given the layer's
    float32 output pose ``vec7`` (n,[1,]7) and the incoming tangent gradient ``v`` (n,[1,]6)
    (DeclarativeFunctionLie.backward :249-267 -> gradient :13-82).  All zeros when the optimality check fails."""
    P = _prep(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight)
    n, h, w = P['n'], P['h'], P['w']
    T = vec7.reshape(n, 7).to(F64)
    fY, fYY, g2u, g3u, t, g = objective_derivatives(P, T)
    shapes = dict(flow=(n, 2, h, w), pcl1=(n, 3, h, w), pcl2=(n, 3, h, w), w1=(n, 1, h, w), w2=(n, 1, h, w), loss_weight=(n, 2))
    if not torch.allclose(fY, torch.zeros_like(fY), rtol=0.0, atol=eps):            # :43-47
        warnings.warn('Non-zero objective function gradient at y')
        return {k: torch.zeros(s, dtype=F64) for k, s in shapes.items()}, fY, fYY
    H = 0.5 * (fYY + fYY.transpose(1, 2))                                           # :51
    try:
        u = torch.cholesky_solve(-v.reshape(n, 6, 1).to(F64), torch.linalg.cholesky(H))[..., 0]
    except Exception:
        warnings.warn('linear system is not positive definite')
        return {k: torch.zeros(s, dtype=F64) for k, s in shapes.items()}, fY, fYY
    u = torch.nan_to_num(u, nan=0.0, posinf=float('inf'), neginf=float('-inf'))     # u[isnan] = 0
    X = t['X']
    Ju = u[:, None, :3] + _cross(u[:, None, 3:].expand_as(X), X)                    # J u = u_tau + u_phi x X
    lw = P['lw']
    out = {}
    out['w1'] = lw[:, 1:2] * (Ju * t['g2']).sum(-1)
    out['w2'] = lw[:, 0:1] * (Ju * t['g3']).sum(-1)
    out['pcl2'] = (-(t['s3'] * P['w2'] * lw[:, 0:1])[..., None] * Ju).permute(0, 2, 1)
    KJ = torch.einsum('nij,npj->npi', P['K'], Ju)
    out['flow'] = (-(t['s2'] * P['w1'] * lw[:, 1:2])[..., None] * torch.einsum('npci,npi->npc', t['dpi'], KJ)).permute(0, 2, 1)
    R = _se3.se3_matrix(T)[:, :3, :3]
    MJ = torch.einsum('npij,npj->npi', t['M'], Ju) + _cross(g, u[:, None, 3:].expand_as(g))
    out['pcl1'] = torch.einsum('nji,npj->npi', R, MJ).permute(0, 2, 1)               # R^T (M J u + g x u_phi)
    out['loss_weight'] = torch.stack(((u * g3u).sum(-1), (u * g2u).sum(-1)), dim=-1)
    out = {k: torch.nan_to_num(x, nan=0.0, posinf=float('inf'), neginf=float('-inf')).reshape(shapes[k]) for k, x in out.items()}
    return out, fY, fYY
