"""Oracle (test infrastructure): the weighted 2D-reprojection + 3D point-to-point SE(3) solve.

Restates, in plain PyTorch-CPU float64 with closed-form derivatives (no autograd):

  * ``img_coords``            <- core/geometry/pinhole_transforms.py:7-19  (create_img_coords_t)
  * ``evaluate``              <- core/pose/pose_head.py:12-58 (reprojection_objective, depth_objective,
                                 objective) + pinhole_transforms.py:28-30,90-99 (transform, project)
                                 + the left-perturbation Jacobian [I | -[X]x] of :39-42
  * ``lbfgs_solve``           <- core/pose/pose_head.py:60-79 (solve: f64 cast, Identity start, closure
                                 with clip_grad_norm_(y, 10)) driving torch.optim.LBFGS.step
                                 (torch/optim/lbfgs.py, line_search_fn=None branch) with the
                                 LieGroupParameter left retraction group <- exp(t d) * group
  * ``gn_solve``              <- the north-star Gauss-Newton variant on the same objective
                                 (6x6 normal equations H delta = -g, same retraction)
  * ``declarative_forward``   <- core/optimization/declerative_node_lie.py:223-247 (vec7 / log6, f32)

All paths are relative to /root/reference.  ``tests/golden/solver_*.npz`` were produced by running
the reference's own pose_head.py (see oracle/gen_golden.py); tests check this file against them.
"""
import math
import numpy as np
import torch

from . import se3 as _se3

F64 = torch.float64


def img_coords(h, w, dtype=torch.float32):
    """(3, h*w) pixel-centre grid, x fastest: [x+.5, y+.5, 1] (pinhole_transforms.py:7-19)."""
    x = torch.arange(w, dtype=dtype) + 0.5
    y = torch.arange(h, dtype=dtype) + 0.5
    xm = x[None, :].expand(h, w).reshape(-1)
    ym = y[:, None].expand(h, w).reshape(-1)
    return torch.stack((xm, ym, torch.ones_like(xm)), dim=0)


def _prep(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight):
    """pose_head.py:63-64 -- detach/clone and cast every float32 input to float64."""
    n, _, h, w = flow.shape
    d = lambda x: x.detach().to(F64)
    return dict(n=n, h=h, w=w,
                flow=d(flow).reshape(n, 2, -1), pcl1=d(pcl1).reshape(n, 3, -1), pcl2=d(pcl2).reshape(n, 3, -1),
                w1=d(w1).reshape(n, -1), w2=d(w2).reshape(n, -1),
                m1=mask1.reshape(n, -1).bool(), m2=mask2.reshape(n, -1).bool(),
                K=d(K).reshape(n, 3, 3), lw=d(loss_weight).reshape(n, 2),
                pix=img_coords(h, w).to(F64)[:2])


def evaluate(P, T, need_hessian=False):
    """Objective, tangent gradient (and GN Hessian) at poses T (n,7) f64.

    Returns dict with loss2d, loss3d, f (n,), g (n,6) [unclipped df/dxi, xi=(tau,phi), left
    perturbation] and, if requested, H (n,6,6) = 2 * sum c*w*J^T J.
    """
    n, h, w = P['n'], P['h'], P['w']
    hw = h * w
    T = T.reshape(n, 1, 7)
    X = _se3.se3_act(T, P['pcl1'].permute(0, 2, 1))            # (n,HW,3)   transform_forward :28-30
    K = P['K']
    ipts = torch.einsum('nij,npj->npi', K, X)                   # project :93 (bmm(K, X))
    iz = ipts[..., 2]
    depth = torch.clamp(iz, 1e-12, None)                        # :95
    passz = (iz >= 1e-12).to(F64)                               # clamp backward mask
    u = ipts[..., 0] / depth
    v = ipts[..., 1] / depth
    fx = P['pix'][0][None] + P['flow'][:, 0]                    # pose_head.py:19
    fy = P['pix'][1][None] + P['flow'][:, 1]
    ex, ey = fx - u, fy - v
    r2 = (ex * ex + ey * ey) * P['w1']                          # :21-22
    inimg = (fx > 0) & (fy > 0) & (fx < w) & (fy < h)           # :24
    bad = torch.isinf(r2) | torch.isnan(r2) | ~inimg | ~P['m1']  # :25
    gate2 = (~bad).to(F64)
    r2z = torch.where(bad, torch.zeros_like(r2), r2)            # :28
    loss2d = r2z.mean(dim=1) / hw                               # :29 (double normalisation)

    e3 = X - P['pcl2'].permute(0, 2, 1)                         # :41-43
    r3 = (e3 * e3).sum(-1) * P['w2']
    ok3 = P['m1'] & P['m2']                                     # :47
    gate3 = ok3.to(F64)
    r3z = torch.where(ok3, r3, torch.zeros_like(r3))
    loss3d = r3z.mean(dim=1)                                    # :51

    lw = P['lw']
    f = lw[:, 1] * loss2d + lw[:, 0] * loss3d                   # :58
    c2 = (lw[:, 1] / (float(hw) * float(hw)))[:, None]
    c3 = (lw[:, 0] / float(hw))[:, None]

    # ---- gradient, written as autograd would multiply it out (0 * nan stays nan, like the reference)
    gu = -2.0 * ex * P['w1'] * gate2 * c2
    gv = -2.0 * ey * P['w1'] * gate2 * c2
    g_ix = gu / depth
    g_iy = gv / depth
    g_d = -(gu * ipts[..., 0] + gv * ipts[..., 1]) / (depth * depth)
    g_iz = g_d * passz
    gI = torch.stack((g_ix, g_iy, g_iz), dim=-1)
    gX2 = torch.einsum('nij,npi->npj', K, gI)                   # K^T gI
    gX3 = 2.0 * e3 * (P['w2'] * gate3 * c3)[..., None]
    gX = gX2 + gX3
    g = torch.cat((gX.sum(1), _se3._cross(X, gX).sum(1)), dim=-1)   # [I | -[X]x]^T gX
    out = dict(loss2d=loss2d, loss3d=loss3d, f=f, g=g)

    if need_hessian:
        # Jacobian of X wrt xi (left perturbation): J = [I | -[X]x]  (pinhole_transforms.py:39-42)
        eye = torch.eye(3, dtype=F64).expand(n, hw, 3, 3)
        J = torch.cat((eye, -_se3.hat(X)), dim=-1)              # (n,HW,3,6)
        K0, K1, K2 = K[:, None, 0], K[:, None, 1], K[:, None, 2]
        Au = (K0 - (u * passz)[..., None] * K2) / depth[..., None]
        Av = (K1 - (v * passz)[..., None] * K2) / depth[..., None]
        Ju = torch.einsum('npk,npkj->npj', Au, J)
        Jv = torch.einsum('npk,npkj->npj', Av, J)
        s2 = 2.0 * P['w1'] * gate2 * c2
        s2 = torch.where(bad, torch.zeros_like(s2), s2)
        H2 = torch.einsum('npi,npj,np->nij', Ju, Ju, s2) + torch.einsum('npi,npj,np->nij', Jv, Jv, s2)
        s3 = 2.0 * P['w2'] * gate3 * c3
        s3 = torch.where(ok3, s3, torch.zeros_like(s3))
        H3 = torch.einsum('npki,npkj,np->nij', J, J, s3)
        out['H'] = H2 + H3
    return out


def objective(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight, T):
    """pose_head.py:53-58 at pose T (n,7) or (n,1,7)."""
    P = _prep(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight)
    return evaluate(P, torch.as_tensor(T).to(F64).reshape(P['n'], 7))['f']


def _clip(g, max_norm=10.0):
    """torch.nn.utils.clip_grad_norm_(y, 10): one global L2 norm over the whole gradient."""
    total = torch.linalg.vector_norm(g)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return g * coef


def lbfgs_solve(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight, iters, coupled=False,
                history_size=100, tolerance_grad=1e-7, tolerance_change=1e-9, lr=1.0, trace=None):
    """DPoseSE3Head.solve (pose_head.py:60-79) with torch.optim.LBFGS restated.

    ``coupled=True`` reproduces the reference at batch n>1 exactly (one 6n-vector, shared history,
    shared stopping tests, clip over the whole batch).  ``coupled=False`` runs n independent solves,
    which is what the reference does at inference (n == 1, scripts/infer_trajectory.py:57) and what
    the HIP path implements for batches.
    Returns (T (n,7) f64, info dict).
    """
    P = _prep(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight)
    n = P['n']
    T = torch.zeros(n, 7, dtype=F64)
    T[:, 6] = 1.0                                                  # SE3.Identity :68
    groups = [list(range(n))] if coupled else [[i] for i in range(n)]
    max_eval = iters * 5 // 4
    n_iter_out = np.zeros(n, dtype=np.int64)
    evals_out = np.zeros(n, dtype=np.int64)
    stop_out = np.zeros(n, dtype=np.int64)

    def sub(P, rows):
        Q = dict(P)
        Q['n'] = len(rows)
        for k in ('flow', 'pcl1', 'pcl2', 'w1', 'w2', 'm1', 'm2', 'K', 'lw'):
            Q[k] = P[k][rows]
        return Q

    for rows in groups:
        Pg = sub(P, rows)
        Tg = T[rows].clone()

        def closure():
            ev = evaluate(Pg, Tg)
            loss = float(ev['f'].sum())                            # :74
            flat = _clip(ev['g']).reshape(-1)                      # :76
            if trace is not None:
                trace.append(dict(rows=list(rows), T=Tg.clone(), loss=loss, grad=flat.clone()))
            return loss, flat

        loss, flat_grad = closure()
        current_evals = 1
        n_iter = 0
        stop = 0
        if bool(flat_grad.abs().max() <= tolerance_grad):
            stop = 1                                               # optimal at start
        else:
            d = t = prev_flat_grad = prev_loss = None
            old_dirs, old_stps, ro = [], [], []
            H_diag = 1.0
            while n_iter < iters:
                n_iter += 1
                if n_iter == 1:
                    d = flat_grad.neg()
                    old_dirs, old_stps, ro, H_diag = [], [], [], 1.0
                else:
                    y = flat_grad.sub(prev_flat_grad)
                    s = d.mul(t)
                    ys = float(y.dot(s))
                    if ys > 1e-10:
                        if len(old_dirs) == history_size:
                            old_dirs.pop(0); old_stps.pop(0); ro.pop(0)
                        old_dirs.append(y); old_stps.append(s); ro.append(1.0 / ys)
                        H_diag = ys / float(y.dot(y))
                    num_old = len(old_dirs)
                    al = [None] * num_old
                    q = flat_grad.neg()
                    for i in range(num_old - 1, -1, -1):
                        al[i] = float(old_stps[i].dot(q)) * ro[i]
                        q = q - al[i] * old_dirs[i]
                    d = r = q * H_diag
                    for i in range(num_old):
                        be_i = float(old_dirs[i].dot(r)) * ro[i]
                        r = r + (al[i] - be_i) * old_stps[i]
                    d = r
                prev_flat_grad = flat_grad.clone()
                prev_loss = loss
                if n_iter == 1:
                    t = min(1.0, 1.0 / float(flat_grad.abs().sum())) * lr
                else:
                    t = lr
                gtd = float(flat_grad.dot(d))
                if gtd > -tolerance_change:
                    stop = 2
                    break
                # LieGroupParameter.add_: group <- exp(t*d) * group
                Tg = _se3.se3_mul(_se3.se3_exp((t * d).reshape(len(rows), 6)), Tg)
                ls_evals = 0
                opt_cond = False
                if n_iter != iters:
                    loss, flat_grad = closure()
                    opt_cond = bool(flat_grad.abs().max() <= tolerance_grad)
                    ls_evals = 1
                current_evals += ls_evals
                if n_iter == iters:
                    stop = 3
                    break
                if current_evals >= max_eval:
                    stop = 4
                    break
                if opt_cond:
                    stop = 5
                    break
                if bool((d * t).abs().max() <= tolerance_change):
                    stop = 6
                    break
                if abs(loss - prev_loss) < tolerance_change:
                    stop = 7
                    break
        T[rows] = Tg
        n_iter_out[rows] = n_iter
        evals_out[rows] = current_evals
        stop_out[rows] = stop
    return T, dict(n_iter=n_iter_out, evals=evals_out, stop=stop_out)


def gn_solve(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight, iters, tolerance_change=1e-9, trace=None):
    """Gauss-Newton on the same objective: H delta = -g (Cholesky, f64), T <- exp(delta) * T.

    Stops a row when max|delta| <= tolerance_change (stop=6), when H is not positive definite
    (stop=8, pose left unchanged) or after ``iters`` iterations (stop=3).  n independent rows.
    """
    P = _prep(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight)
    n = P['n']
    T = torch.zeros(n, 7, dtype=F64)
    T[:, 6] = 1.0
    n_iter = np.zeros(n, dtype=np.int64)
    stop = np.zeros(n, dtype=np.int64)
    active = np.ones(n, dtype=bool)
    for it in range(iters):
        ev = evaluate(P, T, need_hessian=True)
        if trace is not None:
            trace.append(dict(T=T.clone(), loss=ev['f'].clone(), grad=ev['g'].clone(), H=ev['H'].clone()))
        for i in range(n):
            if not active[i]:
                continue
            n_iter[i] += 1
            H = ev['H'][i]
            g = ev['g'][i]
            L, info = torch.linalg.cholesky_ex(H)
            if int(info) != 0 or not bool(torch.isfinite(g).all()):
                stop[i] = 8
                active[i] = False
                continue
            delta = -torch.cholesky_solve(g[:, None], L)[:, 0]
            T[i] = _se3.se3_mul(_se3.se3_exp(delta[None]), T[i][None])[0]
            if bool(delta.abs().max() <= tolerance_change):
                stop[i] = 6
                active[i] = False
        if not active.any():
            break
    stop[active] = 3
    return T, dict(n_iter=n_iter, stop=stop)


def declarative_forward(T):
    """DeclarativeFunctionLie.forward outputs (declerative_node_lie.py:233-234): vec7, log6 as f32 (n,1,*)."""
    T = T.reshape(-1, 1, 7)
    return T.float(), _se3.se3_log(T).float()
