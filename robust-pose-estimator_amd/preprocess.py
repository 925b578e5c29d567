"""Input side on the GPU: counterparts of the reference's per-frame CPU preprocessing, same names and arguments.

  mask_specularities(img, mask=None, spec_thr=0.96)   dataset/stereo_dataset.py:12-16
  ResizeStereo(size)(left, right, mask)               dataset/transforms.py:20-39

Images may be passed as the reference passes them (float32 (3,H,W) tensors) or as decoded (uint8 (H,W,3)), which fuses
the `.permute(2,0,1).float()` of dataset/stereo_dataset.py:35-37 into the resize.  Everything runs in librpe_hip.so
(rpe_mask_specularities, rpe_resize_crop, rpe_resize_crop_mask); tensors must be on the GPU.
"""
import math

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


def _gpu(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise _lib.RpeError(f'{name}: expected a tensor on the GPU (the HIP path has no CPU fallback)')
    return t.contiguous()


def mask_specularities(img, mask=None, spec_thr=0.96):
    """img: (H,W,3) uint8 RGB as decoded; mask: (H,W) bool/uint8 or None.  Returns the (H,W) uint8 mask cv2.erode gives."""
    img = _gpu(img, 'img')
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
        raise _lib.RpeError('mask_specularities: img must be (H,W,3) uint8')
    h, w, _ = img.shape
    m = None
    if mask is not None:
        m = _gpu(mask, 'mask')
        m = m.view(torch.uint8) if m.dtype == torch.bool else m
        if m.dtype != torch.uint8 or tuple(m.shape) != (h, w):
            raise _lib.RpeError('mask_specularities: mask must be (H,W) bool/uint8')
    out = torch.empty(h, w, dtype=torch.uint8, device=img.device)
    thr = math.ceil(3 * 255 * spec_thr)              # integer sum < float threshold  <=>  sum < ceil(threshold)
    check(lib().rpe_mask_specularities(ptr(img), ptr(m), h, w, thr, ptr(out), stream_ptr()), 'rpe_mask_specularities')
    return out


class ResizeStereo:
    def __init__(self, size):
        self.size = [int(size[1]), int(size[0])]

    def __call__(self, left, right, mask=None):
        # resize with cropping to conserve aspect ratio (transforms.py:25-33)
        h, w = (left.shape[0], left.shape[1]) if left.dtype == torch.uint8 else left.shape[-2:]
        scale = max(self.size[0] / h, self.size[1] / w)
        size = [int(scale * h), int(scale * w)]
        return self._resize_with_crop(left, size), self._resize_with_crop(right, size), self._resize_with_crop(mask, size, nearest=True)

    def _resize_with_crop(self, img, size, nearest=False):
        if img is None:
            return None
        th, tw = self.size
        if size[0] < th or size[1] < tw:
            raise _lib.RpeError('ResizeStereo: resized image smaller than the crop (torchvision would zero-pad; not supported)')
        top, left = int(round((size[0] - th) / 2.0)), int(round((size[1] - tw) / 2.0))
        img = _gpu(img, 'image')
        if nearest:
            m = img.view(torch.uint8) if img.dtype == torch.bool else img
            if m.dtype != torch.uint8 or m.dim() != 3 or m.shape[0] != 1:
                raise _lib.RpeError('ResizeStereo: mask must be (1,H,W) bool/uint8')
            out = torch.empty(1, th, tw, dtype=torch.uint8, device=m.device)
            check(lib().rpe_resize_crop_mask(ptr(m), m.shape[1], m.shape[2], size[0], size[1], top, left, th, tw, ptr(out), stream_ptr()),
                  'rpe_resize_crop_mask')
            return out.view(torch.bool) if img.dtype == torch.bool else out
        if img.dtype == torch.uint8:                                   # decoded (H,W,C)
            h, w, c = img.shape
            u8 = 1
        elif img.dtype == torch.float32 and img.dim() == 3:            # (C,H,W) as the reference passes it
            c, h, w = img.shape
            u8 = 0
        else:
            raise _lib.RpeError('ResizeStereo: image must be (C,H,W) float32 or (H,W,C) uint8')
        out = torch.empty(c, th, tw, dtype=torch.float32, device=img.device)
        check(lib().rpe_resize_crop(ptr(img), u8, c, h, w, size[0], size[1], top, left, th, tw, ptr(out), stream_ptr()), 'rpe_resize_crop')
        return out


# ------------------------------------------------------------------------------------------------- rectification
# Counterpart of dataset/rectification.py:11-77 (StereoRectifier) and dataset/preprocess/stereo_rectify.py:5-51.  The per-frame
# work (two cv2.remap calls with nearest sampling) runs on the GPU (rpe_remap_nearest); the once-per-calibration work -- cv2's
# stereoRectify (Bouguet's algorithm, alpha = 0) and initUndistortRectifyMap -- is host arithmetic in f64 like cv2's, restated
# here from the published algorithm (OpenCV 4.x calib3d; cv2 is not in this image, so this part is UNPINNED against cv2
# itself -- tests check it against first principles: rectified rows coincide, maps invert the camera model).
import numpy as np      # noqa: E402  (host-side calibration arithmetic only)


def rodrigues(v):
    """cv2.Rodrigues: rotation vector (3,) -> matrix (3,3), or matrix -> vector."""
    v = np.asarray(v, np.float64)
    if v.size == 3:
        v = v.reshape(3)
        th = np.linalg.norm(v)
        if th < np.finfo(np.float64).eps:
            return np.eye(3)
        k = v / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx
    R = v.reshape(3, 3)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s, c = np.sqrt((r * r).sum() * 0.25), np.clip((np.trace(R) - 1) * 0.5, -1.0, 1.0)
    th = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (np.diag(R) + 1) * 0.5
        r = np.sqrt(np.maximum(t, 0))
        if R[0, 1] < 0: r[1] = -r[1]
        if R[0, 2] < 0: r[2] = -r[2]
        if abs(r[0]) < abs(r[1]) and abs(r[0]) < abs(r[2]) and (R[1, 2] > 0) != (r[1] * r[2] > 0): r[2] = -r[2]
        return r * (th / np.linalg.norm(r))
    return r * (0.5 / s) * th


def _dist8(d):
    """Distortion coefficients (k1, k2, p1, p2[, k3[, k4, k5, k6]]) padded to 8."""
    d = np.zeros(0) if d is None else np.asarray(d, np.float64).reshape(-1)
    if d.size > 8:
        if np.any(d[8:] != 0):
            raise NotImplementedError('thin-prism / tilt distortion terms')
        d = d[:8]
    return np.concatenate((d, np.zeros(8 - d.size)))


def distort_normalized(x, y, d):
    """OpenCV's lens model on normalised coordinates: radial (rational) + tangential."""
    k1, k2, p1, p2, k3, k4, k5, k6 = _dist8(d)
    r2 = x * x + y * y
    kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
    return x * kr + p1 * 2 * x * y + p2 * (r2 + 2 * x * x), y * kr + p1 * (r2 + 2 * y * y) + p2 * 2 * x * y


def undistort_points(pts, K, d, R=None, P=None, iters=5):
    """cvUndistortPoints: pixel coordinates (n,2) -> ideal coordinates, five fixed-point iterations of the inverse lens model,
    then the rotation R and the new camera matrix P."""
    k1, k2, p1, p2, k3, k4, k5, k6 = _dist8(d)
    pts = np.asarray(pts, np.float64)
    x0, y0 = (pts[:, 0] - K[0, 2]) / K[0, 0], (pts[:, 1] - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    if d is not None and np.any(_dist8(d) != 0):
        for _ in range(iters):
            r2 = x * x + y * y
            icd = (1 + ((k6 * r2 + k5) * r2 + k4) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
            dx, dy = 2 * p1 * x * y + p2 * (r2 + 2 * x * x), p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x, y = (x0 - dx) * icd, (y0 - dy) * icd
    RR = np.eye(3) if R is None else np.asarray(R, np.float64)
    if P is not None:
        RR = np.asarray(P, np.float64)[:3, :3] @ RR
    X = RR @ np.stack((x, y, np.ones_like(x)))
    return np.stack((X[0] / X[2], X[1] / X[2]), 1)


def _rectangles(K, d, R, P, size):
    """icvGetRectangles: inscribed / circumscribed rectangle of the rectified image of a 9x9 grid over the source image."""
    n = 9
    gx, gy = np.meshgrid(np.arange(n), np.arange(n))
    pts = np.stack((gx.reshape(-1).astype(np.float32) * size[0] / (n - 1), gy.reshape(-1).astype(np.float32) * size[1] / (n - 1)), 1)
    p = undistort_points(pts.astype(np.float32), K, d, R, P).astype(np.float32).reshape(n, n, 2)      # (cv2 keeps these points in f32)
    inner = (p[:, 0, 0].max(), p[0, :, 1].max(), p[:, n - 1, 0].min(), p[n - 1, :, 1].min())             # x0, y0, x1, y1
    outer = (p[..., 0].min(), p[..., 1].min(), p[..., 0].max(), p[..., 1].max())
    f = np.float32
    return (inner[0], inner[1], f(inner[2] - inner[0]), f(inner[3] - inner[1])), (outer[0], outer[1], f(outer[2] - outer[0]), f(outer[3] - outer[1]))


def stereo_rectify(K1, d1, K2, d2, size, R, T, alpha=0.0):
    """cv2.stereoRectify(..., flags=CALIB_ZERO_DISPARITY, alpha) -> R1, R2, P1, P2 (Bouguet: both cameras rotated by half of R,
    then the common rotation that brings the baseline onto the x (or y) axis; common focal length; alpha-scaled so that
    alpha = 0 keeps only valid pixels).  size = (width, height)."""
    K1, K2, R = np.asarray(K1, np.float64), np.asarray(K2, np.float64), np.asarray(R, np.float64)
    T = np.asarray(T, np.float64).reshape(3)
    nx, ny = float(size[0]), float(size[1])
    r_r = rodrigues(rodrigues(R) * -0.5)
    t = r_r @ T
    idx = 0 if abs(t[0]) > abs(t[1]) else 1
    c, nt = t[idx], np.linalg.norm(t)
    uu = np.zeros(3); uu[idx] = 1.0 if c > 0 else -1.0
    ww = np.cross(t, uu)
    nw = np.linalg.norm(ww)
    if nw > 0:
        ww = ww * (np.arccos(abs(c) / nt) / nw)
    wR = rodrigues(ww)
    R1, R2 = wR @ r_r.T, wR @ r_r
    t = R2 @ T
    fc_new = np.inf
    for K, d in ((K1, d1), (K2, d2)):
        dk1 = _dist8(d)[0]
        fc = K[idx ^ 1, idx ^ 1]
        if dk1 < 0:
            fc *= 1 + dk1 * (nx * nx + ny * ny) / (4 * fc * fc)
        fc_new = min(fc_new, fc)
    cc = []
    for K, d, Rk in ((K1, d1, R1), (K2, d2, R2)):
        corners = np.array([[0, 0], [nx - 1, 0], [0, ny - 1], [nx - 1, ny - 1]], np.float32)
        p = undistort_points(corners, K, d).astype(np.float32).astype(np.float64)                       # (f32 points, as cv2)
        X = Rk @ np.stack((p[:, 0], p[:, 1], np.ones(4)))
        proj = np.stack((fc_new * X[0] / X[2], fc_new * X[1] / X[2]), 1).astype(np.float32).astype(np.float64)
        cc.append(((nx - 1) / 2 - proj[:, 0].mean(), (ny - 1) / 2 - proj[:, 1].mean()))
    cx = (cc[0][0] + cc[1][0]) * 0.5                                                                    # CALIB_ZERO_DISPARITY
    cy = (cc[0][1] + cc[1][1]) * 0.5
    P1 = np.zeros((3, 4)); P2 = np.zeros((3, 4))
    for P in (P1, P2):
        P[0, 0] = P[1, 1] = fc_new; P[0, 2] = cx; P[1, 2] = cy; P[2, 2] = 1.0
    P2[idx, 3] = t[idx] * fc_new
    if alpha is not None and alpha >= 0:
        alpha = min(alpha, 1.0)
        (i1, o1), (i2, o2) = _rectangles(K1, d1, R1, P1, size), _rectangles(K2, d2, R2, P2, size)

        def ratios(r, c0x, c0y):
            return (c0x / (c0x - r[0]), c0y / (c0y - r[1]), (nx - c0x) / (r[0] + r[2] - c0x), (ny - c0y) / (r[1] + r[3] - c0y))
        s0 = max(ratios(i1, cx, cy) + ratios(i2, cx, cy))
        s1 = min(ratios(o1, cx, cy) + ratios(o2, cx, cy))
        s = s0 * (1 - alpha) + s1 * alpha
        fc_new *= s
        for P in (P1, P2):
            P[0, 0] = P[1, 1] = fc_new
        P2[idx, 3] *= s
    return R1, R2, P1, P2


def init_undistort_rectify_map(K, d, R, P, size):
    """cv2.initUndistortRectifyMap(K, d, R, P, size, CV_32FC1): for every pixel of the rectified image the position in the
    source image, (map_x, map_y) float32 (height, width)."""
    K = np.asarray(K, np.float64)
    iR = np.linalg.inv(np.asarray(P, np.float64)[:3, :3] @ np.asarray(R, np.float64))
    w, h = int(size[0]), int(size[1])
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    X = iR[:, 0, None, None] * u + iR[:, 1, None, None] * v + iR[:, 2, None, None]
    xd, yd = distort_normalized(X[0] / X[2], X[1] / X[2], d)
    return (K[0, 0] * xd + K[0, 2]).astype(np.float32), (K[1, 1] * yd + K[1, 2]).astype(np.float32)


def get_rect_maps(lcam_mat, rcam_mat, rmat, tvec, ldist_coeffs, rdist_coeffs, img_size=(1280, 1024), triangular_intrinsics=False,
                  mode='conventional'):
    """dataset/preprocess/stereo_rectify.py:5-41, same arguments and return value (maps dict, P1, P2).  As there, the RIGHT
    map is built with the LEFT camera's distortion coefficients (stereo_rectify.py:29)."""
    if mode == 'pseudo':
        return {}, np.asarray(lcam_mat, np.float64), np.asarray(rcam_mat, np.float64)
    if mode != 'conventional':
        raise NotImplementedError
    lcam_mat, rcam_mat = np.asarray(lcam_mat, np.float64), np.asarray(rcam_mat, np.float64)
    if triangular_intrinsics:
        lcam_mat = np.array([[lcam_mat[0, 0], 0, lcam_mat[0, 2]], [0, lcam_mat[1, 1], lcam_mat[1, 2]], [0, 0, 1]], np.float64)
        rcam_mat = np.array([[rcam_mat[0, 0], 0, rcam_mat[0, 2]], [0, rcam_mat[1, 1], rcam_mat[1, 2]], [0, 0, 1]], np.float64)
    r1, r2, p1, p2 = stereo_rectify(lcam_mat, ldist_coeffs, rcam_mat, rdist_coeffs, tuple(img_size), rmat, np.asarray(tvec).reshape(3), alpha=0)
    lmap1, lmap2 = init_undistort_rectify_map(lcam_mat, ldist_coeffs, r1, p1, tuple(img_size))
    rmap1, rmap2 = init_undistort_rectify_map(rcam_mat, ldist_coeffs, r2, p2, tuple(img_size))
    return {'lmap1': lmap1, 'lmap2': lmap2, 'rmap1': rmap1, 'rmap2': rmap2}, p1, p2


def remap_nearest(img, mapx, mapy):
    """cv2.remap(img, mapx, mapy, INTER_NEAREST) on the GPU: img (C,H,W) uint8 / float32, maps (h,w) float32 GPU tensors."""
    img = _gpu(img, 'img')
    if img.dim() != 3 or img.dtype not in (torch.uint8, torch.float32):
        raise _lib.RpeError('remap_nearest: img must be (C,H,W) uint8 or float32')
    mapx, mapy = _gpu(mapx, 'mapx'), _gpu(mapy, 'mapy')
    if mapx.dtype != torch.float32 or mapy.dtype != torch.float32 or mapx.dim() != 2 or mapx.shape != mapy.shape:
        raise _lib.RpeError('remap_nearest: maps must be two (h,w) float32 tensors')
    c, h, w = img.shape
    oh, ow = mapx.shape
    out = torch.empty(c, oh, ow, dtype=img.dtype, device=img.device)
    check(lib().rpe_remap_nearest(ptr(img), int(img.dtype == torch.uint8), c, h, w, ptr(mapx), ptr(mapy), oh, ow, ptr(out), stream_ptr()),
          'rpe_remap_nearest')
    return out


def shift_bilinear(img, tx, ty):
    """cv2.warpAffine(img, [[1,0,tx],[0,1,ty]], (w,h)) on the GPU (rpe_shift_bilinear: INTER_LINEAR in OpenCV's 1/32-pixel fixed
    point, constant-0 border): img (C,H,W) uint8 / float32.  The pseudo-rectification of dataset/preprocess/stereo_rectify.py:52-59."""
    img = _gpu(img, 'img')
    if img.dim() != 3 or img.dtype not in (torch.uint8, torch.float32):
        raise _lib.RpeError('shift_bilinear: img must be (C,H,W) uint8 or float32')
    c, h, w = img.shape
    out = torch.empty_like(img)
    check(lib().rpe_shift_bilinear(ptr(img), int(img.dtype == torch.uint8), c, h, w, float(np.float32(tx)), float(np.float32(ty)), ptr(out),
                                   stream_ptr()), 'rpe_shift_bilinear')
    return out


class StereoRectifier:
    """dataset/rectification.py:11-77, same constructor, call and get_rectified_calib().  Images are (3,H,W) GPU tensors (uint8
    or float32) and stay on the GPU; `calib_file` may also be an already-loaded calibration dict (keys lkmat, rkmat, ld, rd, R,
    T, img_size).  mode='pseudo' (configuration/infer_scared.yaml:17): the left image passes through, the right one is shifted by
    the difference of the principal points (cv2.warpAffine -> rpe_shift_bilinear)."""

    def __init__(self, calib_file, img_size_new=None, mode='conventional', device='cuda'):
        import os
        if isinstance(calib_file, dict):
            cal = {k: (np.array(v, np.float64) if k != 'img_size' else tuple(v)) for k, v in calib_file.items()}
        else:
            ext = os.path.splitext(calib_file)[1]
            loaders = {'.json': self._load_calib_json, '.ini': self._load_calib_ini, '.yaml': self._load_calib_yaml}
            if ext not in loaders:
                raise NotImplementedError
            cal = loaders[ext](calib_file)
        assert mode in ['conventional', 'pseudo']
        self.mode = mode
        self.scale = 1.0
        if img_size_new is not None:
            # scale intrinsics (rectification.py:27-37)
            self.scale = img_size_new[0] / cal['img_size'][0]
            h_crop = int((cal['img_size'][1] * self.scale - img_size_new[1]) / 2)
            assert h_crop >= 0, 'only vertical crop implemented'
            cal['lkmat'][:2] *= self.scale
            cal['rkmat'][:2] *= self.scale
            cal['lkmat'][1, 2] -= h_crop
            cal['rkmat'][1, 2] -= h_crop
            cal['img_size'] = img_size_new
        self.img_size = cal['img_size']
        self.cal = cal
        self.maps, self.l_intr, self.r_intr = get_rect_maps(lcam_mat=cal['lkmat'], rcam_mat=cal['rkmat'], rmat=cal['R'], tvec=cal['T'],
                                                            ldist_coeffs=cal['ld'], rdist_coeffs=cal['rd'], img_size=cal['img_size'], mode=self.mode)
        self._gpu_maps = {k: torch.from_numpy(v).to(device) for k, v in self.maps.items()} if torch.cuda.is_available() else None

    def __call__(self, img_left, img_right):
        if self.mode == 'pseudo':                              # rectification.py:55-58
            lk, rk = self.cal['lkmat'], self.cal['rkmat']
            return _gpu(img_left, 'img_left'), shift_bilinear(img_right, lk[0][-1] - rk[0][-1], lk[1][-1] - rk[1][-1])
        m = self._gpu_maps
        if m is None:
            raise _lib.RpeError('StereoRectifier: no GPU (the HIP path has no CPU fallback)')
        return remap_nearest(img_left, m['lmap1'], m['lmap2']), remap_nearest(img_right, m['rmap1'], m['rmap2'])

    def get_rectified_calib(self):
        calib = {'intrinsics': {'left': self.l_intr[:3, :3], 'right': self.r_intr[:3, :3]}, 'extrinsics': np.eye(4)}
        if self.mode == 'conventional':
            calib['extrinsics'][:3, 3] = np.array([self.r_intr[0, 3] / self.r_intr[0, 0], 0., 0.])   # Tx*f (rectification.py:71-72)
        else:
            calib['extrinsics'][:3, 3] = np.asarray(self.cal['T'], np.float64).reshape(3)            # rectification.py:73-74
        calib['bf'] = np.sqrt(np.sum(calib['extrinsics'][:3, 3] ** 2)) * self.l_intr[0, 0]
        calib['bf_orig'] = calib['bf'] / self.scale
        calib['img_size'] = self.img_size
        return calib

    @staticmethod
    def _load_calib_json(fname):
        import json
        with open(fname, 'rb') as f:
            data = json.load(f)['data']
        lk, rk = np.eye(3), np.eye(3)
        for k, i in ((lk, 0), (rk, 1)):
            k[0, 0], k[1, 1] = data['intrinsics'][i]['f']
            k[:2, -1] = data['intrinsics'][i]['c']
        return {'lkmat': lk, 'rkmat': rk, 'ld': np.array(data['intrinsics'][0]['k'], np.float64), 'rd': np.array(data['intrinsics'][1]['k'], np.float64),
                'T': np.array(data['extrinsics']['T'], np.float64), 'R': rodrigues(np.array(data['extrinsics']['om'], np.float64)),
                'img_size': (data['width'], data['height'])}

    @staticmethod
    def _load_calib_ini(fname):
        import configparser
        cfg = configparser.ConfigParser()
        cfg.read(fname)
        L, Rr = cfg['StereoLeft'], cfg['StereoRight']

        def kmat(s):
            k = np.eye(3)
            k[0, 0], k[1, 1], k[0, 2], k[1, 2] = float(s['fc_x']), float(s['fc_y']), float(s['cc_x']), float(s['cc_y'])
            return k
        return {'lkmat': kmat(L), 'rkmat': kmat(Rr), 'ld': np.array([float(L['kc_%d' % i]) for i in range(8)]),
                'rd': np.array([float(Rr['kc_%d' % i]) for i in range(8)]), 'T': np.array([float(Rr['T_%d' % i]) for i in range(3)]),
                'R': np.array([float(Rr['R_%d' % i]) for i in range(9)]).reshape(3, 3), 'img_size': (float(L['res_x']), float(L['res_y']))}

    @staticmethod
    def _load_calib_yaml(fname):
        """OpenCV FileStorage YAML (%YAML:1.0 header, !!opencv-matrix nodes) without cv2."""
        import yaml

        class _L(yaml.SafeLoader):
            pass
        _L.add_constructor('tag:yaml.org,2002:opencv-matrix',
                           lambda ld, node: (lambda m: np.array(m['data'], np.float64).reshape(m['rows'], m['cols']))(ld.construct_mapping(node, deep=True)))
        with open(fname) as f:
            text = f.read()
        if text.startswith('%YAML:1.0'):
            text = '%YAML 1.1' + text[len('%YAML:1.0'):]
        fs = yaml.load(text, Loader=_L)
        return {'lkmat': fs['M1'], 'rkmat': fs['M2'], 'ld': fs['D1'], 'rd': fs['D2'], 'T': fs['T'], 'R': fs['R'],
                'img_size': (int(fs['Camera.width']), int(fs['Camera.height']))}
