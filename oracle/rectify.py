"""TEST INFRASTRUCTURE (CPU oracle) -- restatement of the reference's stereo rectification, one pixel / one point at a time.

Follows dataset/rectification.py:11-77 (StereoRectifier) and dataset/preprocess/stereo_rectify.py:5-51 (get_rect_maps,
rectify_pair).  Everything there is cv2: stereoRectify(alpha=0), initUndistortRectifyMap(CV_32FC1), remap(INTER_NEAREST).
PARITY UNPINNED: cv2 is not installed in this image and the reference holds no calibration fixture or rectified golden
image, so neither cv2 nor the reference can produce vectors here.  What this file restates is the published arithmetic
(OpenCV 4.x calib3d: Bouguet's rectification, the rational + tangential lens model, cvRound = round half to even, constant-0
border); the tests pin it to first principles instead -- the maps invert the forward camera model, rectified rows of a 3-D
point coincide in both views, the disparity is bf / z -- and compare the product's vectorised host code and its GPU gather
against these scalar loops.
"""
import math

import numpy as np


def rodrigues_vec_to_mat(v):
    th = math.sqrt(sum(float(a) * float(a) for a in v))
    if th < 2.3e-16:
        return np.eye(3)
    k = [float(a) / th for a in v]
    c, s = math.cos(th), math.sin(th)
    R = np.zeros((3, 3))
    kx = [[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]]
    for i in range(3):
        for j in range(3):
            R[i, j] = c * (1.0 if i == j else 0.0) + (1 - c) * k[i] * k[j] + s * kx[i][j]
    return R


def distort(x, y, d):
    """normalised ideal point -> normalised distorted point (k1, k2, p1, p2, k3, k4, k5, k6)."""
    d = list(d) + [0.0] * (8 - len(d))
    k1, k2, p1, p2, k3, k4, k5, k6 = d[:8]
    r2 = x * x + y * y
    kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
    return x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x), y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y


def project(X, K, d):
    """3-D point in the camera frame -> pixel, through the lens model (the forward model the maps must invert)."""
    xd, yd = distort(X[0] / X[2], X[1] / X[2], d)
    return K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]


def undistort_rectify_map(K, d, R, P, size):
    """initUndistortRectifyMap, pixel by pixel: rectified pixel (u, v) -> ray (P[:3,:3] R)^-1 (u, v, 1) -> lens model -> source
    pixel; float32 maps (h, w)."""
    iR = np.linalg.inv(np.asarray(P, np.float64)[:3, :3] @ np.asarray(R, np.float64))
    w, h = int(size[0]), int(size[1])
    mx, my = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
    for v in range(h):
        for u in range(w):
            X = iR @ np.array([u, v, 1.0])
            mx[v, u], my[v, u] = project(X, np.asarray(K, np.float64), d)
    return mx, my


def remap_nearest(img, mapx, mapy):
    """cv2.remap(img, mapx, mapy, INTER_NEAREST), constant-0 border: img (C,H,W)."""
    c, h, w = img.shape
    oh, ow = mapx.shape
    out = np.zeros((c, oh, ow), img.dtype)
    for y in range(oh):
        for x in range(ow):
            fx, fy = float(mapx[y, x]), float(mapy[y, x])
            if math.isnan(fx) or math.isnan(fy):
                continue
            sx, sy = round(min(max(fx, -32768.0), 32767.0)), round(min(max(fy, -32768.0), 32767.0))      # Python round = half to even
            if 0 <= sx < w and 0 <= sy < h:
                out[:, y, x] = img[:, sy, sx]
    return out


def warp_affine_shift(img, tx, ty):
    """cv2.warpAffine(img, [[1,0,tx],[0,1,ty]], (w,h)) = dataset/preprocess/stereo_rectify.py:52-59 pseudo_rectify_2d, one pixel at a
    time: img (C,H,W) uint8 or float32.  OpenCV 4.x imgwarp.cpp restated (PARITY UNPINNED like the rest of this file): the inverse
    map source = dst - t in fixed point (AB_BITS 10), rounded to 1/32 pixel; the 5-bit bilinear table (integer weights x 32 summing
    to 2^15 for uint8, float weights for float32); constant-0 border."""
    tx, ty = float(np.float32(tx)), float(np.float32(ty))
    c, h, w = img.shape
    out = np.zeros_like(img)
    rnd = lambda v: int(np.rint(v))                              # cvRound: half to even
    X0 = rnd(-tx * 1024.0) + 16
    for y in range(h):
        Y0 = rnd((y - ty) * 1024.0) + 16
        Y = Y0 >> 5
        sy, ay = max(-32768, min(32767, Y >> 5)), Y & 31
        for x in range(w):
            X = (X0 + x * 1024) >> 5
            sx, ax = max(-32768, min(32767, X >> 5)), X & 31

            def tap(yy, xx, ch):
                return img[ch, yy, xx] if 0 <= yy < h and 0 <= xx < w else img.dtype.type(0)
            for ch in range(c):
                p = [tap(sy, sx, ch), tap(sy, sx + 1, ch), tap(sy + 1, sx, ch), tap(sy + 1, sx + 1, ch)]
                if img.dtype == np.uint8:
                    wts = [(32 - ax) * (32 - ay) * 32, ax * (32 - ay) * 32, (32 - ax) * ay * 32, ax * ay * 32]
                    v = (sum(int(a) * int(b) for a, b in zip(wts, p)) + (1 << 14)) >> 15
                    out[ch, y, x] = min(255, max(0, v))
                else:
                    fx, fy = np.float32(ax) * np.float32(1 / 32), np.float32(ay) * np.float32(1 / 32)
                    one = np.float32(1)
                    wts = [(one - fy) * (one - fx), (one - fy) * fx, fy * (one - fx), fy * fx]
                    acc = np.float32(p[0]) * wts[0]
                    for a, b in zip(wts[1:], p[1:]):
                        acc = np.float32(acc + np.float32(np.float32(b) * a))
                    out[ch, y, x] = acc
    return out
