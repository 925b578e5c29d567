// SE(3) device functions, templated on float/double.
//
// Restates the published lietorch algorithm (princeton-vl/lietorch include/so3.h, se3.h; not on disk,
// unpinned dependency of the reference, README.md:37): unit-quaternion exp/log with Taylor guards at
// theta^2 < 1e-6, translation through the SO(3) left Jacobian.  Pose layout [tx ty tz qx qy qz qw],
// tangent [tau phi].  Mirrors oracle/se3.py operation for operation.
#pragma once
#include <hip/hip_runtime.h>

#define RPE_SE3_EPS 1e-6

template <typename S> struct V3 { S x, y, z; };

template <typename S> __device__ __forceinline__ V3<S> v3(S x, S y, S z) { V3<S> r; r.x = x; r.y = y; r.z = z; return r; }
template <typename S> __device__ __forceinline__ V3<S> cross(const V3<S>& a, const V3<S>& b) {
    return v3<S>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <typename S> __device__ __forceinline__ V3<S> add(const V3<S>& a, const V3<S>& b) { return v3<S>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename S> __device__ __forceinline__ V3<S> scale(const V3<S>& a, S s) { return v3<S>(a.x * s, a.y * s, a.z * s); }
template <typename S> __device__ __forceinline__ S dot(const V3<S>& a, const V3<S>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

template <typename S> struct Pose { V3<S> t; V3<S> qv; S qw; };

template <typename S> __device__ __forceinline__ Pose<S> pose_load(const S* p) {
    Pose<S> P; P.t = v3<S>(p[0], p[1], p[2]); P.qv = v3<S>(p[3], p[4], p[5]); P.qw = p[6]; return P;
}
template <typename S> __device__ __forceinline__ void pose_store(S* p, const Pose<S>& P) {
    p[0] = P.t.x; p[1] = P.t.y; p[2] = P.t.z; p[3] = P.qv.x; p[4] = P.qv.y; p[5] = P.qv.z; p[6] = P.qw;
}
template <typename S> __device__ __forceinline__ Pose<S> pose_identity() {
    Pose<S> P; P.t = v3<S>(0, 0, 0); P.qv = v3<S>(0, 0, 0); P.qw = 1; return P;
}

// R p as p + w*uv + v x uv, uv = 2 v x p
template <typename S> __device__ __forceinline__ V3<S> quat_rotate(const V3<S>& qv, S qw, const V3<S>& p) {
    V3<S> uv = scale(cross(qv, p), (S)2);
    return add(add(p, scale(uv, qw)), cross(qv, uv));
}

template <typename S> __device__ __forceinline__ V3<S> se3_act(const Pose<S>& T, const V3<S>& p) {
    return add(quat_rotate(T.qv, T.qw, p), T.t);
}

template <typename S> __device__ __forceinline__ void so3_exp(const V3<S>& phi, V3<S>& qv, S& qw) {
    S th2 = dot(phi, phi);
    S imag, real;
    if (th2 < (S)RPE_SE3_EPS) {
        S th4 = th2 * th2;
        imag = (S)0.5 - th2 / (S)48.0 + th4 / (S)3840.0;
        real = (S)1.0 - th2 / (S)8.0 + th4 / (S)384.0;
    } else {
        S th = sqrt(th2);
        imag = sin((S)0.5 * th) / th;
        real = cos((S)0.5 * th);
    }
    qv = scale(phi, imag);
    qw = real;
}

template <typename S> __device__ __forceinline__ V3<S> so3_log(const V3<S>& qv, S qw) {
    const S PI = (S)3.14159265358979323846;
    S sq = dot(qv, qv);
    S coef;
    if (sq < (S)(RPE_SE3_EPS * RPE_SE3_EPS)) {
        coef = (S)2.0 / qw - ((S)2.0 / (S)3.0) * sq / (qw * qw * qw);
    } else {
        S n = sqrt(sq);
        S aw = qw < 0 ? -qw : qw;
        if (aw < (S)RPE_SE3_EPS) coef = qw > 0 ? PI / n : -PI / n;
        else coef = (S)2.0 * atan(n / qw) / n;
    }
    return scale(qv, coef);
}

// y = (I + c1 [phi]x + c2 [phi]x^2) v
template <typename S> __device__ __forceinline__ V3<S> apply_poly(const V3<S>& phi, S c1, S c2, const V3<S>& v) {
    V3<S> pv = cross(phi, v);
    V3<S> ppv = cross(phi, pv);
    return add(add(v, scale(pv, c1)), scale(ppv, c2));
}

template <typename S> __device__ __forceinline__ V3<S> left_jacobian_mul(const V3<S>& phi, const V3<S>& v) {
    S th2 = dot(phi, phi);
    S c1, c2;
    if (th2 < (S)RPE_SE3_EPS) {
        c1 = (S)0.5 - th2 / (S)24.0;
        c2 = (S)1.0 / (S)6.0 - th2 / (S)120.0;
    } else {
        S th = sqrt(th2);
        c1 = ((S)1.0 - cos(th)) / th2;
        c2 = (th - sin(th)) / (th2 * th);
    }
    return apply_poly(phi, c1, c2, v);
}

template <typename S> __device__ __forceinline__ V3<S> left_jacobian_inv_mul(const V3<S>& phi, const V3<S>& v) {
    S th2 = dot(phi, phi);
    S c2;
    if (th2 < (S)RPE_SE3_EPS) {
        c2 = (S)1.0 / (S)12.0 + th2 / (S)720.0;
    } else {
        S th = sqrt(th2);
        S half = (S)0.5 * th;
        c2 = ((S)1.0 - th * cos(half) / ((S)2.0 * sin(half))) / th2;
    }
    return apply_poly(phi, (S)-0.5, c2, v);
}

template <typename S> __device__ __forceinline__ Pose<S> se3_exp(const V3<S>& tau, const V3<S>& phi) {
    Pose<S> T;
    so3_exp(phi, T.qv, T.qw);
    T.t = left_jacobian_mul(phi, tau);
    return T;
}

template <typename S> __device__ __forceinline__ void se3_log(const Pose<S>& T, V3<S>& tau, V3<S>& phi) {
    phi = so3_log(T.qv, T.qw);
    tau = left_jacobian_inv_mul(phi, T.t);
}

template <typename S> __device__ __forceinline__ Pose<S> se3_mul(const Pose<S>& A, const Pose<S>& B) {
    Pose<S> C;
    C.t = add(A.t, quat_rotate(A.qv, A.qw, B.t));
    S x1 = A.qv.x, y1 = A.qv.y, z1 = A.qv.z, w1 = A.qw;
    S x2 = B.qv.x, y2 = B.qv.y, z2 = B.qv.z, w2 = B.qw;
    S x = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
    S y = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2;
    S z = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2;
    S w = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
    S nrm = sqrt(x * x + y * y + z * z + w * w);
    C.qv = v3<S>(x / nrm, y / nrm, z / nrm);
    C.qw = w / nrm;
    return C;
}

template <typename S> __device__ __forceinline__ Pose<S> se3_inv(const Pose<S>& T) {
    Pose<S> R;
    R.qv = v3<S>(-T.qv.x, -T.qv.y, -T.qv.z);
    R.qw = T.qw;
    V3<S> rt = quat_rotate(R.qv, R.qw, T.t);
    R.t = v3<S>(-rt.x, -rt.y, -rt.z);
    return R;
}
