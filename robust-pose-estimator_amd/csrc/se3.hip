// Batched SE(3) group kernels behind rpe_se3_* (include/rpe.h).  One thread per group element (per point
// for act): these calls are tiny (tracker bookkeeping, tests); the solver uses se3_device.h inline.
#include "rpe_common.h"
#include "se3_device.h"

// ONE compiled body per group operation and scalar type, called (not inlined) by every kernel of this file: inlined, the same
// source rounds differently from kernel to kernel (which multiply the compiler contracts with which add depends on the code around
// it: measured, one ulp in a translation component between rpe_se3_mul and the same product inside a chaining loop), and the
// composite kernels (rpe_se3_chain, rpe_pose_gate_chain) must give the bits of the separate launches they stand for.
template <typename S> __device__ __attribute__((noinline)) Pose<S> op_mul(const Pose<S> A, const Pose<S> B) { return se3_mul(A, B); }
template <typename S> __device__ __attribute__((noinline)) Pose<S> op_inv(const Pose<S> T) { return se3_inv(T); }
template <typename S> __device__ __attribute__((noinline)) void op_log(const Pose<S> T, V3<S>& tau, V3<S>& phi) { se3_log(T, tau, phi); }

template <typename S> __global__ void k_exp(const S* xi, S* T, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const S* x = xi + i * 6;
    pose_store(T + i * 7, se3_exp(v3<S>(x[0], x[1], x[2]), v3<S>(x[3], x[4], x[5])));
}
template <typename S> __global__ void k_log(const S* T, S* xi, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3<S> tau, phi;
    op_log(pose_load(T + i * 7), tau, phi);
    S* x = xi + i * 6;
    x[0] = tau.x; x[1] = tau.y; x[2] = tau.z; x[3] = phi.x; x[4] = phi.y; x[5] = phi.z;
}
template <typename S> __global__ void k_mul(const S* A, const S* B, S* C, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    pose_store(C + i * 7, op_mul(pose_load(A + i * 7), pose_load(B + i * 7)));
}
template <typename S> __global__ void k_inv(const S* T, S* R, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    pose_store(R + i * 7, op_inv(pose_load(T + i * 7)));
}
template <typename S> __global__ void k_act(const S* T, const S* pts, S* out, int64_t n, int64_t m) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n * m) return;
    Pose<S> P = pose_load(T + (i / m) * 7);
    const S* p = pts + i * 3;
    V3<S> r = se3_act(P, v3<S>(p[0], p[1], p[2]));
    out[i * 3 + 0] = r.x; out[i * 3 + 1] = r.y; out[i * 3 + 2] = r.z;
}
// Serial prefix product; one thread (the dependency chain is inherently sequential and m is a sequence
// length, not a pixel count).
template <typename S> __global__ void k_chain(const S* rel, const S* init, S* out, int64_t m, S s) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Pose<S> P = init ? pose_load(init) : pose_identity<S>();
    for (int64_t k = 0; k < m; ++k) {
        Pose<S> r = pose_load(rel + k * 7);
        r.t = scale(r.t, s);                       // SE3.scale: translation * s (pose_estimator.py:90)
        P = op_mul(P, op_inv(r));                  // pose_estimator.py:91
        pose_store(out + k * 7, P);
    }
}

// The tracker's per-frame bookkeeping (core/pose/pose_estimator.py:81-91) for m consecutive relative poses in ONE launch: the
// failure gate isnan(rel) | any(|log(rel)| > thr) -> identity, then the chain of k_chain.  The same device functions in the same
// order as rpe_se3_log / _inv / _mul give (the dozen element-wise launches and two host synchronisations per frame this replaces
// computed exactly that), so the poses are bit-identical to the step-by-step form.
template <typename S> __global__ void k_gate_chain(const S* rel, const S* init, S* rel_out, S* abs_out, int32_t* ok, int64_t m, S s, S thr) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Pose<S> P = init ? pose_load(init) : pose_identity<S>();
    for (int64_t k = 0; k < m; ++k) {
        const S* rv = rel + k * 7;
        bool bad = false;
#pragma unroll
        for (int e = 0; e < 7; ++e) bad = bad || rv[e] != rv[e];
        Pose<S> r = pose_load(rv);
        V3<S> tau, phi;
        op_log(r, tau, phi);
        const S lg[6] = {tau.x, tau.y, tau.z, phi.x, phi.y, phi.z};
#pragma unroll
        for (int e = 0; e < 6; ++e) bad = bad || (lg[e] < 0 ? -lg[e] : lg[e]) > thr;        // (NaN compares false, like torch.abs(log) > thr)
        if (bad) r = pose_identity<S>();
        if (rel_out) pose_store(rel_out + k * 7, r);
        if (ok) ok[k] = bad ? 0 : 1;
        r.t = scale(r.t, s);                       // SE3.scale: translation * s (pose_estimator.py:90)
        P = op_mul(P, op_inv(r));                  // pose_estimator.py:91
        pose_store(abs_out + k * 7, P);
    }
}

extern "C" {

const char* rpe_version(void) { return "rpe-hip 0.1 gfx950"; }
int rpe_abi_version(void) { return RPE_ABI_VERSION; }
int rpe_abi_minor(void) { return RPE_ABI_MINOR; }

int rpe_se3_exp(const void* xi, void* T, int64_t n, int dtype, void* stream) {
    if (!xi || !T || n < 0) return RPE_E_BADARG;
    if (n == 0) return RPE_OK;
    hipStream_t st = (hipStream_t)stream;
    int blocks = ceil_div(n, 256);
    if (dtype == RPE_F32) hipLaunchKernelGGL(k_exp<float>, dim3(blocks), dim3(256), 0, st, (const float*)xi, (float*)T, n);
    else if (dtype == RPE_F64) hipLaunchKernelGGL(k_exp<double>, dim3(blocks), dim3(256), 0, st, (const double*)xi, (double*)T, n);
    else return RPE_E_BADARG;
    return rpe_check_launch();
}
int rpe_se3_log(const void* T, void* xi, int64_t n, int dtype, void* stream) {
    if (!xi || !T || n < 0) return RPE_E_BADARG;
    if (n == 0) return RPE_OK;
    hipStream_t st = (hipStream_t)stream;
    int blocks = ceil_div(n, 256);
    if (dtype == RPE_F32) hipLaunchKernelGGL(k_log<float>, dim3(blocks), dim3(256), 0, st, (const float*)T, (float*)xi, n);
    else if (dtype == RPE_F64) hipLaunchKernelGGL(k_log<double>, dim3(blocks), dim3(256), 0, st, (const double*)T, (double*)xi, n);
    else return RPE_E_BADARG;
    return rpe_check_launch();
}
int rpe_se3_mul(const void* A, const void* B, void* C, int64_t n, int dtype, void* stream) {
    if (!A || !B || !C || n < 0) return RPE_E_BADARG;
    if (n == 0) return RPE_OK;
    hipStream_t st = (hipStream_t)stream;
    int blocks = ceil_div(n, 256);
    if (dtype == RPE_F32) hipLaunchKernelGGL(k_mul<float>, dim3(blocks), dim3(256), 0, st, (const float*)A, (const float*)B, (float*)C, n);
    else if (dtype == RPE_F64) hipLaunchKernelGGL(k_mul<double>, dim3(blocks), dim3(256), 0, st, (const double*)A, (const double*)B, (double*)C, n);
    else return RPE_E_BADARG;
    return rpe_check_launch();
}
int rpe_se3_inv(const void* T, void* R, int64_t n, int dtype, void* stream) {
    if (!T || !R || n < 0) return RPE_E_BADARG;
    if (n == 0) return RPE_OK;
    hipStream_t st = (hipStream_t)stream;
    int blocks = ceil_div(n, 256);
    if (dtype == RPE_F32) hipLaunchKernelGGL(k_inv<float>, dim3(blocks), dim3(256), 0, st, (const float*)T, (float*)R, n);
    else if (dtype == RPE_F64) hipLaunchKernelGGL(k_inv<double>, dim3(blocks), dim3(256), 0, st, (const double*)T, (double*)R, n);
    else return RPE_E_BADARG;
    return rpe_check_launch();
}
int rpe_se3_act(const void* T, const void* pts, void* out, int64_t n, int64_t m, int dtype, void* stream) {
    if (!T || !pts || !out || n < 0 || m < 0) return RPE_E_BADARG;
    if (n * m == 0) return RPE_OK;
    hipStream_t st = (hipStream_t)stream;
    int blocks = ceil_div(n * m, 256);
    if (dtype == RPE_F32) hipLaunchKernelGGL(k_act<float>, dim3(blocks), dim3(256), 0, st, (const float*)T, (const float*)pts, (float*)out, n, m);
    else if (dtype == RPE_F64) hipLaunchKernelGGL(k_act<double>, dim3(blocks), dim3(256), 0, st, (const double*)T, (const double*)pts, (double*)out, n, m);
    else return RPE_E_BADARG;
    return rpe_check_launch();
}
int rpe_se3_chain(const void* rel, const void* init, void* out, int64_t m, double s, int dtype, void* stream) {
    if (!rel || !out || m < 0) return RPE_E_BADARG;
    if (m == 0) return RPE_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RPE_F32) hipLaunchKernelGGL(k_chain<float>, dim3(1), dim3(64), 0, st, (const float*)rel, (const float*)init, (float*)out, m, (float)s);
    else if (dtype == RPE_F64) hipLaunchKernelGGL(k_chain<double>, dim3(1), dim3(64), 0, st, (const double*)rel, (const double*)init, (double*)out, m, s);
    else return RPE_E_BADARG;
    return rpe_check_launch();
}
int rpe_pose_gate_chain(const void* rel, const void* init, void* rel_out, void* abs_out, int32_t* ok, int64_t m, double s, double thr,
                        int dtype, void* stream) {
    if (!rel || !abs_out || m < 0) return RPE_E_BADARG;
    if (m == 0) return RPE_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RPE_F32) hipLaunchKernelGGL(k_gate_chain<float>, dim3(1), dim3(64), 0, st, (const float*)rel, (const float*)init, (float*)rel_out,
                                             (float*)abs_out, ok, m, (float)s, (float)thr);
    else if (dtype == RPE_F64) hipLaunchKernelGGL(k_gate_chain<double>, dim3(1), dim3(64), 0, st, (const double*)rel, (const double*)init, (double*)rel_out,
                                                  (double*)abs_out, ok, m, s, thr);
    else return RPE_E_BADARG;
    return rpe_check_launch();
}
}  // extern "C"
