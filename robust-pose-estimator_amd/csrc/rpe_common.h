// Shared helpers for the gfx950 kernels of librpe_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rpe.h"

#define RPE_WAVE 64

static inline int rpe_check_launch() {
    return hipGetLastError() == hipSuccess ? RPE_OK : RPE_E_LAUNCH;
}

// 64-lane butterfly sum; every lane ends with the total (deterministic order).
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, RPE_WAVE);
    return v;
}

// One correctly-rounded f32 operation, never contracted into an FMA with its neighbours.  (HIP's __fmul_rn & co. are plain
// operators compiled with contraction allowed unless OCML_BASIC_ROUNDED_OPERATIONS is defined, so they do fuse.)
__device__ __forceinline__ float rn_add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float rn_sub(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ float rn_mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float rn_div(float a, float b) {
#pragma clang fp contract(off)
    return a / b;
}

// Running (count, mean, M2 = sum of squared deviations) in f64 with the parallel-variance merge (Chan et al.): how the
// convolution epilogues' per-wave / per-tile instance-norm statistics are combined without cancellation.
struct StatAcc {
    double n = 0.0, mean = 0.0, m2 = 0.0;
    __device__ __forceinline__ void add(double nb, double mb, double m2b) {
        if (nb <= 0.0) return;
        const double nt = n + nb, d = mb - mean;
        mean += d * nb / nt; m2 += m2b + d * d * n * nb / nt; n = nt;
    }
    // a block given as sums about a pivot p: s1 = sum(v - p), s2 = sum((v - p)^2) over nb values
    __device__ __forceinline__ void add_pivoted(int nb, float s1, float s2, float p) {
        if (nb <= 0) return;
        const double m = (double)s1 / nb;
        add((double)nb, (double)p + m, (double)s2 - (double)s1 * m);
    }
};

// The GRU gates in the convolution epilogues (conv.hip, conv_wino1d.hip) on the hardware exp2 / reciprocal (each within 1 ulp):
// absolute error < 3e-7, the size of one f32 rounding of the values they act on; libm's expf / tanhf cost ~30 vector
// instructions per element, which a matrix-bound kernel cannot hide (vector and f32 matrix instructions share the issue).
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }     // -> +-1 as e^{2x} -> inf | 0
static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
