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

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
