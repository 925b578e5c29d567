// grid_sample position arithmetic shared by the warp and correlation-lookup kernels.
//
// Both reference call sites (core/interpol/flow_utils.py:9-11 and RAFT's bilinear_sampler,
// core/RAFT/core/utils/utils.py) normalise a pixel coordinate v to g = 2*v/(size-1) - 1 and let
// torch.nn.functional.grid_sample(align_corners=True) un-normalise it again as ((g + 1) / 2) * (size - 1).
// The round trip is not the identity in float32, and floor()/rint() of the result decide which pixels are
// read, so it is reproduced here one correctly-rounded operation at a time (contraction into FMAs is switched
// off inside rt_pos; the sequence has no multiply feeding an add anyway).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float rt_pos(float v, int size) {
#pragma clang fp contract(off)      // (__f*_rn are plain operators in this HIP: keep every operation separately rounded)
    float sm1 = (float)(size - 1);
    float g = rn_sub(rn_div(rn_mul(2.0f, v), sm1), 1.0f);
    return rn_mul(rn_div(rn_add(g, 1.0f), 2.0f), sm1);
}

// floor() to int that never overflows: positions beyond +-1e6 (or NaN) are reported far outside any image.
__device__ __forceinline__ int safe_floor(float p, float& pf) {
    if (!(p > -1.0e6f && p < 1.0e6f)) { pf = -1.0e6f; return -1000000; }
    pf = floorf(p);
    return (int)pf;
}
