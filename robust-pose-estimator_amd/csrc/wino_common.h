// Shared pieces of the Winograd convolution kernels (conv_wino.hip: F(2x2,3x3); conv_wino1d.hip: F(4,5) along one axis).
#pragma once
#include "rpe_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA (global -> LDS without staging registers): every lane supplies its own global address, the destination is the
// wave-uniform LDS byte address + lane * size.  Issued as inline asm ON PURPOSE: hipcc waits vmcnt(0) in front of every
// LDS read while one of ITS loads-to-LDS is in flight (it cannot tell the buffers apart), which would serialise the
// three-deep prefetch; these it does not count, and the kernel waits for them itself (s_waitcnt vmcnt(N), in order).
// The global address is a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset + the instruction offset, and the instruction
// offset is added to the LDS address as well: one M0 set-up and no 64-bit vector address arithmetic per group of DMAs.
__device__ __forceinline__ const float* wave_uniform(const float* p) {          // pins a wave-uniform pointer to scalar registers
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float*)(((unsigned long long)hi << 32) | lo);
}
// four 1 KB chunks: global base + voff + 1024 j  ->  LDS lds_addr + 1024 j + lane * 16
__device__ __forceinline__ void dma16x4(const float* base, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    base = wave_uniform(base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
// two 1 KB chunks (the 32-channel tile: one input channel's 32 rows)
__device__ __forceinline__ void dma16x2(const float* base, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    base = wave_uniform(base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
// one 1 KB chunk
__device__ __forceinline__ void dma16x1(const float* base, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    base = wave_uniform(base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
// one chunk, exec-masked
__device__ __forceinline__ void dma16x1_masked(const float* base, unsigned voff, unsigned lds_addr, unsigned long long lane_mask) {
    unsigned keep; unsigned long long ekeep;
    base = wave_uniform(base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %4\n\ts_mov_b64 exec, %5\n\t"
                 "global_load_lds_dwordx4 %2, %3\n\t"
                 "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(ekeep) : "v"(voff), "s"(base), "s"(lds_addr), "s"(lane_mask) : "memory");
}
// two 1 KB chunks of which only the first `lanes` lanes take part (exec-masked): LDS lds_addr + stride j + lane * 16, the
// caller folds -stride j into v_j (and a bias that keeps them non-negative into the base)
template <unsigned STRIDE>
__device__ __forceinline__ void dma16x2_masked(const float* base, unsigned v0, unsigned v1, unsigned lds_addr, unsigned long long lane_mask) {
    unsigned keep; unsigned long long ekeep;
    base = wave_uniform(base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %5\n\ts_mov_b64 exec, %6\n\t"
                 "global_load_lds_dwordx4 %2, %4\n\tglobal_load_lds_dwordx4 %3, %4 offset:%7\n\t"
                 "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(ekeep) : "v"(v0), "v"(v1), "s"(base), "s"(lds_addr), "s"(lane_mask), "n"(STRIDE) : "memory");
}
// Sum over each 16-lane row of the wave with DPP moves (vector-ALU rate, no LDS traffic): quad butterflies, row half-mirror,
// row mirror.  Every lane ends with its row's total.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

