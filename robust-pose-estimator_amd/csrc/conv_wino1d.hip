// The separable 1x5 / 5x1 convolutions of RAFT's SepConvGRU as Winograd F(4, 5) along the filter axis, on the f32 matrix cores.
//
// Replaces (reference's RAFT submodule, call sites core/pose/pose_net.py:47,65,129), twelve times per pass:
//   core/RAFT/core/update.py  SepConvGRU.convz1|convr1, convq1 (1x5) and convz2|convr2, convq2 (5x1) with their gate
//   arithmetic (sigmoid, r*h, tanh, the convex update of h) fused into the epilogue, like rpe_conv_fused's gate modes.
// These are the largest single share of a pass (a direct implicit GEMM needs 5 multiply-adds per output and input channel);
// Toom-Cook / Winograd F(4,5) computes 4 outputs of a 5-tap filter from 8 products instead of 20 (Lavin & Gray 2016, 1-D):
//     y = A^T [ (G g) .* (B^T d) ],  d = 8 consecutive inputs, g = the 5 taps, points {0, +-1, +-2, +-1/2, inf}
//   (B^T is the 8-point matrix of F(6,3); every entry is dyadic, so B^T d is exact up to the usual rounding of its sums).
// In f32 the result differs from a direct f32 accumulation by about as much as two different summation orders do
// (measured on 256-channel sums: mean |err| 6e-6 against 2.4e-6 for a serial direct sum; tools/winograd_numerics.py).
//
// Workgroup = 4 waves = 64 output channels x 64 tiles (16 x 16 output pixels; a tile = 4 pixels along the filter axis);
// wave = 32 channels x 32 tiles = 2 x 2 blocks of v_mfma_f32_16x16x4_f32 per position, 8 positions: 128 accumulator
// registers.  K is walked in steps of 4 input channels; the pipeline (LDS-DMA rings for the transformed weights U and the raw
// input patch, V = B^T d built one step ahead, the barrier in front of the last quarter of a step's matrix instructions)
// is conv_wino.hip's, which explains it.  A (channel | tile) row holds its 8 positions contiguously (32 B): a fragment read is
// 16 B = four positions; the two halves of row r are swapped when ((r >> 2) ^ (r >> 3)) & 1 so that the 16 lanes of a read
// group cover all 64 banks.
#include "wino_common.h"
#include <type_traits>
#include <cstdlib>

#define W1_CO 64
#define W1_NT 64
#define W1_K 4
#define W1_ROW 8
#define W1_USTEP (W1_K * W1_CO * W1_ROW)          // 2048 floats = 8 KB: weights of a step for 64 output channels = V of a step
#define W1_RAWF 1536                              // raw patch of a step: 1x5: 4 ci x 16 rows x 24 columns; 5x1: 4 ci x 20 rows x 16 columns
#define W1_BIAS 1024u                             // keeps the per-lane DMA offsets non-negative

struct W1P {
    const float* x; long long xbs;
    const float* wp; int cin, cout, coP, H, W;
    const float* bias; const float* add; long long abs_;
    float* out; long long obs; float* out2; long long o2bs;
    const float* hid; long long hbs; const float* z; long long zbs;
    int cgate, mode;
};

__device__ __forceinline__ int row_swap(int r) { return ((r >> 2) ^ (r >> 3)) & 1; }

// CB = 16-channel blocks per wave: 2 -> 64 output channels per workgroup; 1 -> 32 (wave = 16 channels x 32 tiles): twice the
// workgroups for launches that do not fill the chip (sequential tracking).  Same products and summation order: bit-identical outputs.
template <bool VERT, int CB>
__global__ __launch_bounds__(256, 2) void k_conv_wino1d(W1P P) {
    constexpr int TCO = 32 * CB, UT_STEP = W1_K * TCO * W1_ROW;
    __shared__ __attribute__((aligned(16))) float Us[3][UT_STEP];             // [ci][co][8 positions], as packed in global memory
    __shared__ __attribute__((aligned(16))) float Vs[2][W1_USTEP];            // [ci][tile][8 positions]
    __shared__ __attribute__((aligned(16))) float Rs[3][W1_RAWF];             // raw input patch [ci][row][column]
    // every kernel argument in ONE batch of scalar loads (left to the compiler: three dependent fetch - wait rounds, 6-9 k cycles
    // before the first DMA)
    asm volatile("" :: "s"(P.x), "s"(P.wp), "s"(P.out), "s"(P.bias), "s"(P.xbs), "s"(P.obs), "s"(P.cin), "s"(P.cout), "s"(P.coP), "s"(P.H),
                 "s"(P.W), "s"(P.mode), "s"(P.cgate));
    asm volatile("" :: "s"(P.add), "s"(P.abs_), "s"(P.out2), "s"(P.o2bs), "s"(P.hid), "s"(P.hbs), "s"(P.z), "s"(P.zbs));
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ptx = (P.W + 15) / 16;
    // each XCD (workgroups are dealt round-robin by linear id) takes a contiguous run of patches: neighbours share halo and output lines in one L2
#ifndef WINO_NO_XCD
    const int pid = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
#else
    const int pid = blockIdx.x;
#endif
    const int x0 = (pid % ptx) * 16, y0 = (pid / ptx) * 16;
    const int co0 = blockIdx.y * TCO, bz = blockIdx.z;
    const int H = P.H, W = P.W, hw = H * W;
    const float* xb = P.x + (size_t)bz * P.xbs;
    const int nsteps = P.cin / W1_K;

    // ---- DMA roles.  Raw patch in 16-B quads (W % 4 == 0): 1x5: columns x0-4 .. x0+19 of 16 rows, 6 quads a row, 384 a step =
    // 8 instructions of 48 lanes; 5x1: rows y0-2 .. y0+17 of 16 columns, 320 quads = 8 instructions of 40 lanes (the other lanes
    // are masked off).  Quads outside the map are read from the clamped position and zeroed after they have landed (border
    // workgroups only).  U: the step's 8 KB slice, two 1 KB chunks per wave.
    constexpr int LANES = VERT ? 40 : 48, RQ_CI = VERT ? 80 : 96, RQ_ROW = VERT ? 4 : 6;
    constexpr unsigned RSTRIDE = LANES * 16u;
    unsigned roff[2] = {0u, 0u};
    unsigned oob = 0;
    if (lane < LANES) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = LANES * (2 * wv + j) + lane;
            const int ci = q / RQ_CI, rem = q - ci * RQ_CI, r = rem / RQ_ROW, cq = rem - r * RQ_ROW;
            int yy = VERT ? y0 - 2 + r : y0 + r, xx = VERT ? x0 + 4 * cq : x0 - 4 + 4 * cq;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) oob |= 1u << j;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy); xx = xx < 0 ? 0 : (xx >= W ? W - 4 : xx);
            roff[j] = (unsigned)(ci * hw + yy * W + xx) * 4u + W1_BIAS - RSTRIDE * j;
        }
    }
    const bool border = VERT ? ((y0 < 2) | (y0 + 18 > H) | (x0 + 16 > W)) : ((y0 + 16 > H) | (x0 < 4) | (x0 + 20 > W));       // workgroup-uniform
    auto patch_raw = [&](int buf) {
        if (border) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if ((oob >> j) & 1) *(f32x4*)&Rs[buf][4 * (LANES * (2 * wv + j) + lane)] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    // packed weights: [step][64-channel tile][ci][co % 64][8].  CB = 2: the wave's two 1 KB chunks are consecutive; CB = 1: wave = input
    // channel, its 32 rows (1 KB) start at row co0 % 64 of that channel's 64
    const float* wslice = CB == 2 ? P.wp + (size_t)(co0 / W1_CO) * W1_USTEP + (size_t)(wv * 2) * 256
                                  : P.wp + (size_t)(co0 / W1_CO) * W1_USTEP + (size_t)wv * (W1_CO * W1_ROW) + (size_t)(co0 % W1_CO) * W1_ROW;
    const unsigned uoff = lane * 16u;
    const size_t wstep = (size_t)(P.coP / W1_CO) * W1_USTEP, rstep = (size_t)W1_K * hw;
    const unsigned us_base = lds_addr_of(&Us[0][0]) + (unsigned)wv * (CB == 2 ? 2048u : 1024u), rs_base = lds_addr_of(&Rs[0][0]) + (unsigned)(2 * wv) * RSTRIDE;
    const unsigned long long lane_mask = (1ull << LANES) - 1ull;
    auto dma_u = [&](const float* src, int buf) {
        if (CB == 2) dma16x2(src, uoff, us_base + (unsigned)buf * (UT_STEP * 4u));
        else dma16x1(src, uoff, us_base + (unsigned)buf * (UT_STEP * 4u));
    };
    auto dma_raw = [&](const float* src, int buf) { dma16x2_masked<RSTRIDE>(src, roff[0], roff[1], rs_base + (unsigned)buf * (W1_RAWF * 4u), lane_mask); };
    auto clamped = [&](int step) { return step < nsteps ? step : nsteps - 1; };    // (past the end: a harmless repeat keeps the DMA count per step constant)
    const float* xsrc = xb - W1_BIAS / 4;

    // ---- transform role: thread -> (input channel of the step = wave, tile = lane).  V = B^T d, 8 positions from the 8 inputs
    // d_k = x[4 t - 2 + k] along the axis:
    //   v0 = d0 - d6 + 21/4 (d4 - d2)                      v7 = d7 - d1 + 21/4 (d3 - d5)
    //   v1|v2 = (d2 + d6 - 17/4 d4) +- (d1 + d5 - 17/4 d3)
    //   v3|v4 = (d6 + 1/4 d2 - 5/4 d4) +- (1/2 d1 - 5/2 d3 + 2 d5)
    //   v5|v6 = (d6 + 4 d2 - 5 d4) +- (2 d1 - 5/2 d3 + 1/2 d5)
    const int t_src = VERT ? wv * (RQ_CI * 4) + (4 * (lane >> 4)) * 16 + (lane & 15) : wv * (RQ_CI * 4) + (lane >> 2) * 24 + 4 * (lane & 3) + 2;
    const int t_dst = (wv * W1_NT + lane) * W1_ROW, t_swap = row_swap(lane & 15);
    float d[8];
    f32x4 vlo, vhi;
    auto tr_read = [&](int rbuf) {
        const float* rp = &Rs[rbuf][t_src];
        if (VERT) {
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = rp[16 * k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; k += 2) { const float2 v = *(const float2*)(rp + k); d[k] = v.x; d[k + 1] = v.y; }
        }
    };
    auto tr_math = [&]() {
        const float a12 = fmaf(-4.25f, d[4], d[2] + d[6]), b12 = fmaf(-4.25f, d[3], d[1] + d[5]);
        const float a34 = fmaf(-1.25f, d[4], fmaf(0.25f, d[2], d[6])), b34 = fmaf(2.0f, d[5], fmaf(-2.5f, d[3], 0.5f * d[1]));
        const float a56 = fmaf(-5.0f, d[4], fmaf(4.0f, d[2], d[6])), b56 = fmaf(0.5f, d[5], fmaf(-2.5f, d[3], 2.0f * d[1]));
        vlo = (f32x4){fmaf(5.25f, d[4] - d[2], d[0] - d[6]), a12 + b12, a12 - b12, a34 + b34};
        vhi = (f32x4){a34 - b34, a56 + b56, a56 - b56, fmaf(5.25f, d[3] - d[5], d[7] - d[1])};
    };
    auto tr_store = [&](int vbuf) {
        *(f32x4*)&Vs[vbuf][t_dst + 4 * t_swap] = vlo;
        *(f32x4*)&Vs[vbuf][t_dst + 4 * (t_swap ^ 1)] = vhi;
    };

    f32x4 acc[8][CB][2];                                       // [position][channel block][tile block]
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[p][c][t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const int cw = wv >> 1, tw = wv & 1, li = lane & 15, lk = lane >> 4;
    float bi_[CB][4];                                           // per-channel bias, requested before the first DMA (conv_wino.hip explains)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + cw * 16 * CB + cb * 16 + 4 * lk + r;
            bi_[cb][r] = P.bias ? P.bias[co < P.cout ? co : P.cout - 1] : 0.0f;
        }

    // ---- prologue: everything the first steps need is requested at once (the groups the loop expects in flight, in its order);
    // the wave waits for U(0), raw(0), raw(1) only, V(0) is built, and raw(3) follows once raw(0)'s buffer is free
    dma_u(wslice, 0); dma_raw(xsrc, 0); dma_raw(xsrc + (size_t)clamped(1) * rstep, 1);
    dma_u(wslice + (size_t)clamped(1) * wstep, 1); dma_raw(xsrc + (size_t)clamped(2) * rstep, 2);
    dma_u(wslice + (size_t)clamped(2) * wstep, 2);
    if (CB == 2) __builtin_amdgcn_s_waitcnt(0x0F76); else __builtin_amdgcn_s_waitcnt(0x0F74);     // vmcnt(2 + 2 + 2 | 1 + 2 + 1)
    patch_raw(0); patch_raw(1);
    __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    tr_read(0); tr_math(); tr_store(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                             // V(0) visible; raw(0)'s buffer free
    dma_raw(xsrc + (size_t)clamped(3) * rstep, 0);
    const float* unext = wave_uniform(wslice + (size_t)clamped(3) * wstep);         // U(s + 3), raw(s + 4) of the step the loop is in
    const float* rnext = wave_uniform(xsrc + (size_t)clamped(4) * rstep);
    // ---- step s: four groups of 8 matrix instructions = (positions 0-3 | 4-7) x (channel block 0 | 1), each against both tile blocks:
    //   g0 (lo, c0) | read A(lo, c1); patch reads of raw(s+1)
    //   g1 (lo, c1) | read A(hi, c0), B(hi, t0), B(hi, t1) | the transform's arithmetic (one cluster: vector instructions are not
    //               hidden by f32 matrix instructions, an isolated one costs ~13 cycles, one more in a cluster ~4)
    //   g2 (hi, c0) | read A(hi, c1) | V(s+1) stored | wait: own DMAs older than the newest group landed | border: patch raw(s+2) | BARRIER
    //   g3 (hi, c1) | read A(lo, c0), B(lo, t0), B(lo, t1) of step s+1 | DMA U(s+3) -> U(s)'s buffer, raw(s+4) -> raw(s+1)'s
    const int sl = 4 * row_swap(li);                          // float offset of the logical low half within this lane's rows
    const int aoff = (lk * TCO + cw * 16 * CB + li) * W1_ROW, boff = (lk * W1_NT + tw * 32 + li) * W1_ROW;
    f32x4 fa = *(const f32x4*)&Us[0][aoff + sl], fb0 = *(const f32x4*)&Vs[0][boff + sl], fb1 = *(const f32x4*)&Vs[0][boff + 16 * W1_ROW + sl];
    // (CB = 1: the four groups are positions (0-1 | 2-3 | 4-5 | 6-7) of the wave's one channel block, four matrix instructions each)
    auto mfma_range = [&](int p0, int np, int cb, const f32x4& a, const f32x4& b0, const f32x4& b1) {
#pragma unroll
        for (int e = 0; e < np; ++e) {
            const int p = p0 + e;
            acc[p][cb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p & 3], b0[p & 3], acc[p][cb][0], 0, 0, 0);
            acc[p][cb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p & 3], b1[p & 3], acc[p][cb][1], 0, 0, 0);
        }
    };
    auto step = [&](auto ubc, auto curc, const int s) {
        constexpr int UB = decltype(ubc)::value, CUR = decltype(curc)::value, UB1 = (UB + 1) % 3, UB2 = (UB + 2) % 3;
        const float* ua = &Us[UB][aoff];
        const float* vb = &Vs[CUR][boff];
        // g0
        f32x4 a_lo1 = fa;
        if (CB == 2) a_lo1 = *(const f32x4*)(ua + 16 * W1_ROW + sl);
        tr_read(UB1);
        __builtin_amdgcn_sched_barrier(0);
        if (CB == 2) mfma_range(0, 4, 0, fa, fb0, fb1); else mfma_range(0, 2, 0, fa, fb0, fb1);
        __builtin_amdgcn_sched_barrier(0);
        // g1
        const f32x4 a_hi0 = *(const f32x4*)(ua + (sl ^ 4)), b_hi0 = *(const f32x4*)(vb + (sl ^ 4)), b_hi1 = *(const f32x4*)(vb + 16 * W1_ROW + (sl ^ 4));
        __builtin_amdgcn_sched_barrier(0);
        tr_math();
        __builtin_amdgcn_sched_barrier(0);
        if (CB == 2) mfma_range(0, 4, CB - 1, a_lo1, fb0, fb1); else mfma_range(2, 2, 0, fa, fb0, fb1);
        __builtin_amdgcn_sched_barrier(0);
        // g2
        f32x4 a_hi1 = a_hi0;
        if (CB == 2) a_hi1 = *(const f32x4*)(ua + 16 * W1_ROW + (sl ^ 4));
        __builtin_amdgcn_sched_barrier(0);
        tr_store(CUR ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (CB == 2) mfma_range(4, 4, 0, a_hi0, b_hi0, b_hi1); else mfma_range(4, 2, 0, a_hi0, b_hi0, b_hi1);
        __builtin_amdgcn_sched_barrier(0);
        // own DMAs except the newest group (CB weight + 2 patch instructions) have landed; the V stores and every fragment read
        // of this step are complete: after the barrier U(s), V(s) and raw(s+1) may be overwritten
        if (CB == 2) __builtin_amdgcn_s_waitcnt(0x0F74); else __builtin_amdgcn_s_waitcnt(0x0F73);     // vmcnt(4 | 3)
        patch_raw(UB2);
        __builtin_amdgcn_s_waitcnt(0xC07F);                                      // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // g3
        fa = *(const f32x4*)&Us[UB1][aoff + sl]; fb0 = *(const f32x4*)&Vs[CUR ^ 1][boff + sl]; fb1 = *(const f32x4*)&Vs[CUR ^ 1][boff + 16 * W1_ROW + sl];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 2 * CB; ++e) {
            const int p = CB == 2 ? 4 + e : 6 + e;
            acc[p][CB - 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_hi1[p & 3], b_hi0[p & 3], acc[p][CB - 1][0], 0, 0, 0);
            acc[p][CB - 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_hi1[p & 3], b_hi1[p & 3], acc[p][CB - 1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (e == 0) { dma_u(unext, UB); if (s + 4 < nsteps) unext += wstep; }
            if (e == 1) { dma_raw(rnext, UB1); if (s + 5 < nsteps) rnext += rstep; }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1; typedef std::integral_constant<int, 2> I2;
        int s = 0;
        while (true) {
            step(I0{}, I0{}, s); if (++s == nsteps) break;
            step(I1{}, I1{}, s); if (++s == nsteps) break;
            step(I2{}, I0{}, s); if (++s == nsteps) break;
            step(I0{}, I1{}, s); if (++s == nsteps) break;
            step(I1{}, I0{}, s); if (++s == nsteps) break;
            step(I2{}, I1{}, s); if (++s == nsteps) break;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // the repeats issued past the end have landed before LDS is released

    // ---- epilogue.  D layout of the 16x16 MFMA: column (tile) = lane % 16, row (channel) = 4 * (lane / 16) + r.
    // y = A^T m:  y0 = m0 + (m1+m2) + (m3+m4) + (m5+m6);   y1 = (m1-m2) + 2 (m3-m4) + 1/2 (m5-m6);
    //             y2 = (m1+m2) + 4 (m3+m4) + 1/4 (m5+m6);   y3 = (m1-m2) + 8 (m3-m4) + 1/8 (m5-m6) + m7;  then the gate arithmetic.
    const int mode = P.mode, cg = P.cgate;
    const float* addb = P.add ? P.add + (size_t)bz * P.abs_ : nullptr;
    const float* hb = P.hid ? P.hid + (size_t)bz * P.hbs : nullptr;
    const float* zb = P.z ? P.z + (size_t)bz * P.zbs : nullptr;
    float* outb = P.out + (size_t)bz * P.obs;
    float* out2b = P.out2 ? P.out2 + (size_t)bz * P.o2bs : nullptr;
    // 5x1: a lane's four outputs are four ROWS of one column.  Moving them (and the add / hidden / z operands) as strided dwords
    // costs four times the memory instructions of the 1x5 case (PMC: matrix pipe 58 % busy against 68 %), so each 16-lane
    // group first transposes its 16 columns x 4 rows through LDS -- the wave's own slice of the idle weight ring -- and a lane
    // ends up with four consecutive columns of one row: every access below is 16 bytes in both orientations.
    float* tsc = &Us[0][wv * (CB == 2 ? 512 : 256)] + lk * 64;
    // Compile-time shapes: MODE = the GRU gate (or -1: linear / ReLU / second output, tested at run time) and FAST = nothing to mask
    // (the patch lies inside the map, every channel of the tile exists, and for GATE_ZR the z | r boundary is a tile boundary, so a
    // workgroup is all z or all r).  As one loop with every test per row and lane the epilogue was 2 000 vector and 1 200 scalar
    // instructions with 330 branches per wave -- a tenth of a 96-step workgroup.
    auto epilogue = [&](auto modec, auto fastc) {
        constexpr int MODE = decltype(modec)::value;
        constexpr bool FAST = decltype(fastc)::value;
        const bool rtile = co0 >= cg;                                     // (FAST GATE_ZR: uniform)
        constexpr int NB = 4 * CB, RU = 2;                                // units of (16 tiles x 16 channels, RU of a lane's four channel rows): n = (tb * CB + cb) * 2 + half
        const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
        auto where = [&](int tb, int& oy, int& ox) {
            const int tile = tw * 32 + tb * 16 + li;
            oy = VERT ? y0 + 4 * (tile >> 4) + (li >> 2) : y0 + (tile >> 2); ox = VERT ? x0 + 4 * (li & 3) : x0 + 4 * (tile & 3);
        };
        // every operand of a block's gate arithmetic (four channel rows: addend, and z | h of the gates), requested at once
        auto load_block = [&](int n, f32x4 (&av)[RU], f32x4 (&xv)[RU], f32x4 (&hv)[RU]) {
            const int tb = (n >> 1) / CB, cb = (n >> 1) % CB, r0 = RU * (n & 1);
            int oy, ox; where(tb, oy, ox);
            const bool pok = FAST || ((oy < H) & (ox < W));               // (W % 4 == 0: a quad is inside or outside as a whole)
            const size_t pxo = (size_t)oy * W + ox;
#pragma unroll
            for (int q = 0; q < RU; ++q) {
                const int co = co0 + cw * 16 * CB + cb * 16 + 4 * lk + r0 + q;
                const bool ok = FAST || (pok && co < P.cout);
                const size_t e0 = (size_t)co * hw + pxo;
                av[q] = (addb && ok) ? *(const f32x4*)(addb + e0) : zero4;
                xv[q] = hv[q] = zero4;
                if (MODE == RPE_CONV_GATE_ZR) { if (FAST ? rtile : (ok && co >= cg)) xv[q] = *(const f32x4*)(hb + e0 - (size_t)cg * hw); }
                else if (MODE == RPE_CONV_GATE_H) { if (ok) { xv[q] = *(const f32x4*)(zb + e0); hv[q] = *(const f32x4*)(hb + e0); } }
            }
        };
        auto finish_block = [&](int n, const f32x4 (&av)[RU], const f32x4 (&xv)[RU], const f32x4 (&hv)[RU]) {
            const int tb = (n >> 1) / CB, cb = (n >> 1) % CB, r0 = RU * (n & 1);
            int oy, ox; where(tb, oy, ox);
            const bool pok = FAST || ((oy < H) & (ox < W));
            const size_t pxo = (size_t)oy * W + ox;
            auto st4 = [&](float* p, size_t e, const float (&v)[4]) {
                if (pok) *(f32x4*)(p + e) = (f32x4){v[0], v[1], v[2], v[3]};
            };
#pragma unroll
            for (int q = 0; q < RU; ++q) {
                const int r = r0 + q, co = co0 + cw * 16 * CB + cb * 16 + 4 * lk + r;
                if (!FAST && co >= P.cout) continue;
                float m[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) m[p] = acc[p][cb][tb][r];
                const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4], s56 = m[5] + m[6], d56 = m[5] - m[6];
                float v[4] = {((m[0] + s12) + s34) + s56, fmaf(0.5f, d56, fmaf(2.0f, d34, d12)), fmaf(0.25f, s56, fmaf(4.0f, s34, s12)),
                              m[7] + fmaf(0.125f, d56, fmaf(8.0f, d34, d12))};
                if (VERT) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) tsc[i * 16 + li] = v[i];                 // [row][column]
                    const f32x4 q = *(const f32x4*)&tsc[(li >> 2) * 16 + 4 * (li & 3)];  // (LDS operations of a wave execute in order)
                    v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
                }
                const size_t e0 = (size_t)co * hw + pxo;
                const float bi = bi_[cb][r];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] + av[q][i] + bi;
                if (MODE == RPE_CONV_GATE_ZR) {
                    // z = sigmoid(.) -> out (channels < gate_channels);  r = sigmoid(.), r * h -> out2 (the other half)
                    float sg[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) sg[i] = sigmoid_f(v[i]);
                    if (FAST ? rtile : co >= cg) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) sg[i] *= xv[q][i];
                        st4(out2b, e0 - (size_t)cg * hw, sg);
                    } else st4(outb, e0, sg);
                } else if (MODE == RPE_CONV_GATE_H) {
                    // h <- (1 - z) h + z tanh(.)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = (1.0f - xv[q][i]) * hv[q][i] + xv[q][i] * tanh_f(v[i]);
                    st4(outb, e0, v);
                } else {
                    if (mode == RPE_CONV_RELU) {                                  // NaN stays NaN, like torch.relu
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = v[i] < 0.0f ? 0.0f : v[i];
                    }
                    st4(outb, e0, v);
                    if (out2b) st4(out2b, e0, v);
                }
            }
        };
        // unit n + 1's operands are requested BEFORE unit n's arithmetic and stores: the tensors may alias as far as the compiler
        // knows (h is updated in place), so it keeps loads behind earlier stores -- written block by block, each block waited a
        // memory round trip behind the previous block's stores, four times per workgroup.  An element is read and written by the
        // same lane only, so requesting a later block's operands early is safe.
        f32x4 av_[2][RU], xv_[2][RU], hv_[2][RU];
        load_block(0, av_[0], xv_[0], hv_[0]);
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            if (n + 1 < NB) load_block(n + 1, av_[(n + 1) & 1], xv_[(n + 1) & 1], hv_[(n + 1) & 1]);
            finish_block(n, av_[n & 1], xv_[n & 1], hv_[n & 1]);
        }
    };
    {
        typedef std::integral_constant<int, RPE_CONV_GATE_ZR> ZR; typedef std::integral_constant<int, RPE_CONV_GATE_H> GH; typedef std::integral_constant<int, -1> RT;
        typedef std::integral_constant<bool, true> Yes; typedef std::integral_constant<bool, false> No;
        const bool inside = (y0 + 16 <= H) & (x0 + 16 <= W) & (co0 + TCO <= P.cout);
        if (mode == RPE_CONV_GATE_ZR) { if (inside && (cg % TCO) == 0) epilogue(ZR{}, Yes{}); else epilogue(ZR{}, No{}); }
        // (GATE_H in full launches keeps the masked shape: measured at batch 32, the q convolutions are 3-6 % SLOWER with the unmasked one
        // (272 vs 265 us, 245 vs 235) and 5 % faster at batch 2 -- with two workgroups per CU an epilogue that issues its 64 KB of loads
        // and stores per wave in one burst starves the other workgroup's DMA ring; alone on a CU it is pure latency.  Same bits.)
        else if (mode == RPE_CONV_GATE_H) { if (inside && CB == 1) epilogue(GH{}, Yes{}); else epilogue(GH{}, No{}); }
        else { if (inside) epilogue(RT{}, Yes{}); else epilogue(RT{}, No{}); }
    }
}

// weight (cout, cin, 5 taps) [= (cout, cin, 1, 5) or (cout, cin, 5, 1)] -> U = G g, laid out
// [step = ci/4][co tile = co/64][ci%4][co%64][8 positions; halves swapped when row_swap(co % 16)]: a workgroup's slice of a step is
// 8 KB contiguous (its LDS image)
__global__ void k_wino1d_pack(const float* __restrict__ w, float* __restrict__ wp, int cout, int cin, int coP, long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int slot = (int)(e & 7), col = (int)((e >> 3) & 63), cil = (int)((e >> 9) & 3);
    const int pos = (((slot >> 2) ^ row_swap(col & 15)) << 2) | (slot & 3);
    const long long rest = e >> 11;
    const int ncot = coP / W1_CO;
    const int co = (int)(rest % ncot) * W1_CO + col, ci = (int)(rest / ncot) * W1_K + cil;
    double v = 0.0;
    if (co < cout && ci < cin) {
        const float* g = w + ((size_t)co * cin + ci) * 5;
        // G = the evaluation matrix at {0, 1, -1, 2, -2, 1/2, -1/2, inf}, rows scaled to match the dyadic B^T used by the kernel
        const double G[8][5] = {{1.0, 0.0, 0.0, 0.0, 0.0},
                                {-2.0 / 9.0, -2.0 / 9.0, -2.0 / 9.0, -2.0 / 9.0, -2.0 / 9.0},
                                {-2.0 / 9.0, 2.0 / 9.0, -2.0 / 9.0, 2.0 / 9.0, -2.0 / 9.0},
                                {1.0 / 90.0, 1.0 / 45.0, 2.0 / 45.0, 4.0 / 45.0, 8.0 / 45.0},
                                {1.0 / 90.0, -1.0 / 45.0, 2.0 / 45.0, -4.0 / 45.0, 8.0 / 45.0},
                                {32.0 / 45.0, 16.0 / 45.0, 8.0 / 45.0, 4.0 / 45.0, 2.0 / 45.0},
                                {32.0 / 45.0, -16.0 / 45.0, 8.0 / 45.0, -4.0 / 45.0, 2.0 / 45.0},
                                {0.0, 0.0, 0.0, 0.0, 1.0}};
#pragma unroll
        for (int k = 0; k < 5; ++k) v += G[pos][k] * (double)g[k];
    }
    wp[e] = (float)v;
}

static inline int w1_cop(int cout) { return (cout + W1_CO - 1) / W1_CO * W1_CO; }

extern "C" size_t rpe_conv_wino1d_packed_floats(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cin % W1_K) return 0;
    return (size_t)(cin / W1_K) * W1_K * W1_ROW * w1_cop(cout);
}

extern "C" int rpe_conv_wino1d_pack(const float* weight, int cout, int cin, float* packed, void* stream) {
    if (!weight || !packed || cout <= 0 || cin <= 0) return RPE_E_BADARG;
    if (cin % W1_K) return RPE_E_UNSUPPORTED;
    const long long total = (long long)rpe_conv_wino1d_packed_floats(cout, cin);
    hipLaunchKernelGGL(k_wino1d_pack, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, packed, cout, cin, w1_cop(cout), total);
    return rpe_check_launch();
}

extern "C" int rpe_conv_wino1d(const rpe_conv_desc* d, void* stream) {
    if (!d || !d->x || !d->packed || !d->out || d->b <= 0 || d->cin <= 0 || d->cout <= 0 || d->h <= 0 || d->w <= 0) return RPE_E_BADARG;
    const bool vert = d->kh == 5 && d->kw == 1, horiz = d->kh == 1 && d->kw == 5;
    if (!(vert || horiz) || (d->stride != 0 && d->stride != 1) || (d->cin % W1_K) || (d->w & 3)) return RPE_E_UNSUPPORTED;
    if (d->mode < RPE_CONV_LINEAR || d->mode > RPE_CONV_GATE_H) return RPE_E_BADARG;
    if (d->mode == RPE_CONV_GATE_ZR && (!d->out2 || !d->hidden || d->gate_channels <= 0 || d->cout != 2 * d->gate_channels)) return RPE_E_BADARG;
    if (d->mode == RPE_CONV_GATE_H && (!d->hidden || !d->zgate)) return RPE_E_BADARG;
    if (d->scale || d->residual || d->stats || d->pre_norm) return RPE_E_UNSUPPORTED;
    // 16-byte accesses: the input quads of the LDS-DMA and the four pixels a lane handles in every tensor of the epilogue
    auto a16 = [](const void* p, long long bs) { return !p || ((((uintptr_t)p) & 15) == 0 && (bs & 3) == 0); };
    if (!a16(d->x, d->x_batch_stride) || !a16(d->packed, 0)) return RPE_E_UNSUPPORTED;
    if ((!a16(d->out, d->out_batch_stride) || !a16(d->out2, d->out2_batch_stride) || !a16(d->add, d->add_batch_stride) ||
                  !a16(d->hidden, d->hidden_batch_stride) || !a16(d->zgate, d->zgate_batch_stride))) return RPE_E_UNSUPPORTED;
    W1P P;
    P.x = d->x; P.xbs = d->x_batch_stride; P.wp = d->packed; P.cin = d->cin; P.cout = d->cout; P.coP = w1_cop(d->cout);
    P.H = d->h; P.W = d->w; P.bias = d->bias; P.add = d->add; P.abs_ = d->add_batch_stride;
    P.out = d->out; P.obs = d->out_batch_stride; P.out2 = d->out2; P.o2bs = d->out2_batch_stride;
    P.hid = d->hidden; P.hbs = d->hidden_batch_stride; P.z = d->zgate; P.zbs = d->zgate_batch_stride; P.cgate = d->gate_channels; P.mode = d->mode;
    // Small launches (sequential tracking: 80-320 workgroups of 64 channels) leave most CUs with one workgroup whose K loop is a chain
    // of DMA latencies: 32-channel tiles double the workgroups.  Measured (whole pass, 640x512): batch 1 8.43 -> 7.87 ms, 4 frame pairs
    // 21.9 -> 21.1 ms, 8 pairs equal, 16 pairs (q convolutions: 1 280 workgroups = 2.5 rounds of the 512 slots) 70.96 -> 71.62 ms --
    // at full occupancy one weight fragment per four matrix instructions beats whole rounds, so only below 1.5 rounds.
#ifndef WINO1D_SMALL_WG
#define WINO1D_SMALL_WG 768LL                     /* (tools/build_variant.sh -DWINO1D_SMALL_WG=... for A/B runs) */
#endif
    const long long small_wg = WINO1D_SMALL_WG;
    const unsigned gx = ceil_div(d->w, 16) * ceil_div(d->h, 16);
    if ((long long)gx * (P.coP / W1_CO) * d->b < small_wg) {
        const dim3 grid(gx, ceil_div(d->cout, 32), d->b);
        if (vert) hipLaunchKernelGGL((k_conv_wino1d<true, 1>), grid, dim3(256), 0, (hipStream_t)stream, P);
        else hipLaunchKernelGGL((k_conv_wino1d<false, 1>), grid, dim3(256), 0, (hipStream_t)stream, P);
        return rpe_check_launch();
    }
    const dim3 grid(gx, P.coP / W1_CO, d->b);
    if (vert) hipLaunchKernelGGL((k_conv_wino1d<true, 2>), grid, dim3(256), 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL((k_conv_wino1d<false, 2>), grid, dim3(256), 0, (hipStream_t)stream, P);
    return rpe_check_launch();
}
