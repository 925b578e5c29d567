// 1x1 convolutions (plain GEMMs over the pixel axis) on the f32 matrix cores with LDS-DMA operand rings.
//
// Replaces (reference's RAFT submodule, call sites core/pose/pose_net.py:47,65,129):
//   core/RAFT/core/update.py    BasicMotionEncoder.convc1 (324 -> 256, + ReLU), twelve times per pass: the consumer of the
//                               correlation lookup; BasicUpdateBlock.mask[2] (256 -> 576)
//   core/RAFT/core/extractor.py BasicEncoder.conv2 (128 -> 256), with RAFT.forward's tanh | relu split for the context encoder
// rpe_conv_fused ran these through its implicit-GEMM kernel (register-staged operand loads, one __syncthreads per K step:
// 88 TFLOP/s on convc1).  A 1x1 convolution needs none of that kernel's tap / halo machinery: out[co][p] = sum_ci W[co][ci]
// x[ci][p] is a GEMM whose B operand (16 input channels x 128 consecutive pixels) is sixteen contiguous 512-byte rows.
//
// Workgroup = 4 waves = 128 output channels x 128 pixels; wave = 64 x 64 = 2 x 2 blocks of v_mfma_f32_32x32x2_f32 (64
// accumulator registers: three workgroups per CU).  K is walked in steps of 16 input channels.  Per step the weight slice
// (16 x 128, packed contiguously by rpe_conv1x1_pack) and the input slice arrive by LDS-DMA (global_load_lds_dwordx4, two 1 KB
// chunks of each per wave) into 3-deep rings, issued two steps ahead; the kernel counts them itself (s_waitcnt vmcnt(4)), see
// wino_common.h for why that is inline asm.  Both tiles are k-major ([k][128]), which is exactly the matrix instruction's operand
// order: lane l reads A[k0 + l/32][m0 + l%32] -- consecutive lanes, consecutive floats, no bank conflicts, no transposition.
#include "wino_common.h"
#include <type_traits>

#define G1_M 128
#define G1_N 128
#define G1_K 16
#define G1_TILE (G1_K * 128)                     // floats per operand tile of a step (8 KB)
#define G1_BIAS 4096u                            // keeps the per-lane DMA offsets non-negative after the -1024 fold

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct G1P {
    const float* x; long long xbs;
    const float* wp; int cin, cout, coP, hw;
    const float* bias;
    float* out; long long obs; float* out2; long long o2bs;
    int mode;
};

// two 1 KB chunks with their own per-lane offsets: global base + v_j + 1024 j -> LDS lds_addr + 1024 j + lane * 16 (the caller
// folds the -1024 j into v_j)
__device__ __forceinline__ void dma16_2v(const float* base, unsigned v0, unsigned v1, unsigned lds_addr) {
    unsigned keep;
    base = wave_uniform(base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "v"(v1), "s"(base), "s"(lds_addr) : "memory");
}

#ifndef G1_OCC
#define G1_OCC 3
#endif
// one 4-byte store per lane: wave-uniform 64-bit base (scalar registers) + 32-bit per-lane byte offset -- no vector address arithmetic
__device__ __forceinline__ void store_sv(float* base, unsigned voff, float v) {
    asm volatile("global_store_dword %0, %1, %2" :: "v"(voff), "v"(v), "s"(base) : "memory");
}

// LDS row rho of a step's operand tiles holds input channel rho / 2 + 8 (rho & 1) of the step: matrix instruction j of a step then adds
// the products of channels j and 8 + j, in that order -- exactly rpe_conv_fused's k_conv_igemm (its lane half lh of k2-step j supplies
// k = 8 lh + j).  Same products in the same order: the two kernels agree BIT FOR BIT, so which of them ops.Conv1x1 picks for a launch
// (by its workgroup count) never shows in the result (tests/test_gpu_conv.py::test_conv1x1_routes_agree_bitwise).
__host__ __device__ __forceinline__ int g1_chan(int rho) { return (rho >> 1) + 8 * (rho & 1); }

template <int MODE, bool OUT2>
__global__ __launch_bounds__(256, G1_OCC) void k_conv1x1(G1P P) {
    __shared__ __attribute__((aligned(16))) float As[3][G1_TILE];             // [k][co], as packed in global memory
    __shared__ __attribute__((aligned(16))) float Bs[3][G1_TILE];             // [k][pixel]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    // Workgroup order: the output-channel tiles of one pixel tile are neighbours (they read the same input slice), and -- workgroup
    // ids being dealt to the 8 XCDs round-robin -- each XCD takes a contiguous run of that order, so the slice is fetched from HBM
    // once and served to the other channel tiles by that XCD's L2.
#ifndef G1_NO_XCD
    const unsigned lid = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
#else
    const unsigned lid = blockIdx.x;
#endif
    const int ncot = P.coP / G1_M;
    const int cot = lid % ncot;
    const int px0 = (lid / ncot) * G1_N, co0 = cot * G1_M, bz = blockIdx.z;
    const int hw = P.hw, cin = P.cin;
    const int nsteps = (cin + G1_K - 1) / G1_K;
    const float* xb = P.x + (size_t)bz * P.xbs;

    // ---- DMA roles: wave wv moves rows k = 4 wv .. 4 wv + 3 of both tiles (two 1 KB chunks = two rows each)
    // A: packed [co tile][step][k][128]: plain copy.  B: row k = input channel 16 s + k, 128 pixels from px0 (clamped into the plane:
    // columns past hw are never stored), rows past cin clamped to the last channel (their weights are zero).
    const float* wsrc = P.wp + ((size_t)cot * nsteps) * G1_TILE + (size_t)(4 * wv) * 128;
    const unsigned aoff = lane * 16u;
    const int bl = lane & 31, bh = lane >> 5;
    int pxl = px0 + 4 * bl; pxl = pxl + 4 <= hw ? pxl : hw - 4;               // (hw % 4 == 0, hw >= 4)
    unsigned boff[2], boff_last[2];
    bool past[2];                                                            // last step: this lane's row lies past the last input channel
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = g1_chan(4 * wv + 2 * j + bh);                           // LDS row -> input channel of the step
        boff[j] = (unsigned)((size_t)k * hw + pxl) * 4u + G1_BIAS - 1024u * j;
        int kl = (nsteps - 1) * G1_K + k;
        past[j] = kl >= cin;
        kl = kl < cin ? kl : cin - 1;                                         // last step: clamp the channel (the row is zeroed once landed)
        boff_last[j] = (unsigned)((size_t)(kl - (nsteps - 1) * G1_K) * hw + pxl) * 4u + G1_BIAS - 1024u * j;
    }
    const bool ragged = (cin % G1_K) != 0;                                    // (workgroup-uniform)
    const unsigned a_lds = lds_addr_of(&As[0][0]) + (unsigned)(4 * wv) * 512u, b_lds = lds_addr_of(&Bs[0][0]) + (unsigned)(4 * wv) * 512u;
    const float* xsrc = xb - G1_BIAS / 4;
    const size_t bstep = (size_t)G1_K * hw;
    auto issue = [&](int s, int buf) {
        const int sc = s < nsteps ? s : nsteps - 1;                           // past the end: a harmless repeat keeps the DMA count per step constant
        dma16x2(wsrc + (size_t)sc * G1_TILE, aoff, a_lds + (unsigned)buf * (G1_TILE * 4u));
        if (sc == nsteps - 1) dma16_2v(xsrc + (size_t)sc * bstep, boff_last[0], boff_last[1], b_lds + (unsigned)buf * (G1_TILE * 4u));
        else dma16_2v(xsrc + (size_t)sc * bstep, boff[0], boff[1], b_lds + (unsigned)buf * (G1_TILE * 4u));
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // the workgroup's 128 bias values go to LDS before the first DMA is issued (an ordinary load returns in order with the DMAs; the
    // loop's first barrier publishes the store): the epilogue then reads them at immediate offsets, no address arithmetic
    __shared__ float bias_s[G1_M];
    const int l31 = lane & 31, lh = lane >> 5;
    if (tid < G1_M) {
        const int co = co0 + tid;
        bias_s[tid] = (P.bias && co < P.cout) ? P.bias[co] : 0.0f;
    }

    issue(0, 0);
    issue(1, 1);
    // Ring positions are compile-time (the step is written out for the three buffers): every LDS read address is a per-lane constant
    // + an immediate.  Vector instructions do not issue in the shadow of f32 matrix instructions (conv_wino.hip), so the loop holds
    // nothing but the 32 matrix instructions, their 32 fragment reads, the wait, the barrier and the four DMA instructions.
    const float* a_l = &As[0][0] + wm * 64 + l31 + lh * 128;
    const float* b_l = &Bs[0][0] + wn * 64 + l31 + lh * 128;
    auto step = [&](auto bufc, int s) {
        constexpr int BUF = decltype(bufc)::value, NB = (BUF + 2) % 3;
        // own DMAs of step s have landed (the 4 of step s + 1 may still fly); after the barrier everybody's have, and everybody is
        // done reading the buffer of step s - 1, which the DMAs of step s + 2 overwrite
#ifndef G1_NO_DMA
        __builtin_amdgcn_s_waitcnt(0x0F74);                                   // vmcnt(4)
#endif
        if (ragged && s == nsteps - 1) {
            // rows past the last input channel were read from the clamped last channel: their weights are zero, but 0 * Inf = NaN where
            // torch's convolution has no such term -- the rows are overwritten with zeros now that they have landed
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (past[j]) *(f32x4*)&Bs[BUF][(4 * wv + 2 * j) * 128 + lane * 4] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            __builtin_amdgcn_s_waitcnt(0xC07F);                               // lgkmcnt(0)
        }
#ifndef G1_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
#ifndef G1_NO_DMA
        issue(s + 2, NB);
#endif
        const float* a = a_l + BUF * G1_TILE;
        const float* b = b_l + BUF * G1_TILE;
#pragma unroll
        for (int kk = 0; kk < G1_K; kk += 2) {
            const float a0 = a[kk * 128], a1 = a[kk * 128 + 32], b0 = b[kk * 128], b1 = b[kk * 128 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    };
    {
        typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1; typedef std::integral_constant<int, 2> I2;
        int s = 0;
        while (true) {
            step(I0{}, s); if (++s == nsteps) break;
            step(I1{}, s); if (++s == nsteps) break;
            step(I2{}, s); if (++s == nsteps) break;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                       // the repeats issued past the end have landed before LDS is released

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  The address of
    // element (i, j, r) = a wave-uniform row base (scalar registers) + this lane's constant offset: no vector address arithmetic.
    float* ob = P.out + (size_t)bz * P.obs;
    float* ob2 = OUT2 ? P.out2 + (size_t)bz * P.o2bs : nullptr;
    const int co_w = co0 + wm * 64;                                           // wave-uniform
    const int cout = P.cout;
    const float* bsl = &bias_s[wm * 64 + 4 * lh];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int px = px0 + wn * 64 + j * 32 + l31;
        const bool pok = px < hw;
        const unsigned loff = ((unsigned)(4 * lh) * (unsigned)hw + (unsigned)px) * 4u;   // this lane's part: its row group of 4 and its pixel
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row_u = co_w + i * 32 + (r & 3) + 8 * (r >> 2);     // wave-uniform part of the output channel
                float v = acc[i][j][r] + bsl[i * 32 + (r & 3) + 8 * (r >> 2)];
                if (MODE == RPE_CONV_RELU) v = v < 0.0f ? 0.0f : v;           // NaN stays NaN, like torch.relu
                else if (MODE == RPE_CONV_TANH) v = tanh_f(v);
#ifdef G1_NO_STORE
                if (pok && row_u + 4 * lh < cout && v == 1234.5f) {
#else
                if (pok && row_u + 4 * lh < cout) {
#endif
                    store_sv(ob + (size_t)row_u * hw, loff, v);
                    if (OUT2) store_sv(ob2 + (size_t)row_u * hw, loff, v);
                }
            }
    }
}

// weight (cout, cin, 1, 1) -> [co tile = co / 128][step = ci / 16][row rho: ci % 16 = g1_chan(rho)][co % 128], zero beyond cin / cout
__global__ void k_conv1x1_pack(const float* __restrict__ w, float* __restrict__ wp, int cout, int cin, int nsteps, long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int col = (int)(e & 127), k = (int)((e >> 7) & 15);
    const long long rest = e >> 11;
    const int step = (int)(rest % nsteps), tile = (int)(rest / nsteps);
    const int co = tile * 128 + col, ci = step * G1_K + g1_chan(k);
    wp[e] = (co < cout && ci < cin) ? w[(size_t)co * cin + ci] : 0.0f;
}

extern "C" size_t rpe_conv1x1_packed_floats(int cout, int cin) {
    if (cout <= 0 || cin <= 0) return 0;
    return (size_t)((cout + 127) / 128) * ((cin + G1_K - 1) / G1_K) * G1_TILE;
}

extern "C" int rpe_conv1x1_pack(const float* weight, int cout, int cin, float* packed, void* stream) {
    if (!weight || !packed || cout <= 0 || cin <= 0) return RPE_E_BADARG;
    const long long total = (long long)rpe_conv1x1_packed_floats(cout, cin);
    hipLaunchKernelGGL(k_conv1x1_pack, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, packed, cout, cin,
                       (cin + G1_K - 1) / G1_K, total);
    return rpe_check_launch();
}

extern "C" int rpe_conv1x1(const rpe_conv_desc* d, void* stream) {
    if (!d || !d->x || !d->packed || !d->out || d->b <= 0 || d->cin <= 0 || d->cout <= 0 || d->h <= 0 || d->w <= 0) return RPE_E_BADARG;
    if (d->kh != 1 || d->kw != 1 || (d->stride != 0 && d->stride != 1)) return RPE_E_UNSUPPORTED;
    if (d->mode != RPE_CONV_LINEAR && d->mode != RPE_CONV_RELU && d->mode != RPE_CONV_TANH) return RPE_E_UNSUPPORTED;
    if (d->add || d->hidden || d->zgate || d->scale || d->residual || d->stats || d->pre_norm) return RPE_E_UNSUPPORTED;
    const long long hw = (long long)d->h * d->w;
    // 16-byte DMA pieces: plane size and the input slice's base / batch stride; 32-bit byte offsets inside a 16-channel step
    if ((hw & 3) || hw < 4 || (((uintptr_t)d->x) & 15) || (d->x_batch_stride & 3) || (((uintptr_t)d->packed) & 15)) return RPE_E_UNSUPPORTED;
    if (hw * 16 * 4 + G1_BIAS >= (1ll << 32)) return RPE_E_UNSUPPORTED;
    G1P P;
    P.x = d->x; P.xbs = d->x_batch_stride; P.wp = d->packed; P.cin = d->cin; P.cout = d->cout; P.coP = (d->cout + 127) / 128 * 128;
    P.hw = (int)hw; P.bias = d->bias; P.out = d->out; P.obs = d->out_batch_stride; P.out2 = d->out2; P.o2bs = d->out2_batch_stride;
    P.mode = d->mode;
    const dim3 grid(ceil_div(hw, G1_N) * (P.coP / 128), 1, d->b);
#define G1_LAUNCH(M_) do { if (d->out2) hipLaunchKernelGGL((k_conv1x1<M_, true>), grid, dim3(256), 0, (hipStream_t)stream, P); \
                           else hipLaunchKernelGGL((k_conv1x1<M_, false>), grid, dim3(256), 0, (hipStream_t)stream, P); } while (0)
    if (d->mode == RPE_CONV_RELU) G1_LAUNCH(RPE_CONV_RELU);
    else if (d->mode == RPE_CONV_TANH) G1_LAUNCH(RPE_CONV_TANH);
    else G1_LAUNCH(RPE_CONV_LINEAR);
#undef G1_LAUNCH
    return rpe_check_launch();
}
