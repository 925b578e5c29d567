// LABELLED VARIANT (never in the headline number): the SepConvGRU's 1x5 / 5x1 convolutions of conv_wino1d.hip -- Winograd F(4,5) along
// the filter axis with the GRU gates in the epilogue -- on the 16-bit matrix cores, every f32 product of the Winograd domain evaluated as
// SIX bf16 products of an exact three-way split (x = hi + mid + lo; conv_wino_x3.hip explains the arithmetic and what the compiler needs).
//
// Replaces the same reference layers as conv_wino1d.hip (the reference's RAFT submodule, call sites core/pose/pose_net.py:47,65,129:
// core/RAFT/core/update.py SepConvGRU.convz1|convr1, convq1 (1x5), convz2|convr2, convq2 (5x1) and their gate arithmetic).
//
// Structure (conv_wino_x3.hip's, with 8 positions instead of 16):
//   * workgroup = 4 waves, one per SIMD, 128 output channels x 64 tiles (16 x 16 pixels; a tile = 4 pixels along the filter axis);
//   * wave w owns a PAIR of positions -- (0, inf), (+1, -1), (+1/2, -1/2), (+2, -2) -- for all channels and tiles: 2 x 4 x 2 blocks of
//     32 x 32 = all 256 accumulator registers.  The kernel is bound by vector-instruction ISSUE (~6 cycles each beside matrix
//     instructions, tools/probes/valu_rate.hip), and the vector work -- transform and split of the B operand -- does not grow with the
//     channel tile: 128 channels per wave put 24 matrix instructions (768 cycles) beside the ~85 vector instructions of a stage,
//     64 channels (the first version: 12 beside 85) ran at 700-1100 cycles a stage, slower than the f32 kernel;
//   * the two positions of a +- pair share the even and the odd half of the transform (v+- = a +- b); ONE instruction sequence with
//     wave-uniform coefficients serves all four waves (a branch per wave cost four taken branches per channel);
//   * A (U = G g, pre-split at pack time) goes global -> VGPR, each 1 KB block read once per workgroup; a (position, plane) piece is
//     requested again right after its last use of the step;
//   * B (V = B^T d) is built in registers from the raw patch in LDS, half a step ahead, into two 24-register slots;
//   * the raw patch (16 channels per step) arrives by LDS-DMA into THREE buffers, requested two steps ahead: ordinary loads and
//     LDS-DMA are not ordered against each other, so every wait for an A piece also waits for whatever DMA is in flight -- the DMAs
//     are issued early enough (>= 1.4 stages before the next such wait) to have landed by then.  Out-of-map quads are never
//     written (EXEC-masked requests into slots zeroed once).  One barrier per step;
//   * epilogue, twice (64 channels each): wave 0 hands (m0, m7), the others (m+ + m-, m+ - m-) to LDS (8 planes of 64 x 64 = 128 KB
//     over the patch buffers); every thread finishes y = A^T m for 16 quads of four x-neighbouring pixels with the gate arithmetic
//     of conv_wino1d.hip, the gate operands fetched four quads ahead (16 dependent round trips were 35-60 thousand cycles).
#include "wino_common.h"
#include <type_traits>
#pragma clang diagnostic ignored "-Wunused-lambda-capture"    // (lambdas name their captures: clang does not capture variables used only in asm operands of a generic lambda)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define Y_CO 128
#define Y_NCB 4                                  // 32-channel blocks per workgroup
#define YK 16                                    // input channels per step
#define Y_U_WAVE 24576                           // bytes of U per (step, 128-channel tile, wave): [pos 2][cb 4][plane 3][32 co][16 ci] bf16

#ifdef Y3_TIMING
// Experiment hook (tools/build_variant.sh ... -DY3_TIMING): cycle stamps of one mid-grid workgroup's wave 0; rpe_debug_y3_timing reads them.
__device__ unsigned long long g_y3_timing[16];
extern "C" int rpe_debug_y3_timing(unsigned long long* out16) { return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_y3_timing), 128) == hipSuccess ? 0 : -1; }
#define YSTAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); Tacc[i] += now_ - Tlast; Tlast = now_; } while (0)
#else
#define YSTAMP(i)
#endif

struct W1X3P {
    const float* x; long long xbs;
    const unsigned short* wp; int cin, cout, coP, H, W;
    const float* bias; const float* add; long long abs_;
    float* out; long long obs; float* out2; long long o2bs;
    const float* hid; long long hbs; const float* z; long long zbs;
    int cgate, mode;
};

// VERT: 5x1 (tiles = 4 rows of one column), else 1x5 (tiles = 4 columns of one row).
// EPI: 1 = GATE_ZR, 2 = GATE_H with an addend and no bias (the GRU's launches: branch-free final pass); 0 = any mode (P.mode), any operands.
// Raw patch of a channel in LDS:  1x5: 16 rows x 6 quads (map columns x0 - 4 .. x0 + 19), rows 24 floats apart, landing 8 bytes into the buffer
// so that a tile's eight inputs (columns 4 tq - 2 .. 4 tq + 5) are two aligned 16-byte reads;  5x1: 20 rows (y0 - 2 .. y0 + 17) x 4 quads
// + one never-written quad per row, rows 20 floats apart (lanes 16-31 of a read then fall on the other banks), a tile's inputs = 8 rows.
template <bool VERT, int EPI>
__global__ __launch_bounds__(256, 1) void k_conv_wino1d_x3(W1X3P P) {
    constexpr int NCB = Y_NCB;
    constexpr int QR = VERT ? 5 : 6, NR = VERT ? 20 : 16, QCH = QR * NR;          // quads per row / rows / quads per channel (96 | 100)
    constexpr int ROUNDS = VERT ? 7 : 6, WQ = 64 * ROUNDS;                        // DMA rounds per wave and step; quad slots per wave (4 channels)
    constexpr int RBUF = 4 * WQ * 4, SHIFT = VERT ? 0 : 2;                        // floats per raw buffer; landing shift (floats)
    __shared__ __attribute__((aligned(16))) float smem[32768];                    // loop: 3 raw buffers; epilogue: 8 x 64 x 64 floats
    asm volatile("" :: "s"(P.x), "s"(P.wp), "s"(P.out), "s"(P.bias), "s"(P.xbs), "s"(P.obs), "s"(P.cin), "s"(P.cout), "s"(P.coP), "s"(P.H),
                 "s"(P.W), "s"(P.mode), "s"(P.cgate));
    asm volatile("" :: "s"(P.add), "s"(P.abs_), "s"(P.out2), "s"(P.o2bs), "s"(P.hid), "s"(P.hbs), "s"(P.z), "s"(P.zbs));
#ifdef Y3_TIMING
    const unsigned long long Tk0 = __builtin_readcyclecounter();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ptx = (P.W + 15) >> 4;
    const int pid = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;     // XCD-contiguous
    const int x0 = (pid % ptx) * 16, y0 = (pid / ptx) * 16;
    const int co0 = blockIdx.z * Y_CO, bz = blockIdx.y;                           // (grid = patches, images, channel tiles)
    const int H = P.H, W = P.W, hw = H * W;
    const float* xb = P.x + (size_t)bz * P.xbs;
    const int nsteps = P.cin / YK;
    const unsigned long long all_lanes = __ballot(true);

    // ---- DMA role: wave w brings channels 4w .. 4w+3 of a step; logical quad lq = 64 k + lane -> (channel, row, quad column).  Lanes whose
    // quad lies outside the map (or is padding) are masked out of the request: their LDS slots keep the zeros written once below.
    unsigned long long msk[ROUNDS];
    unsigned rr_[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int lq = 64 * k + lane;
        const int cl = lq / QCH, rem = lq - cl * QCH, r = rem / QR, qc = rem - r * QR;
        const int yy = VERT ? y0 - 2 + r : y0 + r, xx = VERT ? x0 + 4 * qc : x0 - 4 + 4 * qc;
        const bool ok = lq < 4 * QCH && !(VERT && qc == 4) && yy >= 0 && yy < H && xx >= 0 && xx + 3 < W;
        msk[k] = __ballot(ok);
        rr_[k] = ok ? (unsigned)((4 * wv + cl) * hw + yy * W + xx) * 4u + 3072u - 1024u * (k & 3) : 3072u;
    }
    const bool border = VERT ? ((y0 < 2) | (y0 + 18 > H) | (x0 + 16 > W)) : ((y0 + 16 > H) | (x0 < 4) | (x0 + 20 > W));       // workgroup-uniform
    if (border) {                                                                 // the wave's own slots of the three buffers: zeros for whatever is never written
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int k = 0; k < ROUNDS; ++k) *(f32x4*)&smem[b * RBUF + (wv * WQ + 64 * k + lane) * 4] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        if (SHIFT && wv == 3 && lane < SHIFT) smem[3 * RBUF + lane] = 0.0f;       // (the landing shift's tail behind the last buffer)
        __syncthreads();                                                          // a neighbour's shifted last quad lands in the next wave's first bytes: zeros first
    }
    const unsigned smem_lds = lds_addr_of(&smem[0]);
    const unsigned rs_base = smem_lds + (unsigned)wv * (WQ * 16u) + SHIFT * 4u;
    const size_t rstep = (size_t)YK * hw;
    const float* dma_src = xb;
    unsigned dma_lds = 0;
    auto dma_begin = [&](int step, unsigned bufoff) {
        dma_src = wave_uniform(xb + (size_t)step * rstep - 768);
        dma_lds = rs_base + bufoff;
    };
    // chunk k: global (src - 3072) + rr_k + 1024 (k & 3)  ->  LDS dma_lds + 1024 k + lane * 16, the lanes of msk[k] & enable only
    auto dma_chunk = [&rr_, &dma_src, &dma_lds, &msk](auto kc, unsigned long long enable) {
        constexpr int k = decltype(kc)::value;
        if (k >= ROUNDS) return;
        unsigned keep;
        const unsigned la = dma_lds + (k >= 4 ? 4096u : 0u);
        const unsigned long long lanes = msk[k < ROUNDS ? k : 0] & enable;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_mov_b64 exec, %5\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b64 exec, -1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(rr_[k < ROUNDS ? k : 0]), "s"(dma_src), "s"(la), "n"((k & 3) * 1024), "s"(lanes) : "memory");
    };

    // ---- A (U) role: the wave's 24 KB of a step: [pos 2][cb 4][plane 3][32 co][16 ci]; lane -> (co = lane & 31, k half = lane >> 5).
    // Requested by inline asm and waited for by hand (wait_a), a (position, plane) piece = four loads.
    const unsigned a_lane = (unsigned)(lane & 31) * 32u + (unsigned)(lane >> 5) * 16u;
    const int nct = P.coP / Y_CO;
    const char* ubase = (const char*)P.wp + ((size_t)(co0 / Y_CO) * 4 + wv) * Y_U_WAVE;
    const size_t ustep = (size_t)nct * 4 * Y_U_WAVE;
    u32x4 A[2][NCB][3];                               // [position][channel block][plane]
    auto load_piece = [&A, ubase, ustep, a_lane](unsigned long long lanes, int step, auto pc, auto plc) {
        constexpr int p = decltype(pc)::value, pl = decltype(plc)::value;
        const float* p0 = wave_uniform((const float*)(ubase + (size_t)step * ustep + p * 12288 + pl * 1024));
        const float* p1 = wave_uniform((const float*)(ubase + (size_t)step * ustep + p * 12288 + pl * 1024 + 6144));
        // (the loop runs with all 64 lanes: EXEC is restored to -1; "+v": under an empty mask the old values stay)
        asm volatile("s_mov_b64 exec, %7\n\tglobal_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:3072\n\t"
                     "global_load_dwordx4 %2, %4, %6\n\tglobal_load_dwordx4 %3, %4, %6 offset:3072\n\ts_mov_b64 exec, -1"
                     : "+v"(A[p][0][pl]), "+v"(A[p][1][pl]), "+v"(A[p][2][pl]), "+v"(A[p][3][pl]) : "v"(a_lane), "s"(p0), "s"(p1), "s"(lanes) : "memory");
    };

    // ---- B (V) role: lane -> (tile = lane & 31 of the 32-tile block, channels 8 (lane >> 5) + j)
    const int tl = lane & 31, cg = lane >> 5;
    // float index of the lane's first input d_0 of channel 8 cg, tile block 0 (see the layout above); tile block 1 = 8 rows (1x5) / 2 tile rows (5x1) on
    const int b_first = VERT ? cg * 2 * (WQ * 4) + (4 * (tl >> 4)) * 20 + (tl & 15) : cg * 8 * 384 + (tl >> 2) * 24 + 4 * (tl & 3) + 4;
    const unsigned rd_lane = smem_lds + (unsigned)b_first * 4u;
    constexpr int TB_OFF = VERT ? 8 * 20 * 4 : 8 * 24 * 4;                         // bytes from tile block 0 to 1
    u32x4 B[2][2][3];                                 // [slot = tile block][position of the pair][plane]
    f32x4 rw00 = {0, 0, 0, 0}, rw01 = rw00, rw10 = rw00, rw11 = rw00;          // 1x5: raw reads of a channel pair: [channel][d0-3 | d4-7]
    unsigned long long rv00 = 0, rv01 = 0, rv02 = 0, rv03 = 0, rv10 = 0, rv11 = 0, rv12 = 0, rv13 = 0;   // 5x1: [channel][(d0,d1) (d2,d3) (d4,d5) (d6,d7)]
    // channel j of the lane's eight: byte offset of its patch
    #define Y_CH(j) (VERT ? (((j) >> 2) * (WQ * 4) + ((j) & 3) * (QCH * 4)) * 4 : (j) * 384 * 4)
    auto issue_reads = [&rw00, &rw01, &rw10, &rw11, &rv00, &rv01, &rv02, &rv03, &rv10, &rv11, &rv12, &rv13, rd_lane](auto qc, auto tbc, unsigned bufoff) {
        constexpr int q = decltype(qc)::value, tb = decltype(tbc)::value;
        const unsigned a = rd_lane + bufoff;
        if (VERT) {
            // eight rows of one column: four ds_read2_b32 (rows k, k + 1: 20 floats apart) per channel, one address add per channel
            const unsigned a0 = a + (unsigned)(Y_CH(2 * q) + tb * TB_OFF), a1 = a + (unsigned)(Y_CH(2 * q + 1) + tb * TB_OFF);
            asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:20" : "=v"(rv00) : "v"(a0));
            asm volatile("ds_read2_b32 %0, %1 offset0:40 offset1:60" : "=v"(rv01) : "v"(a0));
            asm volatile("ds_read2_b32 %0, %1 offset0:80 offset1:100" : "=v"(rv02) : "v"(a0));
            asm volatile("ds_read2_b32 %0, %1 offset0:120 offset1:140" : "=v"(rv03) : "v"(a0));
            asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:20" : "=v"(rv10) : "v"(a1));
            asm volatile("ds_read2_b32 %0, %1 offset0:40 offset1:60" : "=v"(rv11) : "v"(a1));
            asm volatile("ds_read2_b32 %0, %1 offset0:80 offset1:100" : "=v"(rv12) : "v"(a1));
            asm volatile("ds_read2_b32 %0, %1 offset0:120 offset1:140" : "=v"(rv13) : "v"(a1));
        } else {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rw00) : "v"(a), "n"(Y_CH(2 * q) + tb * TB_OFF));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rw01) : "v"(a), "n"(Y_CH(2 * q) + tb * TB_OFF + 16));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rw10) : "v"(a), "n"(Y_CH(2 * q + 1) + tb * TB_OFF));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rw11) : "v"(a), "n"(Y_CH(2 * q + 1) + tb * TB_OFF + 16));
        }
    };
    auto wait_reads = [&]() {                        // (input-only uses after the wait keep every raw register reserved until its read has returned)
        if (VERT) asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(rv00), "v"(rv01), "v"(rv02), "v"(rv03), "v"(rv10), "v"(rv11), "v"(rv12), "v"(rv13) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(rw00), "v"(rw01), "v"(rw10), "v"(rw11) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // the wave's two positions of both channels of the pair: conv_wino1d.hip's B^T rows, the even / odd halves shared by a +- pair
    //   wave 0: v0 = d0 - d6 + 21/4 (d4 - d2)                 v7 = d7 - d1 + 21/4 (d3 - d5)
    //   wave 1: v1 | v2 = (d2 + d6 - 17/4 d4) +- (d1 + d5 - 17/4 d3)
    //   wave 2: v3 | v4 = (d6 + 1/4 d2 - 5/4 d4) +- (1/2 d1 - 5/2 d3 + 2 d5)
    //   wave 3: v5 | v6 = (d6 + 4 d2 - 5 d4) +- (2 d1 - 5/2 d3 + 1/2 d5)
    // as ONE instruction sequence with wave-uniform coefficients (a branch per wave costs more than the three extra instructions:
    // measured, four taken branches per channel made a stage 850-1100 cycles instead of 700):
    //   a = ke0 d0 + ke2 d2 + ke4 d4 + ke6 d6,  b = ko1 d1 + ko3 d3 + ko5 d5 + ko7 d7,  first = a + s b,  second = s a - b   (wave 0: s = 0, b = -v7)
    const float ke0 = wv == 0 ? 1.0f : 0.0f, ke2 = wv == 0 ? -5.25f : wv == 1 ? 1.0f : wv == 2 ? 0.25f : 4.0f;
    const float ke4 = wv == 0 ? 5.25f : wv == 1 ? -4.25f : wv == 2 ? -1.25f : -5.0f, ke6 = wv == 0 ? -1.0f : 1.0f;
    const float ko1 = wv == 0 ? 1.0f : wv == 1 ? 1.0f : wv == 2 ? 0.5f : 2.0f, ko3 = wv == 0 ? -5.25f : wv == 1 ? -4.25f : -2.5f;
    const float ko5 = wv == 0 ? 5.25f : wv == 1 ? 1.0f : wv == 2 ? 2.0f : 0.5f, ko7 = wv == 0 ? -1.0f : 0.0f, ks = wv == 0 ? 0.0f : 1.0f;
    // The vector work is cut into pieces of 3-4 instructions, one piece beside each matrix instruction (stage() below): the wave issues
    // in order, a matrix instruction occupies its pipe for 32 cycles and a vector instruction ~6, so a block of vector instructions
    // between two groups of matrix instructions leaves the pipe idle (measured: transforms outside the groups, 1030 cycles per stage
    // for 768 of matrix work).  The empty asm statements pin every piece where it is written (the optimiser would sink it to its use).
    float tt[2][2];                                   // [channel of the pair][position of the wave's pair]
    float ta[2], tbv[2];                              // even / odd half of a channel
    auto raw_of = [&](auto ec, float (&d)[8]) {
        constexpr int e = decltype(ec)::value;
        if (VERT) {
            auto lo32 = [](unsigned long long u) { return __builtin_bit_cast(float, (unsigned)u); };
            auto hi32 = [](unsigned long long u) { return __builtin_bit_cast(float, (unsigned)(u >> 32)); };
            const unsigned long long q0 = e ? rv10 : rv00, q1 = e ? rv11 : rv01, q2 = e ? rv12 : rv02, q3 = e ? rv13 : rv03;
            d[0] = lo32(q0); d[1] = hi32(q0); d[2] = lo32(q1); d[3] = hi32(q1); d[4] = lo32(q2); d[5] = hi32(q2); d[6] = lo32(q3); d[7] = hi32(q3);
        } else {
            const f32x4 lo = e ? rw10 : rw00, hi = e ? rw11 : rw01;
            d[0] = lo[0]; d[1] = lo[1]; d[2] = lo[2]; d[3] = lo[3]; d[4] = hi[0]; d[5] = hi[1]; d[6] = hi[2]; d[7] = hi[3];
        }
    };
    auto t_even = [&](auto ec) {
        constexpr int e = decltype(ec)::value;
        float d[8]; raw_of(ec, d);
        ta[e] = fmaf(ke6, d[6], fmaf(ke4, d[4], fmaf(ke2, d[2], ke0 * d[0])));
        asm volatile("" : "+v"(ta[e]));
    };
    auto t_odd = [&](auto ec) {
        constexpr int e = decltype(ec)::value;
        float d[8]; raw_of(ec, d);
        tbv[e] = fmaf(ko7, d[7], fmaf(ko5, d[5], fmaf(ko3, d[3], ko1 * d[1])));
        asm volatile("" : "+v"(tbv[e]));
    };
    auto t_fin = [&]() {
#pragma unroll
        for (int e = 0; e < 2; ++e) { tt[e][0] = fmaf(ks, tbv[e], ta[e]); tt[e][1] = fmaf(ks, ta[e], -tbv[e]); }
        asm volatile("" : "+v"(tt[0][0]), "+v"(tt[0][1]), "+v"(tt[1][0]), "+v"(tt[1][1]));
    };
    auto transform = [&]() { t_even(std::integral_constant<int, 0>{}); t_odd(std::integral_constant<int, 0>{}); t_even(std::integral_constant<int, 1>{}); t_odd(std::integral_constant<int, 1>{}); t_fin(); };
    // position i of channel pair q -> element q of the three fragments B[slot][i][plane].  The split rounds to nearest (v_cvt_pk_bf16_f32
    // packs the two channels' bf16 parts in one instruction; the residual x - bf16(x) is exact in f32): x = hi + mid + lo to 2^-27
    // where truncation gives 2^-24, unbiased, for the same 11 instructions per pair as mask / subtract / permute.
    // (The residual as ONE v_dot2_f32_bf16 with the constant (-1, 0) -- exact, tools/probes/valu_rate.hip, and 7 instructions per pair --
    // was tried: the dot instructions run on the matrix pipe, cost 8.5 issue cycles beside matrix instructions instead of 6, need wait
    // states the compiler does not insert around inline asm (wrong results), and the stage got 15% longer.)
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    auto pack2 = [](float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_)); };
    float r1a = 0.0f, r1b = 0.0f, r2a = 0.0f, r2b = 0.0f;
    unsigned pk_h = 0, pk_m = 0;
    auto f_hi = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        pk_h = pack2(tt[0][i], tt[1][i]);
        r1a = tt[0][i] - __builtin_bit_cast(float, pk_h << 16);
        r1b = tt[1][i] - __builtin_bit_cast(float, pk_h & 0xFFFF0000u);
        asm volatile("" : "+v"(pk_h), "+v"(r1a), "+v"(r1b));
    };
    auto f_mid = [&]() {
        pk_m = pack2(r1a, r1b);
        r2a = r1a - __builtin_bit_cast(float, pk_m << 16);
        r2b = r1b - __builtin_bit_cast(float, pk_m & 0xFFFF0000u);
        asm volatile("" : "+v"(pk_m), "+v"(r2a), "+v"(r2b));
    };
    auto f_lo = [&](auto qc, auto slc, auto ic) {
        constexpr int q = decltype(qc)::value, sl = decltype(slc)::value, i = decltype(ic)::value;
        unsigned pl = pack2(r2a, r2b);
        asm volatile("" : "+v"(pl));
        B[sl][i][0][q] = pk_h; B[sl][i][1][q] = pk_m; B[sl][i][2][q] = pl;
    };
    auto finish = [&](auto qc, auto slc, auto ic) { f_hi(ic); f_mid(); f_lo(qc, slc, ic); };

    f32x16 acc[2][NCB][2];                            // [position][channel block][tile block]
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][cb][tb][r] = 0.0f;
    // matrix instruction i of stage (p, tb): the six products, smallest terms first, the four channel blocks in turn; slot = tile block
    auto M = [&](auto pc, auto tbc, auto ic) {
        constexpr int p = decltype(pc)::value, tb = decltype(tbc)::value, i = decltype(ic)::value;
        constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
        constexpr int t = i / NCB, cb = i % NCB;
        acc[p][cb][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[p][cb][pa[t]]), __builtin_bit_cast(bf16x8, B[tb][p][pb[t]]),
                                                                 acc[p][cb][tb], 0, 0, 0);
    };
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1; typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 3> I3;
    typedef std::integral_constant<int, 4> I4; typedef std::integral_constant<int, 5> I5; typedef std::integral_constant<int, 6> I6;
    // One stage = the 24 matrix instructions of (position p, tile block tb), each with one piece of vector work beside it: two channel
    // pairs (q0, q0 + 1) of the OTHER tile block's fragments (slot 1 - tb) are built from the patch buffer at byte offset bufp.  On entry the
    // reads of pair q0 are in flight; those of q0 + 1 are issued in slot 4, those of the next stage's first pair (qn, tbn, bufn) in slot 15
    // (unless `chain` is false: behind the barrier).  vm(k): the memory requests of slot k.
    // A piece of A has landed when at most the YOUNGER ordinary loads are outstanding (they return in order among themselves; any LDS-DMA
    // still in flight only makes the wait longer): first use in tile block 0, position 0: A[1]'s lo and mid are younger (8 loads; its hi
    // is requested behind the wait); position 1: nothing is.
#define Y_FENCE __builtin_amdgcn_sched_barrier(0)
#define Y_SLOT(k, work) do { M(pc, tbc, std::integral_constant<int, k>{}); work; vm(std::integral_constant<int, k>{}); Y_FENCE; } while (0)
    auto stage = [&](auto pc, auto tbc, auto q0c, unsigned bufp, auto qnc, auto tbnc, unsigned bufn, bool chain, auto vm) {
        constexpr int p = decltype(pc)::value, tb = decltype(tbc)::value, q0 = decltype(q0c)::value;
        typedef std::integral_constant<int, 1 - tb> SP; typedef std::integral_constant<int, q0> Q0; typedef std::integral_constant<int, q0 + 1> Q1;
        if (tb == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(p == 0 ? 2 * NCB : 0) : "memory");
        Y_FENCE;
        Y_SLOT(0, (wait_reads(), t_even(I0{})));
        Y_SLOT(1, t_odd(I0{}));
        Y_SLOT(2, t_even(I1{}));
        Y_SLOT(3, t_odd(I1{}));
        Y_SLOT(4, (t_fin(), issue_reads(Q1{}, SP{}, bufp)));
        Y_SLOT(5, f_hi(I0{}));
        Y_SLOT(6, f_mid());
        Y_SLOT(7, f_lo(Q0{}, SP{}, I0{}));
        Y_SLOT(8, f_hi(I1{}));
        Y_SLOT(9, f_mid());
        Y_SLOT(10, f_lo(Q0{}, SP{}, I1{}));
        Y_SLOT(11, (wait_reads(), t_even(I0{})));
        Y_SLOT(12, t_odd(I0{}));
        Y_SLOT(13, t_even(I1{}));
        Y_SLOT(14, t_odd(I1{}));
        Y_SLOT(15, (t_fin(), chain ? issue_reads(qnc, tbnc, bufn) : (void)0));
        Y_SLOT(16, f_hi(I0{}));
        Y_SLOT(17, f_mid());
        Y_SLOT(18, f_lo(Q1{}, SP{}, I0{}));
        Y_SLOT(19, f_hi(I1{}));
        Y_SLOT(20, f_mid());
        Y_SLOT(21, f_lo(Q1{}, SP{}, I1{}));
        Y_SLOT(22, (void)0);
        Y_SLOT(23, (void)0);
    };

    // ---- prologue: DMA(0), A(0) except A[1]'s hi plane (the loop's first stage requests it, as in every step), DMA(1); then tile
    // block 0's fragments of step 0 (slot 0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               // (the zeros are in LDS)
    dma_begin(0, 0u);
    dma_chunk(I0{}, all_lanes); dma_chunk(I1{}, all_lanes); dma_chunk(I2{}, all_lanes); dma_chunk(I3{}, all_lanes);
    dma_chunk(I4{}, all_lanes); dma_chunk(I5{}, all_lanes); dma_chunk(I6{}, all_lanes);
    load_piece(all_lanes, 0, I0{}, I2{}); load_piece(all_lanes, 0, I0{}, I1{}); load_piece(all_lanes, 0, I0{}, I0{});
    load_piece(all_lanes, 0, I1{}, I2{}); load_piece(all_lanes, 0, I1{}, I1{});
    if (nsteps > 1) {
        dma_begin(1, RBUF * 4u);
        dma_chunk(I0{}, all_lanes); dma_chunk(I1{}, all_lanes); dma_chunk(I2{}, all_lanes); dma_chunk(I3{}, all_lanes);
        dma_chunk(I4{}, all_lanes); dma_chunk(I5{}, all_lanes); dma_chunk(I6{}, all_lanes);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) asm volatile("" :: "v"(A[0][cb][0]), "v"(A[0][cb][1]), "v"(A[0][cb][2]), "v"(A[1][cb][1]), "v"(A[1][cb][2]));
    __builtin_amdgcn_s_barrier();
    issue_reads(I0{}, I0{}, 0u); wait_reads(); transform(); issue_reads(I1{}, I0{}, 0u); finish(I0{}, I0{}, I0{}); finish(I0{}, I0{}, I1{});
    wait_reads(); transform(); issue_reads(I2{}, I0{}, 0u); finish(I1{}, I0{}, I0{}); finish(I1{}, I0{}, I1{});
    wait_reads(); transform(); issue_reads(I3{}, I0{}, 0u); finish(I2{}, I0{}, I0{}); finish(I2{}, I0{}, I1{});
    wait_reads(); transform(); issue_reads(I0{}, I1{}, 0u); finish(I3{}, I0{}, I0{}); finish(I3{}, I0{}, I1{});

    // ---- main loop.  Step s: first half = tile block 0 out of slot 0 while slot 1 (tile block 1) is built from raw(s); BARRIER;
    // second half = tile block 1 out of slot 1 while slot 0 of step s+1 is built from raw(s+1).
    //   requests:  stage 0: A(s)[1] hi (behind the wait for A(s)[0]);   stage 1: DMA(s+2) -> the buffer raw(s-1) left at the last barrier;
    //   stages 2, 3: the pieces of A(s+1), each after its last use (lo: slot 6, mid: slot 18, hi: slot 1 of the next stage).
    //   Stage 1 waits for everything (A(s)[1]; the DMAs in flight there are DMA(s+1)'s, requested 3-4 stages earlier and due at the barrier).
    //   Past the end the requests run with an empty EXEC mask.
#ifdef Y3_TIMING
    unsigned long long Tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, Tlast = __builtin_readcyclecounter();
    const unsigned long long Tstart = Tlast;
#endif
    unsigned cur = 0, nxt = nsteps > 1 ? RBUF * 4u : 0u, nn = 2 * RBUF * 4u;          // byte offsets of raw(s), raw(s+1), raw(s+2)
    for (int s = 0; s < nsteps; ++s) {
        const int s1 = s + 1, s2 = s + 2;
        const unsigned long long m1 = s1 < nsteps ? all_lanes : 0ull, m2 = s2 < nsteps ? all_lanes : 0ull;
        const int sa = s1 < nsteps ? s1 : s;                                    // (an address inside the buffer for the masked requests)
        YSTAMP(0);
        stage(I0{}, I0{}, I0{}, cur, I2{}, I1{}, cur, true, [&](auto k) {
            if (decltype(k)::value == 1) load_piece(all_lanes, s, I1{}, I0{}); });
        YSTAMP(1);
        dma_begin(s2 < nsteps ? s2 : s, nn);
        stage(I1{}, I0{}, I2{}, cur, I0{}, I0{}, cur, false, [&](auto k) {
            if (decltype(k)::value == 2) dma_chunk(I0{}, m2);
            if (decltype(k)::value == 5) dma_chunk(I1{}, m2);
            if (decltype(k)::value == 8) dma_chunk(I2{}, m2);
            if (decltype(k)::value == 11) dma_chunk(I3{}, m2);
            if (decltype(k)::value == 14) dma_chunk(I4{}, m2);
            if (decltype(k)::value == 17) dma_chunk(I5{}, m2);
            if (decltype(k)::value == 20) dma_chunk(I6{}, m2); });
        YSTAMP(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        YSTAMP(3);
        issue_reads(I0{}, I0{}, nxt);
        __builtin_amdgcn_sched_barrier(0);
        stage(I0{}, I1{}, I0{}, nxt, I2{}, I0{}, nxt, true, [&](auto k) {
            if (decltype(k)::value == 6) load_piece(m1, sa, I0{}, I2{});
            if (decltype(k)::value == 18) load_piece(m1, sa, I0{}, I1{}); });
        YSTAMP(4);
        stage(I1{}, I1{}, I2{}, nxt, I0{}, I1{}, nxt, true, [&](auto k) {
            if (decltype(k)::value == 1) load_piece(m1, sa, I0{}, I0{});
            if (decltype(k)::value == 6) load_piece(m1, sa, I1{}, I2{});
            if (decltype(k)::value == 18) load_piece(m1, sa, I1{}, I1{}); });
        YSTAMP(5);
        const unsigned t = cur; cur = nxt; nxt = nn; nn = t;
    }
#ifdef Y3_TIMING
    const unsigned long long Tloop = __builtin_readcyclecounter();
#endif
    // The requests of the step past the end are in flight and their results dead: keep their registers reserved (input-only uses AFTER
    // the wait) until they have landed.
    if (VERT) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" :: "v"(rv00), "v"(rv01), "v"(rv02), "v"(rv03), "v"(rv10), "v"(rv11), "v"(rv12), "v"(rv13) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" :: "v"(rw00), "v"(rw01), "v"(rw10), "v"(rw11) : "memory");
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) asm volatile("" :: "v"(A[p][cb][0]), "v"(A[p][cb][1]), "v"(A[p][cb][2]));
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                                                     // nobody reads or fills the patch buffers any more

    // ---- epilogue, per 64-channel half hf: the wave's pair to LDS, planes [2 w + i][co 64][tile 64]: wave 0 (m0, m7); the others
    // (m+ + m-, m+ - m-); then every thread finishes 16 output quads:
    //   y0 = m0 + (m1+m2) + (m3+m4) + (m5+m6);   y1 = (m1-m2) + 2 (m3-m4) + 1/2 (m5-m6);
    //   y2 = (m1+m2) + 4 (m3+m4) + 1/4 (m5+m6);   y3 = (m1-m2) + 8 (m3-m4) + 1/8 (m5-m6) + m7;  and the gate arithmetic of conv_wino1d.hip.
    // D layout of a 32x32 block: column (tile) = lane & 31, row (channel) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    const int mode = EPI == 1 ? RPE_CONV_GATE_ZR : EPI == 2 ? RPE_CONV_GATE_H : P.mode, cgt = P.cgate;
    const float* addb = P.add ? P.add + (size_t)bz * P.abs_ : nullptr;
    const float* hb = P.hid ? P.hid + (size_t)bz * P.hbs : nullptr;
    const float* zgb = P.z ? P.z + (size_t)bz * P.zbs : nullptr;
    float* outb = P.out + (size_t)bz * P.obs;
    float* out2b = P.out2 ? P.out2 + (size_t)bz * P.o2bs : nullptr;
    struct QuadOps { f32x4 a, h, z; float bi; };
    // quad j of the thread:  1x5: (channel fcol + 4 j, the thread's tile);  5x1: (channel fcol + 16 (j >> 2), row j & 3 of the thread's four tiles)
    const int e_tile = tid & 63, e_cq = tid & 3, e_tr = (tid >> 2) & 3;
    const int e_fcol = VERT ? tid >> 4 : tid >> 6;
    const int e_ox = VERT ? x0 + 4 * e_cq : x0 + 4 * (e_tile & 3), e_oy = VERT ? y0 + 4 * e_tr : y0 + (e_tile >> 2);
    const int e_oxc = e_ox < W ? e_ox : W - 4;                                       // (clamped: the fast paths fetch unconditionally)
    auto quad_of = [&](int j, int& col, int& oy) { col = VERT ? e_fcol + 16 * (j >> 2) : e_fcol + 4 * j; oy = VERT ? e_oy + (j & 3) : e_oy; };
    auto fetch = [&](int hf, int c, QuadOps (&o)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int col, oy;
            quad_of(4 * c + r, col, oy);
            const int co = co0 + 64 * hf + col;
            o[r].bi = 0.0f;
            if (EPI != 0) {
                // operands of an out-of-range quad come from the nearest valid one (never stored)
                const int coc = co < P.cout ? co : P.cout - 1, oyc = oy < H ? oy : H - 1;
                const size_t px = (size_t)oyc * W + e_oxc;
                o[r].a = *(const f32x4*)(addb + (size_t)coc * hw + px);
                if (EPI == 1) { o[r].h = o[r].a; if (co0 + 64 * hf + 63 >= cgt) o[r].h = *(const f32x4*)(hb + (size_t)(coc >= cgt ? coc - cgt : coc) * hw + px); o[r].z = o[r].a; }
                else { o[r].z = *(const f32x4*)(zgb + (size_t)coc * hw + px); o[r].h = *(const f32x4*)(hb + (size_t)coc * hw + px); }
                continue;
            }
            o[r].a = o[r].h = o[r].z = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            if (co >= P.cout || oy >= H || e_ox >= W) continue;
            const size_t e0 = (size_t)co * hw + (size_t)oy * W + e_ox;
            if (P.bias) o[r].bi = P.bias[co];
            if (addb) o[r].a = *(const f32x4*)(addb + e0);
            if (mode == RPE_CONV_GATE_ZR) { if (co >= cgt) o[r].h = *(const f32x4*)(hb + e0 - (size_t)cgt * hw); }
            else if (mode == RPE_CONV_GATE_H) { o[r].z = *(const f32x4*)(zgb + e0); o[r].h = *(const f32x4*)(hb + e0); }
        }
    };
    // one output quad (four x-neighbouring pixels): addend, bias, gate, store
    auto emit = [&](int hf, int col, int oy, f32x4 v, const QuadOps& q) {
        const int co = co0 + 64 * hf + col;
        const bool ok = co < P.cout && oy < H && e_ox < W;
        const size_t e0 = (size_t)co * hw + (size_t)oy * W + e_ox;
        v += q.a;
        if (EPI == 0) v += q.bi;
        if (mode == RPE_CONV_GATE_ZR) {
            f32x4 sg;
#pragma unroll
            for (int i = 0; i < 4; ++i) sg[i] = sigmoid_f(v[i]);
            const bool second = co >= cgt;                                           // z -> out; r * h -> out2 (behind the cgate channels of z)
            float* dst = second ? out2b + (e0 - (size_t)cgt * hw) : outb + e0;
            const f32x4 val = second ? sg * q.h : sg;
            if (ok) *(f32x4*)dst = val;
        } else if (mode == RPE_CONV_GATE_H) {
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (1.0f - q.z[i]) * q.h[i] + q.z[i] * tanh_f(v[i]);
            if (ok) *(f32x4*)(outb + e0) = o;
        } else {
            if (mode == RPE_CONV_RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] < 0.0f ? 0.0f : v[i];
            }
            if (ok) {
                *(f32x4*)(outb + e0) = v;
                if (out2b) *(f32x4*)(out2b + e0) = v;
            }
        }
    };
    // planes: 0 m0, 1 m7, 2 m1+m2, 3 m1-m2, 4 m3+m4, 5 m3-m4, 6 m5+m6, 7 m5-m6
    auto finish_quads = [&](int hf, int c, const QuadOps (&o)[4]) {
        if (VERT) {
            int col, oy;
            quad_of(4 * c, col, oy);
            const float* zr = &smem[col * 64 + e_tr * 16 + 4 * e_cq];
            f32x4 m[8];
#pragma unroll
            for (int pl = 0; pl < 8; ++pl) m[pl] = *(const f32x4*)(zr + pl * 4096);
            f32x4 yv[4];
            yv[0] = ((m[0] + m[2]) + m[4]) + m[6];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                yv[1][i] = fmaf(0.5f, m[7][i], fmaf(2.0f, m[5][i], m[3][i]));
                yv[2][i] = fmaf(0.25f, m[6][i], fmaf(4.0f, m[4][i], m[2][i]));
                yv[3][i] = m[1][i] + fmaf(0.125f, m[7][i], fmaf(8.0f, m[5][i], m[3][i]));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) emit(hf, col, oy + r, yv[r], o[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int col, oy;
                quad_of(4 * c + r, col, oy);
                const float* zr = &smem[col * 64 + e_tile];
                float m[8];
#pragma unroll
                for (int pl = 0; pl < 8; ++pl) m[pl] = zr[pl * 4096];
                const f32x4 v = {((m[0] + m[2]) + m[4]) + m[6], fmaf(0.5f, m[7], fmaf(2.0f, m[5], m[3])), fmaf(0.25f, m[6], fmaf(4.0f, m[4], m[2])),
                                 m[1] + fmaf(0.125f, m[7], fmaf(8.0f, m[5], m[3]))};
                emit(hf, col, oy, v, o[r]);
            }
        }
    };
    const float ksn = wv == 0 ? 1.0f : -1.0f;
    auto exchange = [&](auto hfc) {
        constexpr int hf = decltype(hfc)::value;
        float* zb = &smem[(wv * 2 * 64 + 4 * cg) * 64 + tl];
#pragma unroll
        for (int cbl = 0; cbl < 2; ++cbl)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float ma = acc[0][2 * hf + cbl][tb][r], mb = acc[1][2 * hf + cbl][tb][r];
                    const int col = 32 * cbl + (r & 3) + 8 * (r >> 2);
                    zb[col * 64 + 32 * tb] = fmaf(ks, mb, ma);                      // wave 0: ma | ma + mb
                    zb[(64 + col) * 64 + 32 * tb] = fmaf(ks, ma, ksn * mb);          // wave 0: mb | ma - mb
                }
    };
#ifdef Y3_TIMING
    unsigned long long Te[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define ESTAMP(i) Te[i] = __builtin_readcyclecounter()
#else
#define ESTAMP(i)
#endif
    // The pass is bound by what the memory system delivers for 64-byte pieces of 20 KB planes, not by latency or instruction count
    // (~30 thousand cycles of its 76): three chunks ahead instead of one made the requests wait at ISSUE (22 thousand cycles for 36
    // requests); touching the operands during the last steps of the loop, one dword per piece, moved the same wait into the loop (final
    // pass 54, loop 73 -> 105 thousand); starting the first round's workgroups in four phases changed nothing.  One workgroup per CU
    // has nothing to run beside its final pass -- the f32 kernel's two hide it.
    QuadOps oa[4], ob[4];
    auto half = [&](auto hfc) {
        constexpr int hf = decltype(hfc)::value;
        if (co0 + 64 * hf >= P.cout) return;                                          // (workgroup-uniform: a channel tile's empty upper half)
        fetch(hf, 0, oa);
        if (hf) __syncthreads();                                                      // (the first half has been read)
        if (!hf) ESTAMP(0);
        exchange(hfc);
        if (!hf) ESTAMP(1);
        __syncthreads();
        if (!hf) ESTAMP(2);
        fetch(hf, 1, ob); finish_quads(hf, 0, oa);
        if (!hf) ESTAMP(3);
        fetch(hf, 2, oa); finish_quads(hf, 1, ob);
        fetch(hf, 3, ob); finish_quads(hf, 2, oa);
        finish_quads(hf, 3, ob);
        if (!hf) ESTAMP(4);
    };
    half(I0{});
    half(I1{});
#undef Y_CH
#ifdef Y3_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (blockIdx.x == gridDim.x / 2 && blockIdx.z == 0 && blockIdx.y == gridDim.y / 2 && tid == 0) {
        for (int i = 0; i < 8; ++i) g_y3_timing[i] = Tacc[i];
        g_y3_timing[8] = Tstart - Tk0; g_y3_timing[9] = Tloop - Tstart; g_y3_timing[10] = __builtin_readcyclecounter() - Tloop; g_y3_timing[11] = nsteps;
        g_y3_timing[12] = Te[0] - Tloop; g_y3_timing[13] = Te[1] - Te[0]; g_y3_timing[14] = Te[2] - Te[1]; g_y3_timing[15] = Te[3] - Te[2]; g_y3_timing[7] = Te[4] - Te[3];
    }
#endif
}

// weight (cout, cin, 5 taps) -> U = G g (conv_wino1d.hip's G, evaluated in f64 and rounded once to f32, like k_wino1d_pack), then the exact
// three-way bf16 split, laid out [step = ci/16][co tile = co/128][wave][i][cb = (co%128)/32][plane][co%32][ci%16], positions of wave w:
// (0, 7), (1, 2), (3, 4), (5, 6)
__global__ void k_wino1d_pack_x3(const float* __restrict__ w, unsigned short* __restrict__ wp, int cout, int cin, int coP, long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;              // over [step][co tile][wave][i][cb][co32][ci16]
    if (e >= total) return;
    const int ci16 = (int)(e & 15), co32 = (int)((e >> 4) & 31), cb = (int)((e >> 9) & 3), i = (int)((e >> 11) & 1), wvv = (int)((e >> 12) & 3);
    const long long rest = e >> 14;
    const int nct = coP / Y_CO;
    const int co = (int)(rest % nct) * Y_CO + cb * 32 + co32, ci = (int)(rest / nct) * YK + ci16;
    const int pos = wvv == 0 ? (i ? 7 : 0) : 2 * wvv - 1 + i;
    double v = 0.0;
    if (co < cout && ci < cin) {
        const float* g = w + ((size_t)co * cin + ci) * 5;
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
    // round-to-nearest-even bf16 parts (finite values): hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid), the residuals exact in f32
    auto bf16_rne = [](float f) { unsigned b = __builtin_bit_cast(unsigned, f); b += 0x7FFFu + ((b >> 16) & 1u); return b & 0xFFFF0000u; };
    const float vf = (float)v;
    const unsigned u = bf16_rne(vf);
    const float r1 = vf - __builtin_bit_cast(float, u);
    const unsigned u1 = bf16_rne(r1);
    const float r2 = r1 - __builtin_bit_cast(float, u1);
    unsigned short* d = wp + (e >> 9) * (3 * 512) + co32 * 16 + ci16;
    d[0] = (unsigned short)(u >> 16); d[512] = (unsigned short)(u1 >> 16); d[1024] = (unsigned short)(bf16_rne(r2) >> 16);
}

static inline int y_cop(int cout) { return (cout + Y_CO - 1) / Y_CO * Y_CO; }

extern "C" size_t rpe_conv_wino1d_x3_packed_bytes(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cin % YK) return 0;
    return (size_t)(cin / YK) * (y_cop(cout) / Y_CO) * 4 * Y_U_WAVE;
}

extern "C" int rpe_conv_wino1d_x3_pack(const float* weight, int cout, int cin, void* packed, void* stream) {
    if (!weight || !packed || cout <= 0 || cin <= 0) return RPE_E_BADARG;
    if (cin % YK) return RPE_E_UNSUPPORTED;
    const long long total = (long long)cin * y_cop(cout) * 8;
    hipLaunchKernelGGL(k_wino1d_pack_x3, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, (unsigned short*)packed, cout, cin, y_cop(cout), total);
    return rpe_check_launch();
}

extern "C" int rpe_conv_wino1d_x3(const rpe_conv_desc* d, void* stream) {
    if (!d || !d->x || !d->packed || !d->out || d->b <= 0 || d->cin <= 0 || d->cout <= 0 || d->h <= 0 || d->w <= 0) return RPE_E_BADARG;
    const bool vert = d->kh == 5 && d->kw == 1, horiz = d->kh == 1 && d->kw == 5;
    if (!(vert || horiz) || (d->stride != 0 && d->stride != 1) || (d->cin % YK) || (d->w & 3)) return RPE_E_UNSUPPORTED;
    if (d->mode < RPE_CONV_LINEAR || d->mode > RPE_CONV_GATE_H) return RPE_E_BADARG;
    if (d->mode == RPE_CONV_GATE_ZR && (!d->out2 || !d->hidden || d->gate_channels <= 0 || d->cout != 2 * d->gate_channels)) return RPE_E_BADARG;
    if (d->mode == RPE_CONV_GATE_H && (!d->hidden || !d->zgate)) return RPE_E_BADARG;
    if (d->scale || d->residual || d->stats || d->pre_norm) return RPE_E_UNSUPPORTED;
    auto a16 = [](const void* p, long long bs) { return !p || ((((uintptr_t)p) & 15) == 0 && (bs & 3) == 0); };
    if (!a16(d->x, d->x_batch_stride) || !a16(d->packed, 0) || ((d->h * d->w) & 3)) return RPE_E_UNSUPPORTED;
    if ((!a16(d->out, d->out_batch_stride) || !a16(d->out2, d->out2_batch_stride) || !a16(d->add, d->add_batch_stride) ||
         !a16(d->hidden, d->hidden_batch_stride) || !a16(d->zgate, d->zgate_batch_stride))) return RPE_E_UNSUPPORTED;
    W1X3P P;
    P.x = d->x; P.xbs = d->x_batch_stride; P.wp = (const unsigned short*)d->packed; P.cin = d->cin; P.cout = d->cout; P.coP = y_cop(d->cout);
    P.H = d->h; P.W = d->w; P.bias = d->bias; P.add = d->add; P.abs_ = d->add_batch_stride;
    P.out = d->out; P.obs = d->out_batch_stride; P.out2 = d->out2; P.o2bs = d->out2_batch_stride;
    P.hid = d->hidden; P.hbs = d->hidden_batch_stride; P.z = d->zgate; P.zbs = d->zgate_batch_stride; P.cgate = d->gate_channels; P.mode = d->mode;
    const unsigned gx = ceil_div(d->w, 16) * ceil_div(d->h, 16);
    const dim3 grid(gx, d->b, P.coP / Y_CO);
    // the GRU's launches (gates with an addend, no bias) take the branch-free final pass
    const int epi = (d->add && !d->bias) ? (d->mode == RPE_CONV_GATE_ZR ? 1 : d->mode == RPE_CONV_GATE_H ? 2 : 0) : 0;
#define Y_LAUNCH(V, E) hipLaunchKernelGGL((k_conv_wino1d_x3<V, E>), grid, dim3(256), 0, (hipStream_t)stream, P)
    if (vert) { if (epi == 1) Y_LAUNCH(true, 1); else if (epi == 2) Y_LAUNCH(true, 2); else Y_LAUNCH(true, 0); }
    else { if (epi == 1) Y_LAUNCH(false, 1); else if (epi == 2) Y_LAUNCH(false, 2); else Y_LAUNCH(false, 0); }
#undef Y_LAUNCH
    return rpe_check_launch();
}
