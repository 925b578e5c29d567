// LABELLED VARIANT (never in the headline number): the 3x3 stride-1 Winograd F(2x2,3x3) convolutions of conv_wino.hip on the 16-bit
// matrix cores, every f32 product evaluated as SIX bf16 products of an exact three-way split.
//
// Replaces the same reference layers as conv_wino.hip (the reference's RAFT submodule, call sites core/pose/pose_net.py:47,65,129:
// core/RAFT/core/update.py BasicMotionEncoder.convc2 / convf2 / conv, FlowHead.conv1; core/RAFT/core/extractor.py residual blocks).
//
// Arithmetic.  x = hi + mid + lo to 2^-27 |x| (three round-to-nearest bf16 parts, hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid),
// the residuals exact in f32; the first version truncated: exact to 2^-24 but biased, maximum error up to 1.3x the f32 kernel's), and
//     u v ~= uh vh + (uh vm + um vh) + (uh vl + ul vh + um vm),        f32 accumulation in the matrix core;
// the dropped terms (um vl, ul vm, ul vl) are below 2^-24 |u v|, the rounding of the f32 product itself (k_corr_build_x3 measured this
// scheme at 1.0-1.08x the f32 pipe's RMS error).  U = G g G^T is transformed in f32 and split ONCE at pack time; V = B^T d B
// is transformed in f32 (bit-identical to conv_wino.hip's transform) and split in registers.  Six v_mfma_f32_32x32x16_bf16 (32 cycles)
// replace eight v_mfma_f32_16x16x4_f32 x 2 (512 cycles) per 32 x 32 x 16 block: 3/8 of the matrix time.
//
// Why the kernel is NOT conv_wino.hip with another instruction: at 3/8 of the matrix time the f32 kernel's tiling (64 channels x 32
// tiles, two workgroups per CU) would need 71 B/cycle/CU of pre-split weights from L2 (U is 96 B per (co, ci): 16 positions x 3 planes)
// and 75 % of the LDS read bandwidth.  So:
//   * workgroup = 4 waves = ONE per SIMD (512 registers each), 64 output channels x 64 tiles (a 16 x 16 pixel patch): U traffic 32
//     B/cycle/CU at full matrix rate;
//   * wave xi owns the four Winograd positions (xi, nu = 0..3) for ALL 64 channels x 64 tiles: 4 x 2 x 2 blocks of 32 x 32 = 256
//     accumulators (the AGPR half of the register file).  No operand is shared between waves, so neither U nor V ever sits in LDS:
//       - A (U) fragments: global -> VGPR, 16 bytes per lane, each 1 KB block read exactly once per workgroup, issued four stages ahead;
//       - B (V) fragments: built in registers from the raw input patch in LDS (the only LDS operand: rows shared through the halo),
//         row pass t = d[rA] +- d[rB] (the wave's xi picks the rows), column pass, split, v_perm_b32 packing of channel pairs;
//   * the raw patch (16 channels x 18 rows x 24 floats per step) arrives by LDS-DMA into two buffers, one barrier per step;
//   * epilogue: each wave reduces its four positions to Z[xi][j] = sum_nu M[xi][nu] A^T[j][nu] (2 values), the waves exchange Z through
//     LDS (128 KB, aliasing the patch buffers), and every thread finishes Y[i][j] = sum_xi A^T[i][xi] Z[xi][j] for (channel, tile row,
//     tile pair): 16-byte stores.  Bias / scale / ReLU / residual / second output / instance-norm moments as conv_wino.hip; the moment
//     records keep conv_wino.hip's regions and layout (one per 16 x 4 pixels), so the consumers do not know which kernel ran.
#include "wino_common.h"
#include <type_traits>
#pragma clang diagnostic ignored "-Wunused-lambda-capture"    // (issue_reads names its captures: clang does not capture variables used only in asm operands of a generic lambda)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define X_CO 64                                  // output channels per workgroup
#define XK 16                                    // input channels per step (K of the 32x32x16 matrix instruction)
#define X_QCH 108                                // 16-byte quads per channel of the raw patch: 18 rows x 6 (map columns x0 - 4 .. x0 + 19)
#define X_WQ 448                                 // quad slots per wave and buffer (4 channels = 432 quads + 16 slack: seven whole DMA rounds)
#define X_RBUF (4 * X_WQ * 4)                    // floats per raw buffer (28 KB)
#define X_CHF 432                                // floats per channel
#define X_U_WAVE 24576                           // bytes of U per (step, 64-channel tile, xi): [nu 4][cb 2][plane 3][32 co][16 ci] bf16

#ifdef X3_TIMING
// Experiment hook (tools/build_variant.sh ... -DX3_TIMING): s_memtime stamps at the stage boundaries of the main loop, summed over the
// steps of one mid-grid workgroup's wave 0; read back with rpe_debug_x3_timing.
__device__ unsigned long long g_x3_timing[24];
extern "C" int rpe_debug_x3_timing(unsigned long long* out24) { return hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_x3_timing), 192) == hipSuccess ? 0 : -1; }
#define XSTAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); Tacc[i] += now_ - Tlast; Tlast = now_; } while (0)
#else
#define XSTAMP(i)
#endif
struct WinoX3P {
    const float* x; long long xbs;
    const unsigned short* wp; int cin, cout, coP, H, W;
    const float* bias;
    float* out; long long obs;
    float* out2; long long o2bs;
    int mode;
    const float* scale; const float* res; long long rbs; float* stats; const float* pre;
    int co_base, nrec;                            // first output channel of this launch | moment records per plane (conv_wino.hip's count)
};

// seven 1 KB chunks: global (base - 3072) + v_k + 1024 (k & 3)  ->  LDS lds_addr + 1024 k + lane * 16   (the instruction offset applies to both
// addresses; the caller folds 3072 - 1024 (k & 3) into v_k)
__device__ __forceinline__ void dma_raw7(const float* base, const unsigned (&v)[7], unsigned lds_addr) {
    unsigned keep;
    base = wave_uniform(base - 768);
    const unsigned lds2 = lds_addr + 4096u;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %9\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %8\n\tglobal_load_lds_dwordx4 %2, %8 offset:1024\n\t"
                 "global_load_lds_dwordx4 %3, %8 offset:2048\n\tglobal_load_lds_dwordx4 %4, %8 offset:3072\n\t"
                 "s_mov_b32 m0, %10\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %5, %8\n\tglobal_load_lds_dwordx4 %6, %8 offset:1024\n\t"
                 "global_load_lds_dwordx4 %7, %8 offset:2048\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "s"(base), "s"(lds_addr), "s"(lds2) : "memory");
}

// EPI: 0 bias / ReLU / out2 (update block); 1 + scale and residual (cnet); 2 + moments (fnet); 3 everything at run time.
// PRE: the input is the RAW output of an instance-normalised convolution: relu((x - mean) / std) is applied to the landed patch in LDS.
// NCB: 32-channel blocks per workgroup (2 = 64 channels; 1 = a trailing 32-channel tile).
template <int EPI, bool PRE, int NCB>
__global__ __launch_bounds__(256, 1) void k_conv_wino_x3(WinoX3P P) {
    constexpr bool HAS_AFFINE = EPI == 1 || EPI == 3, HAS_STATS = EPI == 2 || EPI == 3;
    // main loop: two raw buffers (56 KB) + the (-mean / std, 1 / std) table; epilogue: Z[xi 4][j 2][co 64][tile 64] floats = 128 KB over both
    __shared__ __attribute__((aligned(16))) float smem[32768];
    constexpr int PN_AT = 2 * X_RBUF + 16, ROFF_AT = PN_AT + 256;         // (floats; ROFF_AT .. + 2048: the DMA offsets of the 256 threads)
    asm volatile("" :: "s"(P.x), "s"(P.wp), "s"(P.out), "s"(P.bias), "s"(P.xbs), "s"(P.obs), "s"(P.cin), "s"(P.cout), "s"(P.coP), "s"(P.H),
                 "s"(P.W), "s"(P.co_base), "s"(P.mode), "s"(P.out2), "s"(P.o2bs));
#ifdef X3_TIMING
    const unsigned long long Tk0 = __builtin_readcyclecounter();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ptx = (P.W + 15) >> 4;
    const int pid = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;     // XCD-contiguous
    const int px = pid % ptx, py = pid / ptx;
    const int x0 = px * 16, y0 = py * 16;
    // grid = (patches, images, channel tiles): the workgroups in flight share ONE channel tile's U (1.5 MB at cin = 256: L2-resident in every
    // XCD); with the channel tile as the middle dimension three tiles' 4.7 MB thrashed the 4 MB L2s and the loop waited for its A fragments
    const int co0 = P.co_base + blockIdx.z * (32 * NCB), bz = blockIdx.y;
    const int H = P.H, W = P.W, hw = H * W;
    const float* xb = P.x + (size_t)bz * P.xbs;
    const int nsteps = P.cin / XK;

    // ---- DMA role: wave w brings channels 4w .. 4w+3 of the step: logical quad lq = 64 k + lane (< 432) -> (channel, row, quad column);
    // out-of-map quads are never requested (the lanes are masked out of the DMA instruction, their slots zeroed once) -- with the
    // loader-side norm (PRE) they read a clamped in-map quad and are overwritten after landing with the rest of the patch-up (fix_raw).
    // Lanes past 432 repeat the last quad into the wave's 16 slack slots.  The seven per-lane offsets live in LDS (ROFF_AT; two 16-byte reads per step): the loop has no
    // vector register to spare.
    unsigned oob = 0;
    unsigned long long msk[7];                   // lanes of chunk k whose quad lies inside the map (the DMA request runs under this mask)
    {
        unsigned roff[8];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int lq0 = 64 * k + lane, lq = lq0 < 432 ? lq0 : 431;
            const int cl = lq / X_QCH, rem = lq - cl * X_QCH, r = rem / 6, qc = rem - r * 6;
            int yy = y0 - 1 + r, xx = x0 - 4 + 4 * qc;
            const bool out = lq0 < 432 && (yy < 0 || yy >= H || xx < 0 || xx >= W);
            if (out) oob |= 1u << k;
            msk[k] = __ballot(!out);
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy); xx = xx < 0 ? 0 : (xx >= W ? W - 4 : xx);
            roff[k] = (unsigned)((4 * wv + cl) * hw + yy * W + xx) * 4u + 3072u - 1024u * (k & 3);
        }
        roff[7] = 0u;
        *(u32x4*)&smem[ROFF_AT + 8 * tid] = (u32x4){roff[0], roff[1], roff[2], roff[3]};
        *(u32x4*)&smem[ROFF_AT + 8 * tid + 4] = (u32x4){roff[4], roff[5], roff[6], roff[7]};
    }
    const bool border = (y0 < 1) | (y0 + 16 >= H) | (x0 < 4) | (x0 + 16 >= W);              // workgroup-uniform
    if (!PRE && border) {                                 // the wave's own slots of both buffers: zeros for the quads no request ever writes
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int k = 0; k < 7; ++k) *(f32x4*)&smem[b * X_RBUF + (wv * X_WQ + 64 * k + lane) * 4] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    const unsigned smem_lds = lds_addr_of(&smem[0]);
    const unsigned rs_base = smem_lds + (unsigned)wv * (X_WQ * 16u) + 4u;                  // (+4: patch column 0 lands on an 8-byte boundary)
    const size_t rstep = (size_t)XK * hw;
    auto dma_raw = [&](int step, int buf) {
        const u32x4 r0 = *(const u32x4*)&smem[ROFF_AT + 8 * tid], r1 = *(const u32x4*)&smem[ROFF_AT + 8 * tid + 4];     // (the thread's own words)
        const unsigned rr[7] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2]};
#ifdef X3_DMA_HOT
        step = 0;                                                                      // ablation: every step re-reads step 0's patch
#endif
        dma_raw7(xb + (size_t)step * rstep, rr, rs_base + (unsigned)buf * (X_RBUF * 4u));
    };
    // the same seven chunks one at a time (main loop: one per matrix-instruction group)
    unsigned rr_[7] = {0, 0, 0, 0, 0, 0, 0};
    const float* dma_src = xb;
    unsigned dma_lds = 0;
    auto dma_begin = [&](int step, int buf) {
        const u32x4 r0 = *(const u32x4*)&smem[ROFF_AT + 8 * tid], r1 = *(const u32x4*)&smem[ROFF_AT + 8 * tid + 4];
        rr_[0] = r0[0]; rr_[1] = r0[1]; rr_[2] = r0[2]; rr_[3] = r0[3]; rr_[4] = r1[0]; rr_[5] = r1[1]; rr_[6] = r1[2];
#ifdef X3_DMA_HOT
        step = 0;
#endif
        dma_src = wave_uniform(xb + (size_t)step * rstep - 768);
        dma_lds = rs_base + (unsigned)buf * (X_RBUF * 4u);
    };
    auto dma_chunk = [&rr_, &dma_src, &dma_lds, &msk](auto kc, unsigned long long lanes) {
        constexpr int k = decltype(kc)::value;
        unsigned keep;
        if (!PRE) lanes &= msk[k];                                       // (out-of-map quads are never written: their slots keep the zeros of the prologue)
        const unsigned la = dma_lds + (k >= 4 ? 4096u : 0u);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_mov_b64 exec, %5\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b64 exec, -1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(rr_[k]), "s"(dma_src), "s"(la), "n"((k & 3) * 1024), "s"(lanes) : "memory");
    };
    // after landing: the lane patches ITS quads (no barrier needed in front): padding, and with PRE relu((x - mean) / std) in place
    auto fix_raw = [&](int step, int buf) {
        if (!PRE) return;                                                // (without the loader-side norm nothing is patched: masked requests, zeroed slots)
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int lq0 = 64 * k + lane;
            float* q4 = &smem[buf * X_RBUF + (wv * X_WQ + lq0) * 4 + 1];
            if (PRE) {
                if (lq0 < 432) {
                    const int cl = lq0 / X_QCH;
                    const float2 pn = *(const float2*)&smem[PN_AT + 2 * (step * XK + 4 * wv + cl)];
                    float v0 = q4[0], v1 = q4[1], v2 = q4[2], v3 = q4[3];
                    v0 = fmaxf(fmaf(v0, pn.y, pn.x), 0.0f); v1 = fmaxf(fmaf(v1, pn.y, pn.x), 0.0f);
                    v2 = fmaxf(fmaf(v2, pn.y, pn.x), 0.0f); v3 = fmaxf(fmaf(v3, pn.y, pn.x), 0.0f);
                    if ((oob >> k) & 1) { v0 = 0.0f; v1 = 0.0f; v2 = 0.0f; v3 = 0.0f; }
                    q4[0] = v0; q4[1] = v1; q4[2] = v2; q4[3] = v3;
                }
            } else if ((oob >> k) & 1) { q4[0] = 0.0f; q4[1] = 0.0f; q4[2] = 0.0f; q4[3] = 0.0f; }
        }
    };

    // ---- A (U) role: the wave's 24 KB of a step: [nu][cb][plane][32 co][16 ci]; lane -> (co = lane & 31, k half = lane >> 5).
    // Requested by inline asm and waited for by hand (wait_a): left to the compiler the loads sink to their first use (no prefetch).
    // Ordinary loads and LDS-DMA return in order, so the one counter serves both.
    const unsigned a_lane = (unsigned)(lane & 31) * 32u + (unsigned)(lane >> 5) * 16u;
    const int nct = P.coP / X_CO;
    const char* ubase = (const char*)P.wp + ((size_t)(co0 / X_CO) * 4 + wv) * X_U_WAVE + (size_t)((co0 % X_CO) / 32) * 3072;
    const size_t ustep = (size_t)nct * 4 * X_U_WAVE;
    u32x4 A[4][NCB][3];
    // one plane of A[nu] (both channel blocks): requested piece by piece, spread over the matrix-instruction groups (a block of six
    // requests stalls the wave's issue ~30 cycles each while the matrix pipe drains)
    auto load_piece_if = [&A, ubase, ustep, a_lane](unsigned long long lanes, int step, auto nuc, auto plc) {
        constexpr int nu = decltype(nuc)::value, pl = decltype(plc)::value;
#ifdef X3_A_HOT
        step = 0;                                                                      // ablation: every step re-reads step 0's fragments
#endif
        const float* p0 = wave_uniform((const float*)(ubase + (size_t)step * ustep + nu * 6144 + pl * 1024));
        // (the loop runs with all 64 lanes: EXEC is restored to -1; "+v": under an empty mask the old values stay)
        if (NCB == 2) asm volatile("s_mov_b64 exec, %4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:3072\n\ts_mov_b64 exec, -1"
                                   : "+v"(A[nu][0][pl]), "+v"(A[nu][NCB - 1][pl]) : "v"(a_lane), "s"(p0), "s"(lanes) : "memory");
        else asm volatile("s_mov_b64 exec, %3\n\tglobal_load_dwordx4 %0, %1, %2\n\ts_mov_b64 exec, -1" : "+v"(A[nu][0][pl]) : "v"(a_lane), "s"(p0), "s"(lanes) : "memory");
    };
    const unsigned long long all_lanes = __ballot(true);           // (a register pair, not a 64-bit literal: s_mov_b64 takes no such immediate)
    auto load_piece = [&](int step, auto nuc, auto plc) { load_piece_if(all_lanes, step, nuc, plc); };
    auto load_a = [&](int step, auto nuc) {
        load_piece(step, nuc, std::integral_constant<int, 2>{}); load_piece(step, nuc, std::integral_constant<int, 1>{}); load_piece(step, nuc, std::integral_constant<int, 0>{});
    };
    // Request order of the ordinary loads (pieces of NCB loads; lo, mid, hi = planes 2, 1, 0 in the order they fall dead in a stage):
    //   A'[0] lo mid | hi  A'[1] lo mid | hi  A'[2] lo mid | hi  A'[3] lo mid || next step: A'[3] hi
    // (stages (0,tb1) | (1,tb1) | (2,tb1) | (3,tb1) || (0,tb0) of the next step).  A[nu] has landed when at most `younger` pieces are
    // outstanding: at stage (0,tb0) 8 (A[3] hi is requested after the wait), then 6, 3, 0.  The LDS-DMA requests in between are NOT
    // counted: ordinary loads return in order among themselves, but (measured: wrong results with the DMAs counted) not with respect to
    // LDS-DMA, so the count must hold whichever DMAs are still in flight.  nu = 3 waits for everything: that is also the "patch has
    // landed" wait of the step.
    auto wait_a = [](auto nuc) {
        constexpr int nu = decltype(nuc)::value, younger = NCB * (nu == 0 ? 8 : nu == 1 ? 6 : nu == 2 ? 3 : 0);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(younger) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- B (V) role: lane -> (tile = lane & 31 of the 32-tile block, channels 8 (lane >> 5) + j).  Patch element (channel c, row r, patch
    // column pc) sits at float c * 432 + 64 (c >> 2) + r * 24 + 4 + pc of a buffer.  The wave's xi picks the rows of the row pass:
    //   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]:  t = d[rA] + sigma d[rB],  (rA, rB, sigma) = (0,2,-) (1,2,+) (2,1,-) (1,3,-)
    const int tl = lane & 31, cg = lane >> 5, tx = tl & 7, tyl = tl >> 3;
    const int rA = wv == 0 ? 0 : wv == 2 ? 2 : 1, rB = wv == 2 ? 1 : wv == 3 ? 3 : 2;
    const float sigma = wv == 1 ? 1.0f : -1.0f;
    const unsigned b_lane = smem_lds + (unsigned)(cg * (8 * X_CHF + 128) + tyl * 48 + 2 * tx + 4) * 4u;
    const unsigned rd_a = b_lane + (unsigned)rA * 96u, rd_b = b_lane + (unsigned)rB * 96u;
    u32x4 B[2][2][3];                                 // [slot][position of the pair][plane]: 8 bf16 = channels 8 cg .. 8 cg + 7 of the step
    unsigned long long rw00 = 0, rw01 = 0, rw02 = 0, rw03 = 0, rw10 = 0, rw11 = 0, rw12 = 0, rw13 = 0;   // raw reads of a channel pair: [channel of the pair][row A lo, row A hi, row B lo, row B hi]
    // reads of channel pair q of tile block tb (issued one stage ahead of their use; the wave waits itself).  The immediate must be a
    // literal per instantiation: written out
#define X_RD(e, q, tb, aa, ab)                                                                                                   \
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(rw##e##0) : "v"(aa), "n"(((2 * (q) + (e)) * X_CHF + 64 * ((2 * (q) + (e)) >> 2) + (tb) * 192) * 4));      \
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(rw##e##1) : "v"(aa), "n"(((2 * (q) + (e)) * X_CHF + 64 * ((2 * (q) + (e)) >> 2) + (tb) * 192) * 4 + 8));  \
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(rw##e##2) : "v"(ab), "n"(((2 * (q) + (e)) * X_CHF + 64 * ((2 * (q) + (e)) >> 2) + (tb) * 192) * 4));      \
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(rw##e##3) : "v"(ab), "n"(((2 * (q) + (e)) * X_CHF + 64 * ((2 * (q) + (e)) >> 2) + (tb) * 192) * 4 + 8));
    auto issue_reads = [&rw00, &rw01, &rw02, &rw03, &rw10, &rw11, &rw12, &rw13, rd_a, rd_b](auto qc, auto tbc, unsigned bufoff) {
        constexpr int q = decltype(qc)::value, tb = decltype(tbc)::value;
        const unsigned aa = rd_a + bufoff, ab = rd_b + bufoff;
        X_RD(0, q, tb, aa, ab)
        X_RD(1, q, tb, aa, ab)
    };
    // The wait is followed by an input-only use of all sixteen registers: the reads return asynchronously, and a half the row pass does
    // not consume (column 3 for positions 0-1, column 0 for 2-3) would otherwise be dead to the allocator from the moment the read is
    // ISSUED and handed to another value, which the returning data then overwrites (measured: wrong results).  Input-only, because an
    // in/out tie makes the allocator copy the registers.  The fence keeps the consumers behind the wait.
    auto wait_reads = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(rw00), "v"(rw01), "v"(rw02), "v"(rw03), "v"(rw10), "v"(rw11), "v"(rw12), "v"(rw13) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // channel pair q of tile block tb in two parts.  row_pass consumes the sixteen raw registers (so the next pair's reads can be issued
    // into them right away): t = d[rA] + sigma d[rB] per patch column; finish = column pass, split, pack -> element q of the twelve
    // fragments B[tb][nu][plane] (52 vector instructions, placed beside the stage's matrix instructions).  The empty asm statements pin
    // each result where it is written: the compiler otherwise sinks the whole computation to the fragments' first use, half a step later.
    float tt[2][4];
    auto row_pass = [&](auto npc) {                  // (position pair 0 needs columns 0-2, pair 1 columns 1-3)
        constexpr int np = decltype(npc)::value;
        auto lo = [](unsigned long long u) { return __builtin_bit_cast(float, (unsigned)u); };
        auto hi = [](unsigned long long u) { return __builtin_bit_cast(float, (unsigned)(u >> 32)); };
#ifdef X3_FULL_ROW
        constexpr bool ALLC = true;
#else
        constexpr bool ALLC = false;
#endif
        if (ALLC || np == 0) { tt[0][0] = fmaf(sigma, lo(rw02), lo(rw00)); tt[1][0] = fmaf(sigma, lo(rw12), lo(rw10)); }
        tt[0][1] = fmaf(sigma, hi(rw02), hi(rw00)); tt[1][1] = fmaf(sigma, hi(rw12), hi(rw10));
        tt[0][2] = fmaf(sigma, lo(rw03), lo(rw01)); tt[1][2] = fmaf(sigma, lo(rw13), lo(rw11));
        if (ALLC || np == 1) { tt[0][3] = fmaf(sigma, hi(rw03), hi(rw01)); tt[1][3] = fmaf(sigma, hi(rw13), hi(rw11)); }
        if (ALLC) asm volatile("" : "+v"(tt[0][0]), "+v"(tt[0][3]), "+v"(tt[1][0]), "+v"(tt[1][3]));
        asm volatile("" : "+v"(tt[0][np]), "+v"(tt[0][np + 1]), "+v"(tt[0][np + 2]), "+v"(tt[1][np]), "+v"(tt[1][np + 1]), "+v"(tt[1][np + 2]));
    };
    // position nu (slot `sl`, index i = nu & 1 of its pair) of channel pair q: column pass, split, pack
    auto finish_nu = [&](auto qc, auto slc, auto nuc) {
        constexpr int q = decltype(qc)::value, sl = decltype(slc)::value, nu = decltype(nuc)::value, i = nu & 1;
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e)
            v[e] = nu == 0 ? tt[e][0] - tt[e][2] : nu == 1 ? tt[e][1] + tt[e][2] : nu == 2 ? tt[e][2] - tt[e][1] : tt[e][1] - tt[e][3];
        // round-to-nearest split: v_cvt_pk_bf16_f32 packs the channel pair's bf16 parts [even | odd << 16] in one instruction, the residual
        // x - bf16(x) is exact in f32: x = hi + mid + lo to 2^-27 (truncation: 2^-24, biased), the same 11 instructions per pair
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
        auto pack2 = [](float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_)); };
        unsigned ph = pack2(v[0], v[1]);
        const float r10 = v[0] - __builtin_bit_cast(float, ph << 16), r11 = v[1] - __builtin_bit_cast(float, ph & 0xFFFF0000u);
        unsigned pm = pack2(r10, r11);
        const float r20 = r10 - __builtin_bit_cast(float, pm << 16), r21 = r11 - __builtin_bit_cast(float, pm & 0xFFFF0000u);
        unsigned pl = pack2(r20, r21);
        asm volatile("" : "+v"(ph), "+v"(pm), "+v"(pl));
        B[sl][i][0][q] = ph; B[sl][i][1][q] = pm; B[sl][i][2][q] = pl;
    };

    f32x16 acc[4][NCB][2];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nu][cb][tb][r] = 0.0f;
    // the six products of (nu, tb), smallest terms first; the two channel blocks alternate (independent accumulators).
    // Group g of four = instructions [g 6 NCB / 4, (g + 1) 6 NCB / 4) of the stage; the B fragments are slot sl's, index nu & 1.
    auto mfma_group = [&](auto nuc, auto tbc, auto slc, auto gc) {
        constexpr int nu = decltype(nuc)::value, tb = decltype(tbc)::value, sl = decltype(slc)::value, g = decltype(gc)::value;
        constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int i = g * 6 * NCB / 4; i < (g + 1) * 6 * NCB / 4; ++i) {
            const int t = i / NCB, cb = i % NCB;
#ifdef X3_NO_MFMA
            asm volatile("" :: "v"(A[nu][cb][pa[t]]), "v"(B[sl][nu & 1][pb[t]]));
            continue;
#endif
            acc[nu][cb][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[nu][cb][pa[t]]), __builtin_bit_cast(bf16x8, B[sl][nu & 1][pb[t]]),
                                                                      acc[nu][cb][tb], 0, 0, 0);
        }
    };
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1; typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 3> I3;
    // One stage = the 6 NCB matrix instructions of (nu, tb) from slot sl, in four groups; beside them the wave builds two channel pairs
    // (q0, q0 + 1) of the position pair np of tile block tbp into the OTHER slot (finish_nu after groups 0 / 1 and 2 / 3), reading the
    // patch buffer at byte offset bufp.  The reads of a pair are issued one half stage ahead: after pair q0's row pass those of q0 + 1,
    // after q0 + 1's those of the next stage's first pair (qn, tbn, bufn; skipped when `chain` is false: the next stage sits behind the
    // barrier).  PROD = false: matrix instructions only.  The fences keep the four vector chains apart: the scheduler prices registers
    // against the unified 512-entry file, but vector instructions cannot use the accumulator half, and the allocator then spills.
    auto stage = [&](auto prodc, auto nuc, auto tbc, auto slc, auto npc, auto tbpc, auto q0c, unsigned bufp, auto qnc, auto tbnc, unsigned bufn, bool chain, auto vm) {
        constexpr bool PROD = decltype(prodc)::value;
        constexpr int sl = decltype(slc)::value, np = decltype(npc)::value, q0 = decltype(q0c)::value;
        typedef std::integral_constant<int, 1 - sl> SP; typedef std::integral_constant<int, 2 * np> NA; typedef std::integral_constant<int, 2 * np + 1> NB;
        typedef std::integral_constant<int, q0> Q0; typedef std::integral_constant<int, q0 + 1> Q1;
#ifdef X3_NO_PROD
        constexpr bool PRODV = false;
#else
        constexpr bool PRODV = PROD;
#endif
        if (PRODV) { wait_reads(); row_pass(npc); issue_reads(Q1{}, tbpc, bufp); }
#ifdef X3_WAIT_ALL
        wait_a(nuc);
#else
        if (decltype(tbc)::value == 0) wait_a(nuc);            // (first use of A[nu] in the step; at its second, in tile block 1, it is resident --
                                                                //  and the younger requests counted above are not the ones in flight there)
#endif
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(nuc, tbc, slc, I0{}); vm(I0{}); if (PRODV) finish_nu(Q0{}, SP{}, NA{}); __builtin_amdgcn_sched_barrier(0);
        mfma_group(nuc, tbc, slc, I1{}); vm(I1{}); if (PRODV) finish_nu(Q0{}, SP{}, NB{}); __builtin_amdgcn_sched_barrier(0);
        if (PRODV) { wait_reads(); row_pass(npc); if (chain) issue_reads(qnc, tbnc, bufn); }
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(nuc, tbc, slc, I2{}); vm(I2{}); if (PRODV) finish_nu(Q1{}, SP{}, NA{}); __builtin_amdgcn_sched_barrier(0);
        mfma_group(nuc, tbc, slc, I3{}); vm(I3{}); if (PRODV) finish_nu(Q1{}, SP{}, NB{}); __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::true_type PY;

    // ---- prologue: DMA(0), A(0) except A[3]'s hi plane (the loop's first stage requests it, as in every step), DMA(1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               // (the offsets are in LDS)
#ifdef X3_TIMING
    const unsigned long long Tp1 = __builtin_readcyclecounter();
#endif
    if (PRE) dma_raw(0, 0);
    else {
        typedef std::integral_constant<int, 4> J4; typedef std::integral_constant<int, 5> J5; typedef std::integral_constant<int, 6> J6;
        dma_begin(0, 0);
        dma_chunk(I0{}, all_lanes); dma_chunk(I1{}, all_lanes); dma_chunk(I2{}, all_lanes); dma_chunk(I3{}, all_lanes);
        dma_chunk(J4{}, all_lanes); dma_chunk(J5{}, all_lanes); dma_chunk(J6{}, all_lanes);
    }
    load_a(0, I0{}); load_a(0, I1{}); load_a(0, I2{}); load_piece(0, I3{}, I2{}); load_piece(0, I3{}, I1{});
    if (nsteps > 1) { dma_begin(1, 1); dma_chunk(I0{}, all_lanes); dma_chunk(I1{}, all_lanes); }          // (chunks 2-6 of DMA(1): the loop's first two stages)
    if (PRE) {
        for (int i = tid; i < P.cin; i += 256) {
            const float m = P.pre[((size_t)bz * P.cin + i) * 2], iv = P.pre[((size_t)bz * P.cin + i) * 2 + 1];
            smem[PN_AT + 2 * i] = -m * iv; smem[PN_AT + 2 * i + 1] = iv;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                                 // the table is complete before anybody's fix_raw
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // DMA(0) has landed (no order between DMA and ordinary loads: wait for all)
    }
#ifdef X3_TIMING
    const unsigned long long Tp2 = __builtin_readcyclecounter();
#endif
    fix_raw(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef X3_TIMING
    const unsigned long long Tp3 = __builtin_readcyclecounter();
#endif
    // B[tile block 0][positions 0, 1] of step 0 -> slot 0 (no matrix work to run beside it), then the reads quarter 0 starts with
    issue_reads(I0{}, I0{}, 0u); wait_reads(); row_pass(I0{}); issue_reads(I1{}, I0{}, 0u); finish_nu(I0{}, I0{}, I0{}); finish_nu(I0{}, I0{}, I1{});
    wait_reads(); row_pass(I0{}); issue_reads(I2{}, I0{}, 0u); finish_nu(I1{}, I0{}, I0{}); finish_nu(I1{}, I0{}, I1{});
    wait_reads(); row_pass(I0{}); issue_reads(I3{}, I0{}, 0u); finish_nu(I2{}, I0{}, I0{}); finish_nu(I2{}, I0{}, I1{});
    wait_reads(); row_pass(I0{}); issue_reads(I0{}, I0{}, 0u); finish_nu(I3{}, I0{}, I0{}); finish_nu(I3{}, I0{}, I1{});

    // ---- main loop.  Step s = four quarters of two stages; quarter k multiplies positions (2 (k & 1), + 1) of tile block k >> 1 out of
    // slot k & 1 and builds the next quarter's fragments into the other slot:
    //   k = 0: tile block 0, positions 2, 3     k = 1: tile block 1, positions 0, 1     k = 2: tile block 1, positions 2, 3   (all from raw(s))
    //   BARRIER (DMA(s+1) landed long ago: it is older than A(s)[2]; its patch-up happens here) ; DMA(s+2) -> raw(s)'s buffer
    //   k = 3: tile block 0, positions 0, 1 of step s+1 from raw(s+1)
    // A(s+1)[nu] is requested after A(s)[nu]'s last use (the tile block 1 stage of nu), three stages before its first.
    // One loop body for every step (a peeled last step had register spills, and a spill of an A fragment whose load is still in flight
    // stores garbage: the allocator does not know these loads are asynchronous).  Past the end no request is made (empty EXEC mask, see m1 / m2); the fragments built
    // for a step past the end are computed from whatever the other buffer holds and never used.
#ifdef X3_TIMING
    unsigned long long Tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, Tlast = __builtin_readcyclecounter();
    const unsigned long long Tstart = Tlast;
#endif
    unsigned cur = 0;                                                                 // byte offset of raw(s)'s buffer
    for (int s = 0; s < nsteps; ++s) {
        const unsigned nxt = cur ^ (X_RBUF * 4u);
        const int s1 = s + 1, s2 = s + 2;
        const bool more1 = s1 < nsteps, more2 = s2 < nsteps;
        // No request is made for a step past the end: the request instructions are issued with an empty EXEC mask (no memory access, the
        // destination registers keep their values) instead of being branched around -- a branch would cut a stage's instruction groups
        // into basic blocks, and matrix and vector instructions are interleaved within a block only.
        const unsigned long long m1 = more1 ? all_lanes : 0ull, m2 = more2 ? all_lanes : 0ull;
        XSTAMP(0);
        //    produce    matrix (nu, tb, slot) | builds (pair, tile block, first channel pair, buffer) | next reads | memory requests per group
        // Memory requests, at most one instruction group per matrix-instruction group and wave (the four waves run in lockstep and share
        // the CU's one address unit: bunched, the seven DMA chunks of a step cost each wave ~100 cycles apiece):
        //   tile block 1 stages: the planes of A(s+1)[nu], each in the group after its last use (lo: group 0, mid: 2, hi: 3 of stage (nu, 1))
        //   stages (2,1), (3,1): DMA(s+2) chunks 0, 1 -> raw(s)'s buffer, free after the barrier;   stages (0,0), (1,0) of step s+1: chunks 2-6
        //   (all of DMA(s+2) has been requested two stages before stage (3,0)'s wait for everything)
        auto none = [](auto) {};
        typedef std::integral_constant<int, 4> I4; typedef std::integral_constant<int, 5> I5; typedef std::integral_constant<int, 6> I6;
        dma_begin(more1 ? s1 : s, (int)(nxt != 0));                                    // (chunks 2-6 of DMA(s+1))
        stage(PY{}, I0{}, I0{}, I0{}, I1{}, I0{}, I0{}, cur, I2{}, I0{}, cur, true, [&](auto g) {
            if (decltype(g)::value == 0) load_piece(s, I3{}, I0{});
            if (decltype(g)::value == 1) dma_chunk(I2{}, m1);
            if (decltype(g)::value == 2) dma_chunk(I3{}, m1);
            if (decltype(g)::value == 3) dma_chunk(I4{}, m1); }); XSTAMP(1);
        stage(PY{}, I1{}, I0{}, I0{}, I1{}, I0{}, I2{}, cur, I0{}, I1{}, cur, true, [&](auto g) {
            if (decltype(g)::value == 0) dma_chunk(I5{}, m1);
            if (decltype(g)::value == 1) dma_chunk(I6{}, m1); }); XSTAMP(2);
        stage(PY{}, I2{}, I0{}, I1{}, I0{}, I1{}, I0{}, cur, I2{}, I1{}, cur, true, none); XSTAMP(3);
        stage(PY{}, I3{}, I0{}, I1{}, I0{}, I1{}, I2{}, cur, I0{}, I1{}, cur, true, none); XSTAMP(4);
        stage(PY{}, I0{}, I1{}, I0{}, I1{}, I1{}, I0{}, cur, I2{}, I1{}, cur, true, [&](auto g) {
            if (decltype(g)::value == 1) load_piece_if(m1, s1, I0{}, I2{});
            if (decltype(g)::value == 3) load_piece_if(m1, s1, I0{}, I1{}); }); XSTAMP(5);
        stage(PY{}, I1{}, I1{}, I0{}, I1{}, I1{}, I2{}, cur, I0{}, I0{}, cur, false, [&](auto g) {
            if (decltype(g)::value == 0) load_piece_if(m1, s1, I0{}, I0{});
            if (decltype(g)::value == 1) load_piece_if(m1, s1, I1{}, I2{});
            if (decltype(g)::value == 3) load_piece_if(m1, s1, I1{}, I1{}); }); XSTAMP(6);
        if (more1) fix_raw(s1, (int)(nxt != 0));
        dma_begin(more2 ? s2 : s, (int)(cur != 0));                                    // (chunks 0, 1 of DMA(s+2))
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        XSTAMP(7);
        __builtin_amdgcn_s_barrier();
        // the barrier releases the four waves in the same cycle, and from then on they would present every memory request to the CU's one
        // address unit together (each waiting for the other three: ~50 cycles per request).  Stagger them by X3_STAGGER cycles per wave.
#ifndef X3_STAGGER
#define X3_STAGGER 0            /* (swept 0 / 16 / 32 / 64 on convc2: 505 / 512 / 512 / 520 us -- the request cost is not contention between the waves) */
#endif
        if (X3_STAGGER) {
            if (wv & 1) { for (int i = 0; i < X3_STAGGER / 16; ++i) asm volatile("s_nop 15"); }
            if (wv & 2) { for (int i = 0; i < X3_STAGGER / 8; ++i) asm volatile("s_nop 15"); }
        }
        XSTAMP(8);
        issue_reads(I0{}, I0{}, nxt);
        __builtin_amdgcn_sched_barrier(0);
        XSTAMP(9);
        stage(PY{}, I2{}, I1{}, I1{}, I0{}, I0{}, I0{}, nxt, I2{}, I0{}, nxt, true, [&](auto g) {
            if (decltype(g)::value == 0) load_piece_if(m1, s1, I1{}, I0{});
            if (decltype(g)::value == 1) load_piece_if(m1, s1, I2{}, I2{});
            if (decltype(g)::value == 2) dma_chunk(I0{}, m2);
            if (decltype(g)::value == 3) load_piece_if(m1, s1, I2{}, I1{}); }); XSTAMP(10);
        stage(PY{}, I3{}, I1{}, I1{}, I0{}, I0{}, I2{}, nxt, I0{}, I0{}, nxt, true, [&](auto g) {
            if (decltype(g)::value == 0) load_piece_if(m1, s1, I2{}, I0{});
            if (decltype(g)::value == 1) load_piece_if(m1, s1, I3{}, I2{});
            if (decltype(g)::value == 2) dma_chunk(I1{}, m2);
            if (decltype(g)::value == 3) load_piece_if(m1, s1, I3{}, I1{}); }); XSTAMP(11);
        cur = nxt;
    }
#ifdef X3_TIMING
    const unsigned long long Tloop = __builtin_readcyclecounter();
#endif
    // The requests of the step past the end are in flight and their results dead: keep their registers reserved (input-only uses AFTER
    // the wait) until they have landed -- the allocator hands a dead asm output's register to the epilogue's values at once, and the
    // returning load then overwrites them (measured with the 32-channel tile: wrong accumulators).
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" :: "v"(rw00), "v"(rw01), "v"(rw02), "v"(rw03), "v"(rw10), "v"(rw11), "v"(rw12), "v"(rw13) : "memory");
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) asm volatile("" :: "v"(A[nu][cb][0]), "v"(A[nu][cb][1]), "v"(A[nu][cb][2]));
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                                                     // nobody reads or fills the patch buffers any more

    // ---- epilogue 1: Z[xi][j] = sum_nu M[xi][nu] A^T[j][nu], A^T = [1 1 1 0; 0 1 -1 -1], into LDS [xi][j][co][tile].
    // D layout of the 32x32 block: column (tile) = lane & 31, row (channel) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    // (every wave is past the last barrier, after which nobody reads the patch buffers: no barrier needed before these writes)
    {
        float* zb = &smem[(wv * 2 * 64 + 4 * cg) * 64 + tl];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float m0 = acc[0][cb][tb][r], m1 = acc[1][cb][tb][r], m2 = acc[2][cb][tb][r], m3 = acc[3][cb][tb][r];
                    const int col = 32 * cb + (r & 3) + 8 * (r >> 2);
                    zb[col * 64 + 32 * tb] = (m0 + m1) + m2;
                    zb[(64 + col) * 64 + 32 * tb] = (m1 - m2) - m3;
                }
    }
    // per-channel constants of the final pass (thread -> channel tid >> 5 + 8 it)
    const int m4 = tid & 3, fty = (tid >> 2) & 7, fcol = tid >> 5;
    const int oy = y0 + 2 * fty, ox = x0 + 4 * m4;
    const bool pix_ok = (oy < H) & (ox < W);                                     // (H even, W % 4 == 0: both rows and all four columns, or none)
    float* ob = P.out + (size_t)bz * P.obs;
    float* ob2 = P.out2 ? P.out2 + (size_t)bz * P.o2bs : nullptr;
    const float* rsb = (HAS_AFFINE && P.res) ? P.res + (size_t)bz * P.rbs : nullptr;
    // the per-channel constants of all 4 NCB iterations are requested here, ahead of the barrier (loaded inside the loop each iteration
    // waited a memory round trip: the final pass took 6.6 k cycles of a 64-step workgroup's 121 k)
    float bi_[4 * NCB], sc_[4 * NCB];
#pragma unroll
    for (int it = 0; it < 4 * NCB; ++it) {
        const int co = co0 + fcol + 8 * it, cc = co < P.cout ? co : P.cout - 1;
        bi_[it] = P.bias ? P.bias[cc] : 0.0f;
        sc_[it] = (HAS_AFFINE && P.scale) ? P.scale[cc] : 1.0f;
    }
#ifdef X3_TIMING
    const unsigned long long Tz = __builtin_readcyclecounter();
#endif
    __syncthreads();
#ifdef X3_TIMING
    const unsigned long long Tb = __builtin_readcyclecounter();
#endif
    // ---- epilogue 2: Y[i][j] = sum_xi A^T[i][xi] Z[xi][j] for (channel, tile row, two x-neighbouring tiles): two 16-byte rows
    const unsigned long long grp = __ballot(pix_ok);
    const float nvalid = 8.0f * (float)__popcll(grp & (0xFFull << (lane & 56)));     // pixels of this lane's 16 x 4 region inside the map
    const float inv_nvalid = nvalid > 0.0f ? 1.0f / nvalid : 0.0f;
#pragma unroll
    for (int it = 0; it < 4 * NCB; ++it) {
        const int col = fcol + 8 * it, co = co0 + col;
        const bool cok = co < P.cout;
        const float bi = bi_[it], sc = sc_[it];
        const float* zr = &smem[col * 64 + 8 * fty + 2 * m4];
        float2 z[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int j = 0; j < 2; ++j) z[xi][j] = *(const float2*)(zr + (xi * 2 + j) * 4096);
        // rows i = 0, 1; pixels [tile a: j 0, j 1 | tile b: j 0, j 1]
        f32x4 y0v = {(z[0][0].x + z[1][0].x) + z[2][0].x, (z[0][1].x + z[1][1].x) + z[2][1].x, (z[0][0].y + z[1][0].y) + z[2][0].y, (z[0][1].y + z[1][1].y) + z[2][1].y};
        f32x4 y1v = {(z[1][0].x - z[2][0].x) - z[3][0].x, (z[1][1].x - z[2][1].x) - z[3][1].x, (z[1][0].y - z[2][0].y) - z[3][0].y, (z[1][1].y - z[2][1].y) - z[3][1].y};
        if (HAS_AFFINE && P.scale) { y0v *= sc; y1v *= sc; }
        y0v += bi; y1v += bi;
        if (HAS_STATS && (EPI == 2 || P.stats)) {
            // moments of the 16 x 4 pixel region (8 lanes: tile rows 2g, 2g + 1 x 4 tile pairs) about its first value: the regions and the
            // record order of conv_wino.hip (patch 16 x 8, tile halves), so rpe_instnorm_apply / _finalize read either kernel's records
            const float piv = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane & 56) << 2, __builtin_bit_cast(int, y0v[0])));
            float s1 = 0.0f, s2 = 0.0f;
            if (pix_ok) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d0 = y0v[e] - piv, d1 = y1v[e] - piv; s1 += d0; s2 += d0 * d0; s1 += d1; s2 += d1 * d1; }
            }
            s1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, true));
            s2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0xB1, 0xF, 0xF, true));
            s1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0x4E, 0xF, 0xF, true));
            s2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0x4E, 0xF, 0xF, true));
            s1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0x141, 0xF, 0xF, true));   // row_half_mirror: the other quad of the 8
            s2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0x141, 0xF, 0xF, true));
            const int g = fty >> 1, prow = 2 * py + (g >> 1);                 // conv_wino.hip's patch row (16 x 8 patches) and tile half g & 1
            if ((lane & 7) == 0 && cok && prow * 8 < H) {
                const float mean = s1 * inv_nvalid;
                float* st = P.stats + (((size_t)bz * P.nrec + 2 * (prow * ptx + px) + (g & 1)) * P.cout + co) * 3;
                st[0] = nvalid; st[1] = piv + mean; st[2] = s2 - s1 * mean;
            }
        }
        if (P.mode == RPE_CONV_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { y0v[e] = y0v[e] < 0.0f ? 0.0f : y0v[e]; y1v[e] = y1v[e] < 0.0f ? 0.0f : y1v[e]; }
        }
        if (pix_ok && cok) {
            const size_t e0 = (size_t)co * hw + (size_t)oy * W + ox;
            if (rsb) {
                y0v += *(const f32x4*)(rsb + e0); y1v += *(const f32x4*)(rsb + e0 + W);
#pragma unroll
                for (int e = 0; e < 4; ++e) { y0v[e] = y0v[e] < 0.0f ? 0.0f : y0v[e]; y1v[e] = y1v[e] < 0.0f ? 0.0f : y1v[e]; }
            }
            *(f32x4*)(ob + e0) = y0v; *(f32x4*)(ob + e0 + W) = y1v;
            if (ob2) { *(f32x4*)(ob2 + e0) = y0v; *(f32x4*)(ob2 + e0 + W) = y1v; }
        }
    }
#ifdef X3_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (blockIdx.x == gridDim.x / 2 && blockIdx.z == 0 && blockIdx.y == gridDim.y / 2 && tid == 0) {
        for (int i = 0; i < 12; ++i) g_x3_timing[i] = Tacc[i];
        g_x3_timing[12] = Tstart - Tk0; g_x3_timing[13] = Tloop - Tstart; g_x3_timing[14] = __builtin_readcyclecounter() - Tloop; g_x3_timing[15] = nsteps;
        g_x3_timing[16] = Tz - Tloop; g_x3_timing[17] = Tb - Tz; g_x3_timing[18] = Tp1 - Tk0; g_x3_timing[19] = Tp2 - Tp1; g_x3_timing[20] = Tp3 - Tp2; g_x3_timing[21] = Tstart - Tp3;
    }
#endif
}

// weight (cout, cin, 3, 3) -> U = G g G^T in f32 exactly as k_wino_pack (conv_wino.hip), then the exact three-way bf16 split, laid out
// [step = ci/16][co tile = co/64][xi][nu][cb = (co%64)/32][plane][co%32][ci%16]: a wave's slice of a step is 24 KB contiguous, a fragment
// (32 channels x 16 input channels of one plane) 1 KB in the matrix instruction's operand order
__global__ void k_wino_pack_x3(const float* __restrict__ w, unsigned short* __restrict__ wp, int cout, int cin, int coP, long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;              // over [step][co tile][xi][nu][cb][co32][ci16]
    if (e >= total) return;
    const int ci16 = (int)(e & 15), co32 = (int)((e >> 4) & 31), cb = (int)((e >> 9) & 1), nu = (int)((e >> 10) & 3), xi = (int)((e >> 12) & 3);
    const long long rest = e >> 14;
    const int nct = coP / X_CO;
    const int co = (int)(rest % nct) * X_CO + cb * 32 + co32, ci = (int)(rest / nct) * XK + ci16;
    float v = 0.0f;
    if (co < cout && ci < cin) {
        const float* g = w + ((size_t)co * cin + ci) * 9;
        float col[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g0 = g[0 * 3 + c], g1 = g[1 * 3 + c], g2 = g[2 * 3 + c];
            col[c] = xi == 0 ? g0 : xi == 1 ? 0.5f * ((g0 + g1) + g2) : xi == 2 ? 0.5f * ((g0 - g1) + g2) : g2;
        }
        v = nu == 0 ? col[0] : nu == 1 ? 0.5f * ((col[0] + col[1]) + col[2]) : nu == 2 ? 0.5f * ((col[0] - col[1]) + col[2]) : col[2];
    }
    // round-to-nearest-even bf16 parts: hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid), the residuals exact in f32
    auto bf16_rne = [](float f) { unsigned b = __builtin_bit_cast(unsigned, f); b += 0x7FFFu + ((b >> 16) & 1u); return b & 0xFFFF0000u; };
    const unsigned u = bf16_rne(v);
    const float r1 = v - __builtin_bit_cast(float, u);
    const unsigned u1 = bf16_rne(r1);
    const float r2 = r1 - __builtin_bit_cast(float, u1);
    unsigned short* d = wp + (e >> 9) * (3 * 512) + co32 * 16 + ci16;
    d[0] = (unsigned short)(u >> 16); d[512] = (unsigned short)(u1 >> 16); d[1024] = (unsigned short)(bf16_rne(r2) >> 16);
}

static inline int x3_cop(int cout) { return (cout + X_CO - 1) / X_CO * X_CO; }

extern "C" size_t rpe_conv_wino_x3_packed_bytes(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cin % XK) return 0;
    return (size_t)(cin / XK) * (x3_cop(cout) / X_CO) * 4 * X_U_WAVE;
}

extern "C" int rpe_conv_wino_x3_pack(const float* weight, int cout, int cin, void* packed, void* stream) {
    if (!weight || !packed || cout <= 0 || cin <= 0) return RPE_E_BADARG;
    if (cin % XK) return RPE_E_UNSUPPORTED;
    const long long total = (long long)cin * x3_cop(cout) * 16;
    hipLaunchKernelGGL(k_wino_pack_x3, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, (unsigned short*)packed, cout, cin, x3_cop(cout), total);
    return rpe_check_launch();
}

extern "C" int rpe_conv_wino_x3(const rpe_conv_desc* d, void* stream) {
    if (!d || !d->x || !d->packed || !d->out || d->b <= 0 || d->cin <= 0 || d->cout <= 0 || d->h <= 0 || d->w <= 0) return RPE_E_BADARG;
    if (d->kh != 3 || d->kw != 3 || (d->stride != 0 && d->stride != 1) || (d->cin % XK) || (d->h & 1) || (d->w & 3)) return RPE_E_UNSUPPORTED;
    if (d->mode != RPE_CONV_LINEAR && d->mode != RPE_CONV_RELU) return RPE_E_UNSUPPORTED;
    if (d->add || d->hidden || d->zgate) return RPE_E_UNSUPPORTED;
    if (d->pre_norm && d->cin > 128) return RPE_E_UNSUPPORTED;
    auto a16 = [](const void* p, long long bs) { return !p || ((((uintptr_t)p) & 15) == 0 && (bs & 3) == 0); };
    if (!a16(d->x, d->x_batch_stride) || ((d->h * d->w) & 3) || !a16(d->out, d->out_batch_stride) || !a16(d->out2, d->out2_batch_stride) ||
        !a16(d->residual, d->residual_batch_stride) || (((uintptr_t)d->packed) & 15)) return RPE_E_UNSUPPORTED;
    WinoX3P P;
    P.x = d->x; P.xbs = d->x_batch_stride; P.wp = (const unsigned short*)d->packed; P.cin = d->cin; P.cout = d->cout; P.coP = x3_cop(d->cout);
    P.H = d->h; P.W = d->w; P.bias = d->bias; P.out = d->out; P.obs = d->out_batch_stride; P.out2 = d->out2; P.o2bs = d->out2_batch_stride;
    P.mode = d->mode; P.scale = d->scale; P.res = d->residual; P.rbs = d->residual_batch_stride; P.stats = d->stats; P.pre = d->pre_norm;
    P.nrec = rpe_conv_wino_stats_tiles(d->h, d->w);
    const bool enc = d->scale || d->residual || d->stats || d->pre_norm;
    const int epi = !enc ? 0 : (d->stats && !d->scale && !d->residual) ? 2 : !d->stats ? 1 : 3;
    const int rem = d->cout % X_CO, tail32 = rem > 0 && rem <= 32;
    const int n64 = tail32 ? d->cout / X_CO : P.coP / X_CO;
    const unsigned gx = ceil_div(d->w, 16) * ceil_div(d->h, 16);
    hipStream_t s = (hipStream_t)stream;
    auto launch = [&](auto cbc, dim3 grid) {
        constexpr int CBv = decltype(cbc)::value;
#define X3_LAUNCH(E, PR) hipLaunchKernelGGL((k_conv_wino_x3<E, PR, CBv>), grid, dim3(256), 0, s, P)
        if (d->pre_norm) { if (epi == 2) X3_LAUNCH(2, true); else X3_LAUNCH(3, true); }
        else if (epi == 0) X3_LAUNCH(0, false);
        else if (epi == 1) X3_LAUNCH(1, false);
        else if (epi == 2) X3_LAUNCH(2, false);
        else X3_LAUNCH(3, false);
#undef X3_LAUNCH
    };
    P.co_base = 0;
    if (n64 > 0) launch(std::integral_constant<int, 2>{}, dim3(gx, d->b, n64));
    if (tail32) {
        P.co_base = n64 * X_CO;
        launch(std::integral_constant<int, 1>{}, dim3(gx, d->b, 1));
    }
    return rpe_check_launch();
}
