// 3x3 stride-1 "same" convolutions of RAFT's update block as Winograd F(2x2, 3x3) on the f32 matrix cores.
//
// Replaces (reference's RAFT submodule, call sites core/pose/pose_net.py:47,65,129), twelve times per pass:
//   core/RAFT/core/update.py  BasicMotionEncoder.convc2 (256->192), convf2 (128->64), conv (256->126), FlowHead.conv1 (128->256)
//   each followed by bias + ReLU (+ the torch.cat / copy into the GRU input buffers), fused into the epilogue.
// These four layers are 51 % of the multiply-adds of a GRU iteration.  As direct implicit GEMMs (conv.hip) they are bound
// by the f32 matrix pipe; Winograd trades 2.25x fewer matrix FLOPs for cheap vector adds:
//     Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A        (Lavin & Gray 2016; correlation form, as torch.conv2d)
//   per 2x2 output tile: d = the 4x4 input patch, g = the 3x3 filter; the 16 element-wise products over (ci) are 16
//   independent GEMMs  M_p[co][tile] = sum_ci U_p[co][ci] V_p[ci][tile],  p = 0..15.
// Exact in real arithmetic; in f32 the transforms add a few ulp (|G| <= 1, |B|, |A| in {0, +-1}); the parity tests keep
// their f64-reference tolerances.
//
// Workgroup = 4 waves = 64 output channels x 32 tiles (a patch of 16 x 8 output pixels); wave = 32 channels x 16 tiles =
// 2 blocks of v_mfma_f32_16x16x4_f32 per position, all 16 positions: 128 accumulator registers per lane, so a lane holds
// every position of its (channel, tile) outputs and the output transform needs no cross-lane traffic.  K is walked in
// steps of 4 input channels.  Per step the U slice 4 x 64 x 16 (weights pre-transformed once, L2-resident) and the raw
// 4 x 10 x 18 input patch arrive by LDS-DMA, issued two / three steps ahead (measured: with register-staged loads issued
// one step ahead the matrix pipe waited for L2 most of the time: 60 TFLOP/s executed against 104 with the loads removed);
// two waves turn the raw patch into V 4 x 32 x 16 (B^T d B) for the next step.  The 16 positions of a (channel, output
// channel | tile) row are contiguous, so a lane's fragments for four positions are one 16-B LDS read.
#include "wino_common.h"
#include <type_traits>

#define WB_CO 64
#define WB_TX 8
#define WB_TY 4
#define WB_NT (WB_TX * WB_TY)
#define WK 4
#define PS 20                // floats per (ci, tile) row of V in LDS: 16 positions + 4 pad (rows 80 B apart tile the 64 banks for 16
                             // consecutive rows: the 16-B fragment reads, lane = row, are conflict-free)
#define RAW_W (2 * WB_TX + 2)                                 // input patch of the workgroup: 18 x 10 pixels per channel
#define RAW_H (2 * WB_TY + 2)
#define RAW_CH (RAW_W * RAW_H)
#define RAW_N (WK * RAW_CH)                                   // 720 dwords per step
#define RAW_CHUNKS ((RAW_N + 63) / 64)                        // 12 LDS-DMA instructions of 64 dwords, 3 per wave
#define RAW_BUF (RAW_CHUNKS * 64)
#define U_STEP (WK * WB_CO * 16)                              // 4096 floats per 64-channel slice of a step = 16 DMA instructions of 1 KB

struct WinoP {
    const float* x; long long xbs;
    const float* wp; int cin, cout, coP, H, W;
    const float* bias;
    float* out; long long obs;
    float* out2; long long o2bs;
    int mode;
    // encoder epilogues (ENC instantiation): v = acc * scale + bias; partial instance-norm moments of v; ReLU; residual + ReLU;
    // and the input normalised + ReLU'd while it is transformed (pre: (b, cin, 2) = mean, 1/std of the previous convolution)
    const float* scale; const float* res; long long rbs; float* stats; const float* pre;
    int co_base;                                  // first output channel of this launch (a trailing 32-channel tile is its own launch)
    int v4;                                       // W % 4 == 0 and out / out2 / residual 16-byte aligned: the epilogue moves 16 bytes per lane
};

// three 256-B chunks of gathered dwords: global base + v_j + 256 j  ->  LDS lds_addr + 256 j + lane * 4   (the caller folds the
// -256 j into v_j and keeps it non-negative by biasing the base)
__device__ __forceinline__ void dma4x3(const float* base, unsigned v0, unsigned v1, unsigned v2, unsigned lds_addr) {
    unsigned keep;
    base = wave_uniform(base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, %4\n\tglobal_load_lds_dword %2, %4 offset:256\n\tglobal_load_lds_dword %3, %4 offset:512\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "v"(v1), "v"(v2), "s"(base), "s"(lds_addr) : "memory");
}
#if defined(WINO_TIMING) || defined(WINO_PHASES)
// Experiment hook (tools/build_variant.sh ... -DWINO_TIMING): s_memtime stamps at the group boundaries of the main loop, summed over
// the steps of workgroup (0, 0, 0), wave 0; read back with rpe_debug_wino_timing.
__device__ unsigned long long g_wino_timing[8];
extern "C" int rpe_debug_wino_timing(unsigned long long* out8) { return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_wino_timing), 64) == hipSuccess ? 0 : -1; }
#endif
#ifdef WINO_TIMING
#define STAMP(i) T[i] = __builtin_readcyclecounter()
#else
#define STAMP(i)
#endif
// CB = 16-channel blocks per wave: 2 -> 64 output channels per workgroup, 1 -> 32 (the trailing tile of cout = 96: no padded half)
// EPI = the epilogue's compile-time shape: 0 bias / ReLU / out2 (update block); 1 + scale and residual (cnet: folded batch norm);
// 2 + moments (fnet); 3 all of them at run time.  (With every feature behind a run-time branch the encoder epilogue took 14-22 k
// cycles per workgroup against 9 k for the plain one.)
// RQ: the raw patch arrives as 16-byte quads -- rows [x0 - 4, x0 + 20) of the map, 24 floats apart in LDS, ONE global_load_lds_dwordx4 per
// wave and step instead of three dword gathers, landing 4 bytes into the buffer so that patch column 0 (map column x0 - 1) sits on
// an 8-byte boundary.  The 24-float pitch puts the four tile rows a 32-lane group reads on disjoint banks, and the six 8-byte patch
// reads are single ds_read_b64 (2 LDS cycles each): with 18-float rows the compiler's ds_read2_b64 form (8 cycles, 32 banks, 16-lane
// groups) had tile rows 0 / 1 overlap on 12 banks -- the bank conflicts PMC showed (21 % of the kernel's LDS cycles).  Measured
// (MI355X, batch 32): convc2 586 -> 570 us, encoder layer 1 / 2 1390 -> 1351 / 798 -> 772 us.  Needs W % 4 == 0 and 16-byte aligned
// planes (quads are then wholly inside or wholly outside the map); other shapes keep the dword gather (RQ = false).  (Tried: the
// same reads at 4-byte alignment without the shift -- correct, and 1.7x slower for the whole kernel: misaligned ds_read_b64.)
template <int EPI, bool PRE, int CB, bool RQ>
__global__ __launch_bounds__(256, 2) void k_conv_wino(WinoP P) {
    constexpr bool HAS_AFFINE = EPI == 1 || EPI == 3, HAS_STATS = EPI == 2 || EPI == 3;
    constexpr int TCO = 32 * CB, UT_STEP = WK * TCO * 16;
    constexpr int RROW = RQ ? 24 : RAW_W, RCH = RROW * RAW_H, RBUF = RQ ? 1024 : RAW_BUF, RCOL0 = RQ ? 3 : 0;   // (patch column 0 = map column x0 - 1)
    constexpr int NRAW = RQ ? 1 : 3;                                         // raw-patch DMA instructions per wave and step
    __shared__ __attribute__((aligned(16))) float Us[3][UT_STEP];            // [ci][co][position], as packed in global memory
    __shared__ __attribute__((aligned(16))) float Vs[2][WK][WB_NT][PS];      // [ci][tile][position]
    __shared__ __attribute__((aligned(16))) float Rs[3][RBUF];               // raw input patches [ci][row][col]
    // PRE: (-mean / std, 1 / std) of every input channel (cin <= 128: the encoders' widths).  With the quad patch buffers the workgroup's
    // LDS is exactly half a CU's 160 KB, so the 128 pairs live in the PADDING of V's first buffer (floats 16-17 of its 128 rows of 20,
    // which neither the transform nor the fragment reads touch)
    __shared__ float Pn[PRE && !RQ ? 2 * 128 : 2];
    auto pn_at = [&](int i) -> float* { return PRE && RQ ? &Vs[0][i >> 5][i & 31][16] : &Pn[2 * i]; };
    // (Pn is referenced, hence allocated, only by the pre-norm kernels on the dword-gather path)
    static_assert(sizeof(Us) + sizeof(Vs) + sizeof(Rs) + (PRE && !RQ ? sizeof(Pn) : 0) <= 81920, "two workgroups per CU: at most half of the 160 KB LDS each");
    // every kernel argument the set-up needs is fetched in ONE batch of scalar loads: left to the compiler they were three
    // dependent fetch - wait rounds (6-9 k cycles before the first DMA could be issued, of a 16-step workgroup's 57 k)
    asm volatile("" :: "s"(P.x), "s"(P.wp), "s"(P.out), "s"(P.bias), "s"(P.xbs), "s"(P.obs), "s"(P.cin), "s"(P.cout), "s"(P.coP), "s"(P.H),
                 "s"(P.W), "s"(P.co_base), "s"(P.v4), "s"(P.mode), "s"(P.out2), "s"(P.o2bs));
    if (EPI != 0) asm volatile("" :: "s"(P.scale), "s"(P.res), "s"(P.rbs), "s"(P.stats), "s"(P.pre));
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef WINO_PHASES
    const unsigned long long ph0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();     // (shader cycles | 100 MHz reference ticks)
#endif
#ifdef WINO_PAD
    __shared__ float padlds[WINO_PAD];
    if (P.cin < 0) { padlds[tid] = 1.0f; P.out[0] = padlds[tid ^ 1]; }
#endif
    const int ptx = (P.W + 2 * WB_TX - 1) / (2 * WB_TX);
    // Workgroups are dealt to the 8 XCDs round-robin by linear id (= blockIdx.x mod 8 when gridDim.x is a multiple of 8): each XCD
    // takes a contiguous run of patches, so neighbours -- shared halo rows, the two 64-B halves of an output line -- meet in one L2
#ifndef WINO_NO_XCD
    const int pid = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
#else
    const int pid = blockIdx.x;
#endif
    const int x0 = (pid % ptx) * (2 * WB_TX), y0 = (pid / ptx) * (2 * WB_TY);
    const int co0 = P.co_base + blockIdx.y * TCO, bz = blockIdx.z;
    const int H = P.H, W = P.W, hw = H * W;
    const float* xb = P.x + (size_t)bz * P.xbs;
    const int nsteps = P.cin / WK;

    // ---- DMA roles.  Raw patch: chunk c = 3*wave + j, element e = 64 c + lane -> (ci, row, col) of the 4 x 10 x 18 patch, read
    // from the clamped pixel; in workgroups on the map's border the lane overwrites its out-of-map elements with the padding
    // value once they have landed (patch_raw), so the transform itself carries no masks.  U: chunk c = 4*wave + j, plain copy.
    unsigned roff[3];                                         // byte offsets from (step's first channel - 512 B), chunk offset folded in
    unsigned oob = 0;                                         // bit j: this lane's element of chunk j is outside the map
    if (RQ) {
        // quad q = 64 * wave + lane -> (ci, row, quad column) of the 4 x 10 x 6 quads; lanes past 240 repeat the last quad into the
        // buffer's slack.  Out-of-map quads read a clamped in-map quad and are overwritten once landed (patch_raw)
        const int q0 = wv * 64 + lane, q = q0 < 240 ? q0 : 239;
        const int ci = q / 60, rem = q - ci * 60, r = rem / 6, qc = rem - r * 6;
        int yy = y0 - 1 + r, xx = x0 - 4 + 4 * qc;
        if (q0 < 240 && (yy < 0 || yy >= H || xx < 0 || xx >= W)) oob = 1u;
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy); xx = xx < 0 ? 0 : (xx >= W ? W - 4 : xx);
        roff[0] = (unsigned)(ci * hw + yy * W + xx) * 4u; roff[1] = roff[2] = 0u;
    } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int e0 = (wv * 3 + j) * 64 + lane;
            const int e = e0 < RAW_N ? e0 : RAW_N - 1;
            const int ci = e / RAW_CH, rem = e - ci * RAW_CH, r = rem / RAW_W, c = rem - r * RAW_W;
            int yy = y0 - 1 + r, xx = x0 - 1 + c;
            if (e0 < RAW_N && (yy < 0 || yy >= H || xx < 0 || xx >= W)) oob |= 1u << j;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy); xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            roff[j] = (unsigned)(ci * hw + yy * W + xx) * 4u + 512u - 256u * j;
        }
    }
    // (RQ: the patch's right halo column x0 + 16 sits in a quad of its own, so a patch ending exactly at the map's edge is a border one)
    const bool border = (y0 < 1) | (y0 + 2 * WB_TY >= H) | (x0 < 1) | (x0 + 2 * WB_TX >= W);       // workgroup-uniform
    // padding: zero; with the loader-side normalisation -inf, which relu((x - mean) / std) turns into the zero torch pads with (1/std > 0)
    const float padv = PRE ? -__builtin_inff() : 0.0f;
    auto patch_raw = [&](int buf) {
        if (border) {
            if (RQ) {
                if (oob) {
                    float* q4 = &Rs[buf][4 * (wv * 64 + lane) + 1];
                    q4[0] = padv; q4[1] = padv; q4[2] = padv; q4[3] = padv;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if ((oob >> j) & 1) Rs[buf][(wv * 3 + j) * 64 + lane] = padv;
            }
        }
    };
    // packed weights: [step][64-channel tile][ci][co % 64][16].  CB = 2: the wave's four 1 KB chunks are consecutive; CB = 1: wave =
    // input channel, its 32 rows start at row co0 % 64 of that channel's 64
    const float* wslice = CB == 2 ? P.wp + (size_t)(co0 / WB_CO) * U_STEP + (size_t)(wv * 4) * 256
                                  : P.wp + (size_t)(co0 / WB_CO) * U_STEP + (size_t)wv * (WB_CO * 16) + (size_t)(co0 % WB_CO) * 16;
    const unsigned uoff = lane * 16u;
    const size_t wstep = (size_t)(P.coP / WB_CO) * U_STEP, rstep = (size_t)WK * hw;
    const unsigned us_base = lds_addr_of(&Us[0][0]) + (unsigned)wv * (CB == 2 ? 4096u : 2048u);
    const unsigned rs_base = lds_addr_of(&Rs[0][0]) + (RQ ? (unsigned)wv * 1024u + 4u : (unsigned)(wv * 3) * 256u);
    auto dma_u = [&](const float* src, int buf) {
        if (CB == 2) dma16x4(src, uoff, us_base + (unsigned)buf * (UT_STEP * 4u));
        else dma16x2(src, uoff, us_base + (unsigned)buf * (UT_STEP * 4u));
    };
    auto dma_raw = [&](const float* src, int buf) {
        if (RQ) dma16x1_masked(src, roff[0], rs_base + (unsigned)buf * (RBUF * 4u), wv == 3 ? 0x0000FFFFFFFFFFFFull : ~0ull);
        else dma4x3(src, roff[0], roff[1], roff[2], rs_base + (unsigned)buf * (RBUF * 4u));
    };
    auto clamped = [&](int step) { return step < nsteps ? step : nsteps - 1; };    // (past the end: a harmless repeat keeps the DMA count per step constant)
    const float* xsrc = RQ ? xb : xb - 128;                                         // (the 512-B bias of roff)

    // ---- transform role: thread -> (half, input channel of the step, tile).  V = B^T d B with B^T = [1 0 -1 0; 0 1 1 0;
    // 0 -1 1 0; 0 1 0 -1].  Half 0 produces position rows 0-1: t0 = d0 - d2, t1 = d1 + d2; half 1 rows 2-3: t2 = d2 - d1,
    // t3 = d1 - d3.  With the patch rows a thread reads ordered (A, B, C) = (d0, d2, d1) | (d2, d1, d3) both halves compute
    // A - B and B + sigma C (sigma = +1 | -1, exact), so the column pass has no selects: vector instructions do NOT issue in the
    // shadow of the f32 matrix instructions (measured: ~4 cycles each on top, experiments/mfma_filler_probe.hip), their count matters.
    const int v_half = tid >> 7, v_ci = (tid >> 5) & 3, v_tile = tid & 31, v_tx = v_tile & 7, v_ty = v_tile >> 3;
    const int v_base = v_ci * RCH + 2 * v_ty * RROW + 2 * v_tx + RCOL0;
    const int srcA = v_base + (v_half ? 2 : 0) * RROW, srcB = v_base + (v_half ? 1 : 2) * RROW, srcC = v_base + (v_half ? 3 : 1) * RROW;
    const float sigma = v_half ? -1.0f : 1.0f;
    // The transform of step s+1 is written as three slices (patch reads, column pass, row pass + store) placed by the main loop
    float tdA[4], tdB[4], tdC[4], tta[4], ttb[4];
    float2 pn = make_float2(0.0f, 1.0f);
    // RQ: the six 8-byte patch reads as single ds_read_b64 instructions.  Inline asm because the compiler merges the two reads of a
    // row into ds_read2_b64; the wave waits for them itself (tr_wait) before the column pass.
    unsigned long long rqa0 = 0, rqa1 = 0, rqb0 = 0, rqb1 = 0, rqc0 = 0, rqc1 = 0;
    const unsigned rq_base = lds_addr_of(&Rs[0][0]) + (RQ ? 4u : 0u);               // (the landing shift)
    const unsigned rq_a = rq_base + (unsigned)srcA * 4u, rq_b = rq_base + (unsigned)srcB * 4u, rq_c = rq_base + (unsigned)srcC * 4u;
    auto tr_read = [&](int step, auto rbufc) {
        constexpr int rbuf = decltype(rbufc)::value;
        if (RQ) {
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(rqa0) : "v"(rq_a), "n"(rbuf * RBUF * 4));
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(rqa1) : "v"(rq_a), "n"(rbuf * RBUF * 4 + 8));
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(rqb0) : "v"(rq_b), "n"(rbuf * RBUF * 4));
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(rqb1) : "v"(rq_b), "n"(rbuf * RBUF * 4 + 8));
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(rqc0) : "v"(rq_c), "n"(rbuf * RBUF * 4));
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(rqc1) : "v"(rq_c), "n"(rbuf * RBUF * 4 + 8));
            if (PRE) pn = *(const float2*)pn_at(step * WK + v_ci);
            return;
        }
        const float* rp = &Rs[rbuf][0];
        const float2 a0 = *(const float2*)(rp + srcA), a1 = *(const float2*)(rp + srcA + 2), b0 = *(const float2*)(rp + srcB), b1 = *(const float2*)(rp + srcB + 2);
        const float2 c0 = *(const float2*)(rp + srcC), c1 = *(const float2*)(rp + srcC + 2);
        if (PRE) pn = *(const float2*)pn_at(step * WK + v_ci);                       // (-mean / std, 1 / std)
        tdA[0] = a0.x; tdA[1] = a0.y; tdA[2] = a1.x; tdA[3] = a1.y;
        tdB[0] = b0.x; tdB[1] = b0.y; tdB[2] = b1.x; tdB[3] = b1.y;
        tdC[0] = c0.x; tdC[1] = c0.y; tdC[2] = c1.x; tdC[3] = c1.y;
    };
    auto tr_wait = [&]() {                                    // the asm reads have returned (LDS operations return in order)
        if (RQ) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rqa0), "+v"(rqa1), "+v"(rqb0), "+v"(rqb1), "+v"(rqc0), "+v"(rqc1));
            auto lo = [](unsigned long long v) { return __builtin_bit_cast(float, (unsigned)v); };
            auto hi = [](unsigned long long v) { return __builtin_bit_cast(float, (unsigned)(v >> 32)); };
            tdA[0] = lo(rqa0); tdA[1] = hi(rqa0); tdA[2] = lo(rqa1); tdA[3] = hi(rqa1);
            tdB[0] = lo(rqb0); tdB[1] = hi(rqb0); tdB[2] = lo(rqb1); tdB[3] = hi(rqb1);
            tdC[0] = lo(rqc0); tdC[1] = hi(rqc0); tdC[2] = lo(rqc1); tdC[3] = hi(rqc1);
        }
    };
    auto tr_cols = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float va = tdA[c], vb = tdB[c], vc = tdC[c];
            if (PRE) { va = fmaxf(fmaf(va, pn.y, pn.x), 0.0f); vb = fmaxf(fmaf(vb, pn.y, pn.x), 0.0f); vc = fmaxf(fmaf(vc, pn.y, pn.x), 0.0f); }   // relu((x - mean) / std)
            tta[c] = va - vb;
            ttb[c] = fmaf(sigma, vc, vb);
        }
    };
    auto tr_store = [&](int vbuf) {
        const f32x4 va = {tta[0] - tta[2], tta[1] + tta[2], tta[2] - tta[1], tta[1] - tta[3]};
        const f32x4 vb2 = {ttb[0] - ttb[2], ttb[1] + ttb[2], ttb[2] - ttb[1], ttb[1] - ttb[3]};
        *(f32x4*)&Vs[vbuf][v_ci][v_tile][8 * v_half] = va;
        *(f32x4*)&Vs[vbuf][v_ci][v_tile][8 * v_half + 4] = vb2;
    };

    f32x4 acc[16][CB];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[p][c] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const int cw = wv >> 1, tw = wv & 1, li = lane & 15, lk = lane >> 4;
    // per-channel epilogue constants, requested before the first DMA (ordinary loads return in order with the DMAs: issued here they
    // have landed long before the first counted wait, issued in the epilogue they cost it a memory round trip)
    float bi_[CB][4], sc_[CB][4];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + cw * 16 * CB + cb * 16 + 4 * lk + r;
            const int cc = co < P.cout ? co : P.cout - 1;            // (clamped: no branch around the load; rows past cout are never stored)
            bi_[cb][r] = P.bias ? P.bias[cc] : 0.0f;
            sc_[cb][r] = (HAS_AFFINE && P.scale) ? P.scale[cc] : 1.0f;
        }

    // ---- prologue: everything the first steps need is requested at once (the groups the loop expects in flight, in its order);
    // the wave waits for U(0), raw(0), raw(1) only, V(0) is built, and raw(3) follows once raw(0)'s buffer is free.  Raw
    // s_barrier + explicit waits: __syncthreads() would drain the DMA queue.
#ifdef WINO_PHASES
    const unsigned long long pha = __builtin_readcyclecounter();
#endif
    dma_u(wslice, 0); dma_raw(xsrc, 0); dma_raw(xsrc + (size_t)clamped(1) * rstep, 1);
    dma_u(wslice + (size_t)clamped(1) * wstep, 1); dma_raw(xsrc + (size_t)clamped(2) * rstep, 2);
    dma_u(wslice + (size_t)clamped(2) * wstep, 2);
    if (PRE) {
        // (-mean / std, 1 / std) of the input channels -> LDS, requested AFTER the first DMA groups: in front of them (where this block
        // stood) the workgroup waited one memory round trip for these pairs and then a second one for its first operands -- 15 % of a
        // 16-step workgroup.  They are the youngest requests, so the compiler's wait for them also covers every DMA above (the counted
        // wait below is then a no-op).  (No global loads inside the K loop: hipcc would drain the DMA queue for them.)
        for (int i = tid; i < P.cin; i += 256) {
            const float m = P.pre[((size_t)bz * P.cin + i) * 2], iv = P.pre[((size_t)bz * P.cin + i) * 2 + 1];
            float* d = pn_at(i);
            d[0] = -m * iv; d[1] = iv;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * 2 * CB + NRAW));                                        // vmcnt(U(1) + raw(2) + U(2) still in flight)
#ifdef WINO_PHASES
    const unsigned long long phb = __builtin_readcyclecounter();
#endif
    patch_raw(0); patch_raw(1);
    __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    tr_read(0, std::integral_constant<int, 0>{}); tr_wait(); tr_cols(); tr_store(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                             // V(0) visible; raw(0)'s buffer free
    dma_raw(xsrc + (size_t)clamped(3) * rstep, 0);
    const float* unext = wave_uniform(wslice + (size_t)clamped(3) * wstep);         // U(s + 3), raw(s + 4) of the step the loop is in
    const float* rnext = wave_uniform(xsrc + (size_t)clamped(4) * rstep);
    // ---- step s, four groups of 8 (CB = 2) matrix instructions, one per four positions:
    //   g0  | fragment reads of g1, patch reads of raw(s+1)
    //   g1  | fragment reads of g2 | column pass of the transform
    //   g2  | fragment reads of g3 | row pass, V(s+1) stored | wait: own DMAs older than the newest group landed | border: patch
    //       raw(s+2) | BARRIER
    //   g3  | fragment reads of g0 of step s+1 (U(s+1), V(s+1) are visible now) | DMA U(s+3) -> U(s)'s buffer, raw(s+4) -> raw(s+1)'s
    // so nothing but the barrier itself separates two steps' matrix instructions, and the vector instructions sit in two
    // clusters (an isolated one between two matrix instructions costs ~13 cycles, one more in a cluster ~4).  A DMA group is
    // issued two barriers before its data is read.  The ring positions are compile-time (the loop body is written out for
    // the six (step % 3, step % 2) combinations): every LDS address is a per-thread constant + an immediate.
    // U rows are 64 B with no padding (they arrive by 1 KB DMA chunks), so position group g of row r sits in 16-B slot
    // g ^ ((r >> 2) & 3) (k_wino_pack stores it that way): the 16 rows of a fragment read then cover all 64 banks
    const int sw = (li >> 2) & 3;
    const int uoffl = (lk * TCO + cw * 16 * CB + li) * 16, voffl = (lk * WB_NT + tw * 16 + li) * PS;
    f32x4 fa0 = *(const f32x4*)(&Us[0][uoffl] + 4 * sw), fa1 = *(const f32x4*)(&Us[0][uoffl] + (CB == 2 ? 16 * 16 : 0) + 4 * sw), fb = *(const f32x4*)(&Vs[0][0][0][0] + voffl);
#ifdef WINO_TIMING
    unsigned long long T[7] = {0, 0, 0, 0, 0, 0, 0}, Tp[7] = {0, 0, 0, 0, 0, 0, 0}, Ta[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    auto mfma_group = [&](int g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[4 * g + e][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[e], fb[e], acc[4 * g + e][0], 0, 0, 0);
            if (CB == 2) acc[4 * g + e][CB - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[e], fb[e], acc[4 * g + e][CB - 1], 0, 0, 0);
        }
    };
    auto step = [&](auto ubc, auto curc, const int s) {
        constexpr int UB = decltype(ubc)::value, CUR = decltype(curc)::value, UB1 = (UB + 1) % 3, UB2 = (UB + 2) % 3;
        const float* ua = &Us[UB][uoffl];
        const float* vb = &Vs[CUR][0][0][0] + voffl;
        const int tstep = s + 1 < nsteps ? s + 1 : nsteps - 1;   // (last step: a redundant transform into the buffer nobody reads again)
        f32x4 na0, na1, nb;
        auto frag_reads = [&](const float* u, const float* v, int slot, int vpos) {
            na0 = *(const f32x4*)(u + 4 * (slot ^ sw)); na1 = na0;
            if (CB == 2) na1 = *(const f32x4*)(u + 16 * 16 + 4 * (slot ^ sw));
            nb = *(const f32x4*)(v + vpos);
        };
        STAMP(0);
        // g0
        frag_reads(ua, vb, 1, 4);
        tr_read(tstep, std::integral_constant<int, UB1>{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(0);
        __builtin_amdgcn_sched_barrier(0);
        fa0 = na0; fa1 = na1; fb = nb;
        STAMP(1);
        // g1
        tr_wait();
        __builtin_amdgcn_sched_barrier(0);
        frag_reads(ua, vb, 2, 8);
        __builtin_amdgcn_sched_barrier(0);
        tr_cols();
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(1);
        __builtin_amdgcn_sched_barrier(0);
        fa0 = na0; fa1 = na1; fb = nb;
        STAMP(2);
        // g2
        frag_reads(ua, vb, 3, 12);
        __builtin_amdgcn_sched_barrier(0);
        tr_store(CUR ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(2);
        __builtin_amdgcn_sched_barrier(0);
        fa0 = na0; fa1 = na1; fb = nb;
        STAMP(3);
        // own DMAs except the newest group (2 * CB weight + 3 patch instructions) have landed; the V stores and every fragment
        // read of this step are complete: after the barrier U(s), V(s) and raw(s+1) may be overwritten
        __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * CB + NRAW));                                           // vmcnt(2 CB + 3 | 2 CB + 1)
        patch_raw(UB2);
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                             // lgkmcnt(0)
        STAMP(4);
        __builtin_amdgcn_s_barrier();
        STAMP(5);
#ifdef WINO_TIMING
        if (s > 1) {
#pragma unroll
            for (int i = 0; i < 6; ++i) Ta[i] += Tp[i + 1] - Tp[i];
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        // g3
#ifdef WINO_DMA_EARLY
        dma_u(unext, UB); if (s + 4 < nsteps) unext += wstep;
        dma_raw(rnext, UB1); if (s + 5 < nsteps) rnext += rstep;
        __builtin_amdgcn_sched_barrier(0);
#endif
        frag_reads(&Us[UB1][uoffl], &Vs[CUR ^ 1][0][0][0] + voffl, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[12 + e][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[e], fb[e], acc[12 + e][0], 0, 0, 0);
            if (CB == 2) acc[12 + e][CB - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[e], fb[e], acc[12 + e][CB - 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#ifndef WINO_DMA_EARLY
#ifdef WINO_U_HOT
            if (e == 0) { dma_u(unext, UB); }
#else
            if (e == 0) { dma_u(unext, UB); if (s + 4 < nsteps) unext += wstep; }
#endif
#ifdef WINO_RAW_HOT
            if (e == 1) { dma_raw(rnext, UB1); }
#else
            if (e == 1) { dma_raw(rnext, UB1); if (s + 5 < nsteps) rnext += rstep; }
#endif
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        fa0 = na0; fa1 = na1; fb = nb;
#ifdef WINO_TIMING
        STAMP(6);
#pragma unroll
        for (int i = 0; i < 7; ++i) Tp[i] = T[i];
#endif
    };
#ifdef WINO_PHASES
    const unsigned long long ph1 = __builtin_readcyclecounter();
#endif
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
#ifdef WINO_PHASES
    const unsigned long long ph2 = __builtin_readcyclecounter();
#endif
#ifdef WINO_TIMING
    if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == gridDim.z / 2 && tid == 0) {
        for (int i = 0; i < 6; ++i) g_wino_timing[i] = Ta[i];
        g_wino_timing[6] = nsteps - 2; g_wino_timing[7] = 0;
    }
#endif

#ifdef WINO_NO_EPI
    {   // ablation (tools/build_variant.sh ... -DWINO_NO_EPI): no output transform, no stores -- what the epilogue costs a launch
        float sacc = 0.0f;
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc += acc[p][cb][r];
        if (sacc == 12345.678f) P.out[0] = sacc;
        return;
    }
#endif
    // ---- epilogue.  D layout of the 16x16 MFMA: column (tile) = lane % 16, row (channel) = 4 * (lane / 16) + r.
    // Y = A^T M A,  A^T = [1 1 1 0; 0 1 -1 -1]; then scale / bias, moments, ReLU, residual, and the store(s) into the channel slices.
    const int tl = tw * 16 + li, ty = tl >> 3, tx = tl & 7;
    const int oy = y0 + 2 * ty, ox = x0 + 2 * tx;
    const bool pix_ok = (oy < H) & (ox < W);                        // (H, W even: a tile is inside or outside as a whole)
    float* ob = P.out + (size_t)bz * P.obs;
    float* ob2 = P.out2 ? P.out2 + (size_t)bz * P.o2bs : nullptr;
    const float* rsb = (HAS_AFFINE && P.res) ? P.res + (size_t)bz * P.rbs : nullptr;
    const float nvalid = 4.0f * (float)__popcll(__ballot(pix_ok) & 0xFFFFull);   // pixels of this wave's 16 tiles inside the map
    const float inv_nvalid = nvalid > 0.0f ? 1.0f / nvalid : 0.0f;               // (4, 8, ..., 64: the reciprocal is exact or 1 ulp)
    const bool odd = li & 1;                                        // lane pairs = two x-neighbouring tiles
    // the residual quads of all eight (block, row) iterations are requested up front (the per-channel constants at kernel start):
    // loaded where they are used, each iteration waited a full memory round trip for them behind the previous iteration's stores
    f32x4 rq_[CB][4];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + cw * 16 * CB + cb * 16 + 4 * lk + r;
            const bool cok = co < P.cout;
            rq_[cb][r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            if (HAS_AFFINE && rsb && P.v4 && pix_ok && cok)
                rq_[cb][r] = *(const f32x4*)(rsb + (size_t)co * hw + (size_t)(oy + (odd ? 1 : 0)) * W + (ox - (odd ? 2 : 0)));
        }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        float rec[3] = {0.0f, 0.0f, 0.0f};                             // lane li < 4 collects the moment record of channel row r = li
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = cw * 16 * CB + cb * 16 + 4 * lk + r, co = co0 + col;
            float sa[4], sb[4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                sa[nu] = (acc[0 * 4 + nu][cb][r] + acc[1 * 4 + nu][cb][r]) + acc[2 * 4 + nu][cb][r];
                sb[nu] = (acc[1 * 4 + nu][cb][r] - acc[2 * 4 + nu][cb][r]) - acc[3 * 4 + nu][cb][r];
            }
            const bool cok = co < P.cout;
            const float bi = bi_[cb][r];
            float y[4] = {(sa[0] + sa[1]) + sa[2], (sa[1] - sa[2]) - sa[3], (sb[0] + sb[1]) + sb[2], (sb[1] - sb[2]) - sb[3]};
            if (HAS_AFFINE && P.scale) {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] *= sc_[cb][r]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] += bi;
            if (HAS_STATS && (EPI == 2 || P.stats)) {
                // moments of v over this wave's 16 tiles x 4 pixels about a pivot (the wave's first value of the channel: valid
                // whenever any of its tiles is, because tile 0 is the wave's top-left one), summed over the 16-lane row
                const int y0b = __builtin_bit_cast(int, y[0]);                 // lane 0 of this lane's 16-lane row, without an LDS round trip
                const int q0 = __builtin_amdgcn_readlane(y0b, 0), q1 = __builtin_amdgcn_readlane(y0b, 16);      // (all four, then selects:
                const int q2 = __builtin_amdgcn_readlane(y0b, 32), q3 = __builtin_amdgcn_readlane(y0b, 48);     //  no divergent branches)
                const float piv = __builtin_bit_cast(float, lk == 0 ? q0 : lk == 1 ? q1 : lk == 2 ? q2 : q3);
                float s1 = 0.0f, s2 = 0.0f;
                if (pix_ok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float dv = y[e] - piv; s1 += dv; s2 += dv * dv; }
                }
                s1 = row16_sum(s1); s2 = row16_sum(s2);
                // one (count, mean, M2) record per channel, patch and tile half: rpe_instnorm_apply / _finalize merge records in f64
                // anyway (the pivot is a sample, so s2 - s1^2/n loses at most a factor ~2 in f32).  Every lane of the row has the
                // totals; lane r keeps them, and the four records of a block leave together below (a store per record and value --
                // 24 nearly empty store instructions per wave -- cost the epilogue more than the arithmetic)
                const float m = s1 * inv_nvalid;
                if (li == r) { rec[0] = nvalid; rec[1] = piv + m; rec[2] = s2 - s1 * m; }
            }
            if (P.mode == RPE_CONV_RELU) {                                   // NaN stays NaN, like torch.relu
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = y[e] < 0.0f ? 0.0f : y[e];
            }
            if (P.v4) {
                // the pair exchanges halves (DPP quad_perm [1,0,3,2]): the even lane ends with row 0 of both tiles, the odd lane with
                // row 1: one 16-byte access per lane instead of two 8-byte ones (the store tail is issue-bound)
                const float sA = odd ? y[0] : y[2], sB = odd ? y[1] : y[3];
                const float rA = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sA), 0xB1, 0xF, 0xF, true));
                const float rB = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sB), 0xB1, 0xF, 0xF, true));
                f32x4 q = odd ? (f32x4){rA, rB, y[2], y[3]} : (f32x4){y[0], y[1], rA, rB};
                if (pix_ok && cok) {
                    const size_t e4 = (size_t)co * hw + (size_t)(oy + (odd ? 1 : 0)) * W + (ox - (odd ? 2 : 0));
                    if (rsb) {                                               // ResidualBlock tail: relu(x + y)
                        q += rq_[cb][r];
#pragma unroll
                        for (int e = 0; e < 4; ++e) q[e] = q[e] < 0.0f ? 0.0f : q[e];
                    }
                    *(f32x4*)(ob + e4) = q;
                    if (ob2) *(f32x4*)(ob2 + e4) = q;
                }
            } else if (pix_ok && cok) {
                const size_t e0 = (size_t)co * hw + (size_t)oy * W + ox;
                if (rsb) {
                    const float2 ra = *(const float2*)(rsb + e0), rb2 = *(const float2*)(rsb + e0 + W);
                    y[0] += ra.x; y[1] += ra.y; y[2] += rb2.x; y[3] += rb2.y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = y[e] < 0.0f ? 0.0f : y[e];
                }
                *(float2*)(ob + e0) = make_float2(y[0], y[1]);
                *(float2*)(ob + e0 + W) = make_float2(y[2], y[3]);
                if (ob2) { *(float2*)(ob2 + e0) = make_float2(y[0], y[1]); *(float2*)(ob2 + e0 + W) = make_float2(y[2], y[3]); }
            }
        }
        if (HAS_STATS && (EPI == 2 || P.stats)) {
            const int co = co0 + cw * 16 * CB + cb * 16 + 4 * lk + li;       // lanes 0-3 of each 16-lane row: four consecutive channels
            if (li < 4 && co < P.cout) {
                float* st = P.stats + (((size_t)bz * (2 * gridDim.x) + 2 * pid + tw) * P.cout + co) * 3;      // (b, records, cout, 3)
                st[0] = rec[0]; st[1] = rec[1]; st[2] = rec[2];
            }
        }
    }
#ifdef WINO_PHASES
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == gridDim.z / 2 && tid == 0) {
        const unsigned long long ph3 = __builtin_readcyclecounter();
        g_wino_timing[0] = ph1 - ph0; g_wino_timing[1] = ph2 - ph1; g_wino_timing[2] = ph3 - ph2; g_wino_timing[3] = pha - ph0; g_wino_timing[4] = phb - pha; g_wino_timing[5] = ph1 - phb; g_wino_timing[6] = 1;
        g_wino_timing[7] = __builtin_amdgcn_s_memrealtime() - rt0;      // ticks of the 100 MHz reference over the workgroup's life: the shader clock it ran at
    }
#endif
}

// weight (cout, cin, 3, 3) -> U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], laid out
// [step = ci/4][co tile = co/64][ci%4][co%64][16 positions, 4*xi + nu, in 16-B groups swizzled by the row]: a workgroup's slice
// of a step is 16 KB contiguous (its LDS image)
__global__ void k_wino_pack(const float* __restrict__ w, float* __restrict__ wp, int cout, int cin, int coP, long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int col = (int)((e >> 4) & 63), cil = (int)((e >> 10) & 3);
    const int pos = (int)(((((e >> 2) & 3) ^ ((col >> 2) & 3)) << 2) | (e & 3));     // slot s of row `col` holds position group s ^ ((col >> 2) & 3)
    const long long rest = e >> 12;
    const int ncot = coP / WB_CO;
    const int co = (int)(rest % ncot) * WB_CO + col, ci = (int)(rest / ncot) * WK + cil;
    float v = 0.0f;
    if (co < cout && ci < cin) {
        const float* g = w + ((size_t)co * cin + ci) * 9;
        const int xi = pos >> 2, nu = pos & 3;
        // row xi of G applied to the columns of g, then row nu of G applied to the result
        float col[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g0 = g[0 * 3 + c], g1 = g[1 * 3 + c], g2 = g[2 * 3 + c];
            col[c] = xi == 0 ? g0 : xi == 1 ? 0.5f * ((g0 + g1) + g2) : xi == 2 ? 0.5f * ((g0 - g1) + g2) : g2;
        }
        v = nu == 0 ? col[0] : nu == 1 ? 0.5f * ((col[0] + col[1]) + col[2]) : nu == 2 ? 0.5f * ((col[0] - col[1]) + col[2]) : col[2];
    }
    wp[e] = v;
}

static inline int wino_cop(int cout) { return (cout + WB_CO - 1) / WB_CO * WB_CO; }

extern "C" size_t rpe_conv_wino_packed_floats(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cin % WK) return 0;
    return (size_t)(cin / WK) * 16 * WK * wino_cop(cout);
}

extern "C" int rpe_conv_wino_pack(const float* weight, int cout, int cin, float* packed, void* stream) {
    if (!weight || !packed || cout <= 0 || cin <= 0) return RPE_E_BADARG;
    if (cin % WK) return RPE_E_UNSUPPORTED;
    const long long total = (long long)rpe_conv_wino_packed_floats(cout, cin);
    hipLaunchKernelGGL(k_wino_pack, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, packed, cout, cin, wino_cop(cout), total);
    return rpe_check_launch();
}

extern "C" int rpe_conv_wino_stats_tiles(int h, int w) {
    return (h > 0 && w > 0) ? 2 * ceil_div(w, 2 * WB_TX) * ceil_div(h, 2 * WB_TY) : 0;      // two records (tile halves) per 16x8 patch
}

extern "C" int rpe_conv_wino(const rpe_conv_desc* d, void* stream) {
    if (!d || !d->x || !d->packed || !d->out || d->b <= 0 || d->cin <= 0 || d->cout <= 0 || d->h <= 0 || d->w <= 0) return RPE_E_BADARG;
    if (d->kh != 3 || d->kw != 3 || (d->stride != 0 && d->stride != 1) || (d->cin % WK) || (d->h & 1) || (d->w & 1)) return RPE_E_UNSUPPORTED;
    if (d->mode != RPE_CONV_LINEAR && d->mode != RPE_CONV_RELU) return RPE_E_UNSUPPORTED;
    if (d->add || d->hidden || d->zgate) return RPE_E_UNSUPPORTED;
    const bool enc = d->scale || d->residual || d->stats || d->pre_norm;
    if ((d->pre_norm && d->cin > 128) || (d->residual && ((((uintptr_t)d->residual) & 7) || (d->residual_batch_stride & 1)))) return RPE_E_UNSUPPORTED;
    if ((((uintptr_t)d->packed) & 15) || (((uintptr_t)d->out) & 7) || (d->out_batch_stride & 1) ||
        (d->out2 && ((((uintptr_t)d->out2) & 7) || (d->out2_batch_stride & 1)))) return RPE_E_UNSUPPORTED;
    WinoP P;
    P.x = d->x; P.xbs = d->x_batch_stride; P.wp = d->packed; P.cin = d->cin; P.cout = d->cout; P.coP = wino_cop(d->cout);
    P.H = d->h; P.W = d->w; P.bias = d->bias; P.out = d->out; P.obs = d->out_batch_stride; P.out2 = d->out2; P.o2bs = d->out2_batch_stride;
    P.mode = d->mode;
    auto a16 = [](const void* p, long long bs) { return !p || ((((uintptr_t)p) & 15) == 0 && (bs & 3) == 0); };
    P.v4 = (d->w & 3) == 0 && a16(d->out, d->out_batch_stride) && a16(d->out2, d->out2_batch_stride) && a16(d->residual, d->residual_batch_stride);
    P.scale = d->scale; P.res = d->residual; P.rbs = d->residual_batch_stride; P.stats = d->stats; P.pre = d->pre_norm;
    // 64-channel tiles; a remainder of at most 32 channels (cout = 96) runs as one 32-channel tile instead of a half-empty 64
    const int rem = d->cout % WB_CO, tail32 = rem > 0 && rem <= 32;
    const int n64 = tail32 ? d->cout / WB_CO : P.coP / WB_CO;
    const unsigned gx = ceil_div(d->w, 2 * WB_TX) * ceil_div(d->h, 2 * WB_TY);
    hipStream_t s = (hipStream_t)stream;
    P.co_base = 0;
    // epilogue shape: 0 plain, 1 scale / residual (cnet), 2 moments (fnet), 3 anything else
    const int epi = !enc ? 0 : (d->stats && !d->scale && !d->residual) ? 2 : !d->stats ? 1 : 3;
#ifndef WINO_RAWQ
#define WINO_RAWQ 1
#endif
    // the raw patch as 16-byte quads: rows of whole quads, planes and batch items 16-byte aligned, at least one whole quad per row
    const bool quads = WINO_RAWQ && (d->w & 3) == 0 && d->w >= 4 && ((d->h * d->w) & 3) == 0 && (((uintptr_t)d->x) & 15) == 0 && (d->x_batch_stride & 3) == 0;
    auto launch = [&](auto cbc, dim3 grid) {
        constexpr int CBv = decltype(cbc)::value;
#define WINO_LAUNCH(E, PR, Q) hipLaunchKernelGGL((k_conv_wino<E, PR, CBv, Q>), grid, dim3(256), 0, s, P)
        if (d->pre_norm && quads) { if (epi == 2) WINO_LAUNCH(2, true, true); else WINO_LAUNCH(3, true, true); }
        else if (d->pre_norm) { if (epi == 2) WINO_LAUNCH(2, true, false); else WINO_LAUNCH(3, true, false); }
        else if (quads) {
            if (epi == 0) WINO_LAUNCH(0, false, true);
            else if (epi == 1) WINO_LAUNCH(1, false, true);
            else if (epi == 2) WINO_LAUNCH(2, false, true);
            else WINO_LAUNCH(3, false, true);
        }
        else if (epi == 0) WINO_LAUNCH(0, false, false);
        else if (epi == 1) WINO_LAUNCH(1, false, false);
        else if (epi == 2) WINO_LAUNCH(2, false, false);
        else WINO_LAUNCH(3, false, false);
#undef WINO_LAUNCH
    };
    // Small launches (sequential tracking: batch 1-2) would leave every CU with at most one workgroup = one wave per SIMD, whose
    // K loop is a chain of DMA latencies: 32-channel tiles double the workgroups (two per CU hide each other's stalls).  At full
    // occupancy the 64-channel tile is 15-17 % faster (one weight fragment feeds two matrix instructions), so only below the threshold.
#ifndef WINO_SMALL_WG
#define WINO_SMALL_WG 512LL                       /* (tools/build_variant.sh -DWINO_SMALL_WG=... for A/B runs) */
#endif
    if ((long long)gx * ceil_div(d->cout, WB_CO) * d->b < WINO_SMALL_WG) {
        launch(std::integral_constant<int, 1>{}, dim3(gx, ceil_div(d->cout, 32), d->b));
        return rpe_check_launch();
    }
    if (n64 > 0) launch(std::integral_constant<int, 2>{}, dim3(gx, n64, d->b));
    if (tail32) {
        P.co_base = n64 * WB_CO;
        launch(std::integral_constant<int, 1>{}, dim3(gx, 1, d->b));
    }
    return rpe_check_launch();
}
