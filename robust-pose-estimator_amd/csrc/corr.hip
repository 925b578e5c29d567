// RAFT all-pairs correlation pyramid (build) and radius-4 window lookup for gfx950.
//
// Replaces CorrBlock of the reference's RAFT submodule (core/RAFT/core/corr.py: corr(), __init__ pyramid,
// __call__ lookup; call sites core/pose/pose_net.py:47,65,129).
//
// Pyramid layout (private to this file).  For batch item b, query q1 (row-major over the h8 x w8 grid) and
// level l the (h_l x w_l) correlation map is stored as 4x4 micro-tiles of f32 (64 B, two neighbours in x share
// a 128-B line), micro-tiles row-major, zero-padded to whole tiles:
//     off(y, x) = ((y>>2) * TXC_l + (x>>2)) * 16 + (y&3) * 4 + (x&3),   TXC_l = ceil(w_l/4), S_l = TXC_l*TYC_l*16
//     level l region = [b][q1][S_l] f32, regions of the levels concatenated.
// Why: the lookup reads a 10x10..11x11 window per (query, level).  Row-major maps make that 10 runs of 40 B,
// each touching 1.3-1.6 lines; with micro-tiles every row piece is one aligned 16-B load and a window touches
// ~(13/4)^2 = 10.6 sectors of 64 B instead of ~15.6 -- the kernel is bound by HBM sectors, not instructions.
// Because padded entries are zero, zero padding of out-of-map taps needs checks at tile granularity only.
//
// Kernels:
//   k_permute_fmap2 : fmap2 (b,C,h8,w8) -> B' (b,C,S_0) in micro-tile order (zero padded columns)
//   k_corr_gemm     : C[q1][n'] = sum_c fmap1[c][q1] * B'[c][n'] / sqrt(C)   f32 MFMA 32x32x2 (exact f32
//                     fmaf chain), 128x128 block tile, 4 waves, LDS double buffer.  The GEMM output IS level 0.
//   k_corr_pool     : levels 1..3 by successive 2x2 average pooling (same order of operations as
//                     F.avg_pool2d: ((a+b)+c)+d then /4), one workgroup per query map, staged through LDS.
//   k_corr_lookup   : wave = 64 consecutive queries x one level (coalesced stores of each of the 81 channels);
//                     lane = one query: streams 11 footprint rows with 4 aligned 16-B loads each, aligns them
//                     in registers, and emits the 81 bilinear taps.  Tap positions follow grid_sample's float32
//                     arithmetic per tap (sampling.h), so the integer taps are the reference's.
#include "rpe_common.h"
#include "sampling.h"

#define MAX_LEVELS 4
#define RADIUS 4
#define WIN 9            // 2r+1

struct PyrGeom {
    int b, h8, w8, levels;
    int h[MAX_LEVELS], w[MAX_LEVELS], txc[MAX_LEVELS], tyc[MAX_LEVELS];
    long long S[MAX_LEVELS];        // floats per query map
    long long base[MAX_LEVELS];     // float offset of the level region
    long long total;                // floats
};

static bool make_geom(int b, int h8, int w8, int levels, PyrGeom& G) {
    if (b <= 0 || h8 <= 0 || w8 <= 0 || levels <= 0 || levels > MAX_LEVELS) return false;
    G.b = b; G.h8 = h8; G.w8 = w8; G.levels = levels;
    long long nq = (long long)h8 * w8, off = 0;
    int h = h8, w = w8;
    for (int l = 0; l < levels; ++l) {
        if (h < 2 || w < 2) return false;      // bilinear_sampler divides by (size-1)
        G.h[l] = h; G.w[l] = w; G.txc[l] = (w + 3) / 4; G.tyc[l] = (h + 3) / 4;
        G.S[l] = (long long)G.txc[l] * G.tyc[l] * 16;
        G.base[l] = off;
        off += (long long)b * nq * G.S[l];
        h /= 2; w /= 2;
    }
    for (int l = levels; l < MAX_LEVELS; ++l) { G.h[l] = G.w[l] = G.txc[l] = G.tyc[l] = 0; G.S[l] = 0; G.base[l] = off; }
    G.total = off;
    return true;
}

extern "C" size_t rpe_corr_pyramid_bytes(int b, int h8, int w8, int levels) {
    PyrGeom G;
    if (!make_geom(b, h8, w8, levels, G)) return 0;
    // + one S_0 row of B' per channel is separate scratch appended at the end: (b, C<=256, S_0)
    return (size_t)G.total * 4 + (size_t)b * 256 * G.S[0] * 4 + 256;
}

__device__ __forceinline__ int tile_off(int y, int x, int txc) { return (((y >> 2) * txc + (x >> 2)) << 4) + ((y & 3) << 2) + (x & 3); }

// ------------------------------------------------------------------------------------------------ build
__global__ void k_permute_fmap2(const float* __restrict__ f2, float* __restrict__ Bp, int C, int h8, int w8, int txc, long long S0) {
    // one thread per (c, n'); grid.y = b*C
    long long np = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (np >= S0) return;
    int t = (int)(np >> 4), r = (int)(np & 15);
    int y = (t / txc) * 4 + (r >> 2), x = (t % txc) * 4 + (r & 3);
    float v = 0.0f;
    if (y < h8 && x < w8) v = f2[(size_t)blockIdx.y * h8 * w8 + (size_t)y * w8 + x];
    Bp[(size_t)blockIdx.y * S0 + np] = v;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BN 128
#define BK 16

// A: (b, K, M) row-major (fmap1: K = channels, M = queries);  B: (b, K, N) row-major (B');  C: (b, M, N) row-major.
// EDGE = false: every tile is full and 16-B aligned (M, N multiples of 128): no guards anywhere in the main loop.
template <bool EDGE>
__global__ __launch_bounds__(256) void k_corr_gemm(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                   int M, int N, int K, float scale) {
    __shared__ float As[2][BK][BM];
    __shared__ float Bs[2][BK][BN];
    const int bz = blockIdx.z;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const float* Ab = A + (size_t)bz * K * M;
    const float* Bb = B + (size_t)bz * K * N;
    float* Cb = C + (size_t)bz * M * N;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    // loader mapping: thread -> (k = tid>>5 [+8], 4 consecutive columns at (tid&31)*4)
    const int lk = tid >> 5, lc = (tid & 31) * 4;
    const bool a_full = !EDGE || ((m0 + BM <= M) && (M % 4 == 0)), b_full = !EDGE || ((n0 + BN <= N) && (N % 4 == 0));

    auto load4 = [&](const float* base, int ld, int k, int c0, int limit, bool full) -> float4 {
        const float* p = base + (size_t)k * ld + c0;
        if (!EDGE || full) return *(const float4*)p;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + 0 < limit) v.x = p[0];
        if (c0 + 1 < limit) v.y = p[1];
        if (c0 + 2 < limit) v.z = p[2];
        if (c0 + 3 < limit) v.w = p[3];
        return v;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    constexpr int NL = BK / 8;                  // loader rows per thread (8 k-rows per sweep of the 256 threads)
    float4 ra[NL], rb[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        ra[u] = load4(Ab, M, lk + 8 * u, m0 + lc, M, a_full);
        rb[u] = load4(Bb, N, lk + 8 * u, n0 + lc, N, b_full);
    }
#pragma unroll
    for (int u = 0; u < NL; ++u) { *(float4*)&As[0][lk + 8 * u][lc] = ra[u]; *(float4*)&Bs[0][lk + 8 * u][lc] = rb[u]; }
    __syncthreads();

    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            const int k1 = (kt + 1) * BK;
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                ra[u] = load4(Ab, M, k1 + lk + 8 * u, m0 + lc, M, a_full);
                rb[u] = load4(Bb, N, k1 + lk + 8 * u, n0 + lc, N, b_full);
            }
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int kr = kk + (lane >> 5);
            float a0 = As[cur][kr][wm * 64 + (lane & 31)];
            float a1 = As[cur][kr][wm * 64 + 32 + (lane & 31)];
            float b0 = Bs[cur][kr][wn * 64 + (lane & 31)];
            float b1 = Bs[cur][kr][wn * 64 + 32 + (lane & 31)];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            const int nxt = cur ^ 1;
#pragma unroll
            for (int u = 0; u < NL; ++u) { *(float4*)&As[nxt][lk + 8 * u][lc] = ra[u]; *(float4*)&Bs[nxt][lk + 8 * u][lc] = rb[u]; }
        }
        __syncthreads();
    }
    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (!EDGE || (row < M && col < N)) Cb[(size_t)row * N + col] = acc[i][j][r] * scale;
            }
        }
}

// Levels 1.. from level 0; one workgroup per (query, batch) map.
__global__ __launch_bounds__(256) void k_corr_pool(float* __restrict__ pyr, PyrGeom G) {
    extern __shared__ float lds[];            // two dense maps: level l (src) and level l+1 (dst)
    const long long nq = (long long)G.h8 * G.w8;
    const long long qb = (long long)blockIdx.y * nq + blockIdx.x;       // (b, q1) flattened
    float* src = lds;
    float* dst = lds + (size_t)G.h[0] * G.w[0];
    // stage level 0 densely
    {
        const float* g0 = pyr + G.base[0] + qb * G.S[0];
        const int h = G.h[0], w = G.w[0], txc = G.txc[0];
        const int n4 = (int)(G.S[0] >> 2);                    // 16-B pieces: one micro-tile row each, contiguous in memory
        for (int p = threadIdx.x; p < n4; p += blockDim.x) {
            const float4 v = ((const float4*)g0)[p];
            const int t = p >> 2, y = (t / txc) * 4 + (p & 3), x = (t % txc) * 4;
            if (y < h) {
                float* d = src + (size_t)y * w + x;
                if (x + 3 < w) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
                else { if (x < w) d[0] = v.x; if (x + 1 < w) d[1] = v.y; if (x + 2 < w) d[2] = v.z; }
            }
        }
    }
    __syncthreads();
    for (int l = 1; l < G.levels; ++l) {
        const int hs = G.h[l - 1], ws = G.w[l - 1], hd = G.h[l], wd = G.w[l], txc = G.txc[l];
        float* gl = pyr + G.base[l] + qb * G.S[l];
        const int Sl = (int)G.S[l];
        (void)hs;
        for (int p = threadIdx.x; p < Sl; p += blockDim.x) {           // every padded slot gets a value
            int t = p >> 4, r = p & 15;
            int y = (t / txc) * 4 + (r >> 2), x = (t % txc) * 4 + (r & 3);
            float v = 0.0f;
            if (y < hd && x < wd) {
                const float* s = src + (size_t)(2 * y) * ws + 2 * x;
                v = (((s[0] + s[1]) + s[ws]) + s[ws + 1]) * 0.25f;
                dst[y * wd + x] = v;
            }
            gl[p] = v;
        }
        __syncthreads();
        float* tmp = src; src = dst; dst = tmp;
    }
}

// ------------------------------------------------------------------------------------------------ lookup
// The 9 tap positions of one axis at one level.  Tap i reads pixels lo+i+dev_i and lo+i+dev_i+1 with weights
// (w0, w1); written as three weights over the pixels lo+i, lo+i+1, lo+i+2 so the inner loop has no selects:
//   dev_i = 0 -> (w0, w1, 0)      dev_i = 1 -> (0, w0, w1)      unusable tap -> (0, 0, 0)
struct TapAxis {
    int lo;                // min_i (floor(pos_i) - i)
    unsigned dev;          // bit i: floor(pos_i) - i == lo + 1
    unsigned bad;          // bit i: position not finite / deviation > 1 -> tap contributes zero
    float a0[WIN], a1[WIN], a2[WIN];
};

__device__ __forceinline__ void make_taps(float c, int size, TapAxis& T) {
    int f[WIN];
    float w0[WIN], w1[WIN];
    int lo = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < WIN; ++i) {
        float pos = rt_pos(rn_add(c, (float)(i - RADIUS)), size);     // centroid + delta, then grid_sample
        float pf;
        f[i] = safe_floor(pos, pf) - i;
        w1[i] = pos - pf;                                                // ix - ix_nw
        w0[i] = (pf + 1.0f) - pos;                                       // ix_se - ix
        lo = f[i] < lo ? f[i] : lo;
    }
    T.lo = lo; T.dev = 0; T.bad = 0;
#pragma unroll
    for (int i = 0; i < WIN; ++i) {
        const int e = f[i] - lo;
        const bool bad = e > 1 || lo < -500000;
        const bool dv = e == 1;
        if (dv) T.dev |= 1u << i;
        if (bad) T.bad |= 1u << i;
        T.a0[i] = (bad || dv) ? 0.0f : w0[i];
        T.a1[i] = bad ? 0.0f : (dv ? w0[i] : w1[i]);
        T.a2[i] = (bad || !dv) ? 0.0f : w1[i];
    }
}

// (m & a) | (~m & b) on the bit patterns: one v_bfi_b32
__device__ __forceinline__ float bfi(unsigned m, float a, float b) {
    return __uint_as_float((m & __float_as_uint(a)) | (~m & __float_as_uint(b)));
}
// element e (0..15) of the 16-wide register row held in four float4
#define ROW_E(e) ((e) < 4 ? f0[(e) & 3] : (e) < 8 ? f1[(e) & 3] : (e) < 12 ? f2[(e) & 3] : f3[(e) & 3])
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS staging geometry: per wave, 64 queries x (4 footprint rows x 64 B) with the query stride padded to 272 B
// so that ds_read_b128 by lane = query is bank-conflict free (4*lane mod 64 distinct within each 16-lane group).
#define LK_WAVES 4
#ifndef LK_RPP
#define LK_RPP 4                               // footprint rows staged per pass (4: rows 0-3, 4-7, 8-10;  2: six passes)
#endif
#define LK_QSTRIDE (LK_RPP * 16 + 4)           // floats per query slot: RPP rows x 16 floats + 4 pad (68 / 36: conflict-free b128)
#define LK_WAVE_FLOATS (64 * LK_QSTRIDE)
#define LK_PASSES ((WIN + 2 + LK_RPP - 1) / LK_RPP)
#define LK_NI (4 * LK_RPP)                     // loader instructions per pass: each serves 64 / (4 * RPP) queries
#define LK_QPI (16 / LK_RPP)                   // queries per loader instruction

// One workgroup = 4 waves = 256 consecutive queries of one (batch item, level).
// Loader role (per pass, 16 instructions): instruction i serves queries 4i..4i+3 of the wave; lane L fetches the
//   16-B piece (row (L>>2)&3, micro-tile L&3) of query 4i + (L>>4): 16 lanes cover one query's 4 rows x 64 B, the
//   four pieces of a row sit in four neighbouring micro-tiles = one or two 128-B lines, and every line of the
//   footprint is requested exactly once per wave (the first version re-fetched each line ~3x: 13.9 M line requests
//   per launch at batch 32 against ~5 M distinct lines -- profiles/r01_lookup_pmc.txt).
// Consumer role: lane = query; reads its rows back from LDS, aligns them in registers and emits the 81 taps with
//   coalesced stores (64 consecutive queries per channel).
__global__ __launch_bounds__(64 * LK_WAVES) void k_corr_lookup(const float* __restrict__ pyr, const float* __restrict__ coords,
                                                             float* __restrict__ out, PyrGeom G) {
    __shared__ __attribute__((aligned(16))) float stage[LK_WAVES * LK_WAVE_FLOATS];
    const int l = blockIdx.y, bz = blockIdx.z;
    const int nq = G.h8 * G.w8;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const bool qok = q < nq;
    const int qc = qok ? q : nq - 1;
    const int hl = G.h[l], wl = G.w[l], txc = G.txc[l];
    const float inv = 1.0f / (float)(1 << l);
    const float cx = coords[((size_t)bz * 2 + 0) * nq + qc] * inv;       // coords / 2**i  (exact)
    const float cy = coords[((size_t)bz * 2 + 1) * nq + qc] * inv;
    TapAxis X, Y;
    make_taps(cx, wl, X);
    make_taps(cy, hl, Y);
    const size_t S = (size_t)G.S[l];
    const float* lvl = pyr + G.base[l] + (size_t)bz * nq * S;
    float* o = out + ((size_t)bz * G.levels * WIN * WIN + (size_t)l * WIN * WIN) * nq + qc;

    const int xb = (X.lo >> 2) << 2;          // aligned start column of the 16-wide register row
    const int s = X.lo - xb;                  // 0..3
    const int tx0 = xb >> 2;
    // lane masks for the two register-shift stages; blended with v_bfi (kept as bit ops on purpose: written
    // as selects the optimiser turns the shift into a dynamically indexed private array)
    const unsigned m1 = 0u - (unsigned)(s & 1), m2 = 0u - (unsigned)((s >> 1) & 1);
    // what the loader lanes need to know about a query: first row, first micro-tile column, whether the
    // fourth micro-tile is used (columns s..s+10 reach it only for s >= 2)
    // (packed into one word -> one ds_bpermute per loader instruction; far-outside values are clamped, they only
    // have to stay outside the map)
    const int cl_ylo = Y.lo < -30000 ? -30000 : (Y.lo > 30000 ? 30000 : Y.lo);
    const int cl_tx0 = tx0 < -8000 ? -8000 : (tx0 > 8000 ? 8000 : tx0);
    // the 4th micro-tile is touched only if columns s..s+9 (+1 when some x-tap deviates) reach it; the 11th row
    // only if some y-tap deviates
    const int need4 = (s + 9 + (X.dev ? 1 : 0)) >= 12 ? 1 : 0;
    const int need_r10 = Y.dev ? 1 : 0;
    const int my_packed = ((cl_ylo + 32768) << 16) | ((cl_tx0 + 8192) << 2) | (need_r10 << 1) | need4;

    float* wstage = stage + wv * LK_WAVE_FLOATS;
    const int ld_row = (lane >> 2) & (LK_RPP - 1), ld_piece = lane & 3, ld_sub = lane / (4 * LK_RPP);   // loader role of this lane
    const int q_wave0 = blockIdx.x * blockDim.x + wv * 64;                           // first query of this wave

    float hm2[WIN], hm1[WIN], hc[WIN];        // horizontally interpolated rows r-2, r-1, r
#pragma unroll
    for (int i = 0; i < WIN; ++i) { hm2[i] = 0.0f; hm1[i] = 0.0f; }

#pragma unroll
    for (int pass = 0; pass < LK_PASSES; ++pass) {
        // ---- loader: 16 coalesced 16-B loads per lane-group of 16
        f32x4 v[LK_NI];
        unsigned okbits = 0;                    // which of the 16 pieces are inside the map: applied when they go to LDS (a
                                                // select right after the load makes the compiler branch around every load
                                                // and wait for each shuffle in turn)
        int pks[LK_NI];
#pragma unroll
        for (int i = 0; i < LK_NI; ++i) pks[i] = __shfl(my_packed, LK_QPI * i + ld_sub, 64);   // all shuffles first: one LDS round trip
#pragma unroll
        for (int i = 0; i < LK_NI; ++i) {
            const int src = LK_QPI * i + ld_sub;                                          // query (within the wave) served
            const int pk = pks[i];
            const int ylo_s = (int)((unsigned)pk >> 16) - 32768, tx0_s = ((pk >> 2) & 0x3fff) - 8192, need4_s = pk & 1;
            const int last_row = (pk & 2) ? WIN + 1 : WIN;                           // highest footprint row index needed
            const int qs = q_wave0 + src;
            const int yy = ylo_s + LK_RPP * pass + ld_row;
            const int tx = tx0_s + ld_piece;
            // (bitwise &, not &&: short-circuit evaluation puts a branch in front of every load)
            const bool ok = (qs < nq) & ((LK_RPP * pass + ld_row) <= last_row) & (yy >= 0) & (yy < hl) & (tx >= 0) & (tx < txc) &
                            ((ld_piece < 3) | (need4_s != 0));
            const float* p = lvl + (size_t)(ok ? qs : 0) * S + (ok ? ((((yy >> 2) * txc + tx) << 4) + ((yy & 3) << 2)) : 0);
            v[i] = *(const f32x4*)p;
            okbits |= ok ? (1u << i) : 0u;
        }
        __syncthreads();                                                             // previous pass fully consumed
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < LK_NI; ++i)
            *(f32x4*)(wstage + (LK_QPI * i + ld_sub) * LK_QSTRIDE + ld_row * 16 + ld_piece * 4) = ((okbits >> i) & 1) ? v[i] : zero;
        __syncthreads();
        // ---- consumer: lane = query
        const float* mine = wstage + lane * LK_QSTRIDE;
#pragma unroll
        for (int rr = 0; rr < LK_RPP; ++rr) {
            const int r = LK_RPP * pass + rr;
            if (r >= WIN + 2) break;
            const f32x4 f0 = *(const f32x4*)(mine + rr * 16 + 0), f1 = *(const f32x4*)(mine + rr * 16 + 4);
            const f32x4 f2 = *(const f32x4*)(mine + rr * 16 + 8), f3 = *(const f32x4*)(mine + rr * 16 + 12);
            // align in registers: A[k] = row[k + s], two blend stages (s&1, s&2)
            float Bt[13], A[11];
#pragma unroll
            for (int k = 0; k < 13; ++k) { const float u = ROW_E(k), w = ROW_E(k + 1); Bt[k] = bfi(m1, w, u); }
#pragma unroll
            for (int k = 0; k < 11; ++k) A[k] = bfi(m2, Bt[k + 2], Bt[k]);
            // horizontal interpolation of this row for the 9 x-taps
#pragma unroll
            for (int i = 0; i < WIN; ++i) hc[i] = A[i] * X.a0[i] + A[i + 1] * X.a1[i] + A[i + 2] * X.a2[i];
            // rows (r-2, r-1, r) finish window row j = r-2
            if (r >= 2) {
                const int j = r - 2;
                if (qok) {
#pragma unroll
                    for (int i = 0; i < WIN; ++i)                       // channel i*9+j: x offset i-r, y offset j-r
                        o[(size_t)(i * WIN + j) * nq] = hm2[i] * Y.a0[j] + hm1[i] * Y.a1[j] + hc[i] * Y.a2[j];
                }
            }
#pragma unroll
            for (int i = 0; i < WIN; ++i) { hm2[i] = hm1[i]; hm1[i] = hc[i]; }
        }
    }
}

__global__ void k_corr_taps(const float* __restrict__ coords, int32_t* x0, int32_t* y0, PyrGeom G) {
    const int l = blockIdx.y, bz = blockIdx.z;
    const int nq = G.h8 * G.w8;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const float inv = 1.0f / (float)(1 << l);
    const float cx = coords[((size_t)bz * 2 + 0) * nq + q] * inv;
    const float cy = coords[((size_t)bz * 2 + 1) * nq + q] * inv;
    TapAxis X, Y;
    make_taps(cx, G.w[l], X);
    make_taps(cy, G.h[l], Y);
    // floor index the lookup kernel uses for tap i of each axis: lo + i + dev_i  (-1000000: unusable tap)
    for (int i = 0; i < WIN; ++i) {
        size_t o = (((size_t)bz * G.levels + l) * WIN + i) * nq + q;
        x0[o] = ((X.bad >> i) & 1u) ? -1000000 : X.lo + i + (int)((X.dev >> i) & 1u);
        y0[o] = ((Y.bad >> i) & 1u) ? -1000000 : Y.lo + i + (int)((Y.dev >> i) & 1u);
    }
}

__global__ void k_corr_export(const float* __restrict__ pyr, float* __restrict__ dense, PyrGeom G, int l) {
    const long long nq = (long long)G.h8 * G.w8;
    const long long qb = blockIdx.x;                          // (b, q1) flattened
    const int h = G.h[l], w = G.w[l];
    const float* g = pyr + G.base[l] + qb * G.S[l];
    for (int p = threadIdx.x; p < h * w; p += blockDim.x) {
        int y = p / w, x = p - y * w;
        dense[qb * h * w + p] = g[tile_off(y, x, G.txc[l])];
    }
    (void)nq;
}

extern "C" int rpe_corr_build(const float* fmap1, const float* fmap2, int b, int c, int h8, int w8, int levels,
                              void* pyramid, void* stream) {
    PyrGeom G;
    if (!fmap1 || !fmap2 || !pyramid || c <= 0 || c > 256 || c % BK != 0 || !make_geom(b, h8, w8, levels, G)) return RPE_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    float* pyr = (float*)pyramid;
    float* Bp = pyr + G.total;                               // scratch behind the pyramid (b, c, S_0)
    const long long S0 = G.S[0];
    const int nq = h8 * w8;
    hipLaunchKernelGGL(k_permute_fmap2, dim3(ceil_div(S0, 256), b * c), dim3(256), 0, s, fmap2, Bp, c, h8, w8, G.txc[0], S0);
    const bool edge = (nq % BM) || (S0 % BN) || ((uintptr_t)fmap1 % 16) || ((uintptr_t)pyramid % 16);
    if (edge) hipLaunchKernelGGL(k_corr_gemm<true>, dim3(ceil_div(S0, BN), ceil_div(nq, BM), b), dim3(256), 0, s, fmap1, (const float*)Bp,
                                 pyr + G.base[0], nq, (int)S0, c, 1.0f / sqrtf((float)c));
    else hipLaunchKernelGGL(k_corr_gemm<false>, dim3(ceil_div(S0, BN), ceil_div(nq, BM), b), dim3(256), 0, s, fmap1, (const float*)Bp,
                            pyr + G.base[0], nq, (int)S0, c, 1.0f / sqrtf((float)c));
    if (levels > 1) {
        size_t lds = ((size_t)G.h[0] * G.w[0] + (size_t)G.h[1] * G.w[1]) * sizeof(float);
        if (lds > 160 * 1024) return RPE_E_UNSUPPORTED;
        hipLaunchKernelGGL(k_corr_pool, dim3(nq, b), dim3(256), lds, s, pyr, G);
    }
    return rpe_check_launch();
}

extern "C" int rpe_corr_lookup(const void* pyramid, const float* coords, int b, int h8, int w8, int levels, int radius,
                               float* out, void* stream) {
    PyrGeom G;
    if (!pyramid || !coords || !out || radius != RADIUS || !make_geom(b, h8, w8, levels, G)) return RPE_E_BADARG;
    const int nq = h8 * w8;
    hipLaunchKernelGGL(k_corr_lookup, dim3(ceil_div(nq, 256), levels, b), dim3(256), 0, (hipStream_t)stream,
                       (const float*)pyramid, coords, out, G);
    return rpe_check_launch();
}

extern "C" int rpe_corr_lookup_taps(const float* coords, int b, int h8, int w8, int levels, int32_t* x0, int32_t* y0,
                                    void* stream) {
    PyrGeom G;
    if (!coords || !x0 || !y0 || !make_geom(b, h8, w8, levels, G)) return RPE_E_BADARG;
    const int nq = h8 * w8;
    hipLaunchKernelGGL(k_corr_taps, dim3(ceil_div(nq, 256), levels, b), dim3(256), 0, (hipStream_t)stream, coords, x0, y0, G);
    return rpe_check_launch();
}

extern "C" int rpe_corr_export_level(const void* pyramid, int b, int h8, int w8, int levels, int level, float* dense,
                                     void* stream) {
    PyrGeom G;
    if (!pyramid || !dense || level < 0 || level >= levels || !make_geom(b, h8, w8, levels, G)) return RPE_E_BADARG;
    hipLaunchKernelGGL(k_corr_export, dim3(b * h8 * w8), dim3(256), 0, (hipStream_t)stream, (const float*)pyramid, dense, G, level);
    return rpe_check_launch();
}
