// RAFT all-pairs correlation pyramid (build) and radius-4 window lookup for gfx950.
//
// Replaces CorrBlock of the reference's RAFT submodule (core/RAFT/core/corr.py: corr(), __init__ pyramid,
// __call__ lookup; call sites core/pose/pose_net.py:47,65,129).
//
// Pyramid layout (private to this file): GROUP-INTERLEAVED, SKEWED ROWS.
//   A *group* = 8 queries that are neighbours in x: g = (q.y, q.x >> 3), k = q.x & 7.  For batch item b, group g
//   and level l the eight (h_l x w_l) correlation maps of the group are stored together, query index fastest:
//       slot(y, x', k),  x' = x + sk_l - (k >> l),  sk_l = 7 >> l          ("skew": query k's map is shifted left
//       addr = base_l + ((b*ngroups + g) * h_l + y) * wp_l*8 + x'*8 + k      by the distance its window centre moves)
//   with wp_l = (w_l + sk_l) rounded up to 4; slots whose x falls outside [0, w_l) hold zeros (they ARE the zero
//   padding of grid_sample).  One 128-B line = 4 x' x 8 queries of one map row.
// Why: a query reads a 10x10 (11x11) window of ITS OWN map per level, so nothing is shared between queries -- unless
// the maps of neighbouring queries are interleaved.  Neighbouring queries have neighbouring window centres
// (centre = q + flow, and flow is smooth), so after the skew the 8 windows of a group coincide: the group needs
// ~10 rows x (10 + 3)/4 lines and every fetched line is used by all 8 queries.  Per query and level that is ~4 lines
// instead of ~7 with per-query 8x4-pixel lines (and ~15 row-major): the lookup is bound by HBM lines.  Rows are whole
// lines, so a loader pass over any subset of rows requests every line exactly once.  Flow that is NOT smooth inside a
// group costs bandwidth, never correctness: a window that leaves the group's 12 x 16 staging box takes a per-tap
// path straight from global memory.
//
// Kernels:
//   k_permute_fmap  : fmap (b,C,h8,w8) -> (b,C,N') in GEMM tile order, zero padded: fmap1 in group order (padded to
//                     128 queries), fmap2 in 8x16-pixel patch order
//   k_corr_build    : C[q][p] = sum_c fmap1[c][q] * fmap2[c][p] / sqrt(C), f32 MFMA 32x32x2 (exact f32 fmaf chain),
//                     128 queries x 128 pixels (one 8x16 patch) per tile, 4 waves, LDS double buffer; a workgroup walks the
//                     patches of one 8-row band, and the epilogue pools each tile to levels 1-3 (2x2 means in
//                     F.avg_pool2d's order of operations; an 8x16 patch holds whole 8x8 blocks) and scatters all four
//                     levels into the skewed layout -- no separate pooling pass, level 0 is never re-read
//   k_corr_lookup   : wave = 8 groups x one level.  Loader role: 16-B pieces, 8 lanes per 128-B line, each needed line
//                     requested once, staged in wave-private LDS (no workgroup barrier).  Consumer role: lane = query,
//                     4-B LDS reads at its own column offset, 9 horizontal taps per row, 81 outputs with coalesced stores.
//                     Tap positions follow grid_sample's float32 arithmetic per tap (sampling.h): integer taps are the
//                     reference's bit for bit.
#include "rpe_common.h"
#include "sampling.h"

#define MAX_LEVELS 4
#define RADIUS 4
#define WIN 9            // 2r+1
#define GQ 8             // queries per group

struct PyrGeom {
    int b, h8, w8, levels;
    int gx, ngroups;                 // groups per query row, groups per batch item
    int mp;                          // queries in group order, padded to the GEMM tile (128)
    int nbands, npx, np;             // 8-row bands, 16-column patches per band, padded pixel count (nbands*npx*128)
    int h[MAX_LEVELS], w[MAX_LEVELS], sk[MAX_LEVELS], wp[MAX_LEVELS];
    long long base[MAX_LEVELS];      // float offset of the level region [b][group][y][x'][k]
    long long total;                 // floats
};

static bool make_geom(int b, int h8, int w8, int levels, PyrGeom& G) {
    if (b <= 0 || h8 <= 0 || w8 <= 0 || levels <= 0 || levels > MAX_LEVELS) return false;
    G.b = b; G.h8 = h8; G.w8 = w8; G.levels = levels;
    G.gx = (w8 + GQ - 1) / GQ; G.ngroups = G.gx * h8; G.mp = (G.ngroups * GQ + 127) / 128 * 128;
    G.nbands = (h8 + 7) / 8; G.npx = (w8 + 15) / 16; G.np = G.nbands * G.npx * 128;
    long long off = 0;
    int h = h8, w = w8;
    for (int l = 0; l < MAX_LEVELS; ++l) {
        if (l < levels) {
            if (h < 2 || w < 2) return false;      // bilinear_sampler divides by (size-1)
            G.h[l] = h; G.w[l] = w; G.sk[l] = 7 >> l; G.wp[l] = (w + G.sk[l] + 3) / 4 * 4;
            G.base[l] = off;
            off += (long long)b * G.ngroups * h * G.wp[l] * GQ;
            h /= 2; w /= 2;
        } else { G.h[l] = G.w[l] = G.sk[l] = G.wp[l] = 0; G.base[l] = off; }
    }
    G.total = off;
    return true;
}

static size_t scratch_floats(const PyrGeom& G, int c) { return (size_t)G.b * c * ((size_t)G.mp + G.np); }

extern "C" size_t rpe_corr_pyramid_bytes(int b, int h8, int w8, int levels) {
    PyrGeom G;
    if (!make_geom(b, h8, w8, levels, G)) return 0;
    // the permuted feature maps (GEMM operands, <= 256 channels) are scratch behind the pyramid
    return ((size_t)G.total + scratch_floats(G, 256)) * 4 + 256;
}

// ------------------------------------------------------------------------------------------------ build
// mode 0: group order  n' = (y*gx + x/8)*8 + x%8      (fmap1; padded to mp)
// mode 1: patch order  n' = ((y/8)*npx + x/16)*128 + (y%8)*16 + x%16   (fmap2; padded to np)
__global__ void k_permute_fmap(const float* __restrict__ f, float* __restrict__ out, int h8, int w8, int mode, int gx, int npx, int npad) {
    const int np = blockIdx.x * blockDim.x + threadIdx.x;         // grid.y = b*C
    if (np >= npad) return;
    int y, x;
    if (mode == 0) { const int g = np >> 3; y = g / gx; x = (g % gx) * 8 + (np & 7); }
    else { const int t = np >> 7, r = np & 127; y = (t / npx) * 8 + (r >> 4); x = (t % npx) * 16 + (r & 15); }
    float v = 0.0f;
    if (y < h8 && x < w8) v = f[(size_t)blockIdx.y * h8 * w8 + (size_t)y * w8 + x];
    out[(size_t)blockIdx.y * npad + np] = v;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BM 128
#define BN 128
#define BK 16
#define TP 136                      // row pitch of the staged half tile (floats)

// Scatter one level of a half tile (64 queries = 8 groups) into the skewed layout.
//   src[q * pitch + yy * PW + xx]: the level-l values of this patch (R rows x PW columns per query)
// Destination slots of the patch: x' in [x0, x0 + PW + SK) -- slot (x', k) holds x = x' - SK + (k >> L); it is written
// with data when x lies in this patch, with zero when x is outside the map (the left edge by the first patch, the right
// edge and the row padding up to wp by the last), and left to the neighbouring patch otherwise.
template <int L>
__device__ __forceinline__ void scatter_level(const float* __restrict__ src, int pitch, float* __restrict__ lvl, const PyrGeom& G, int bz,
                                              int g0, int band, int px, int tid) {
    constexpr int PW = 16 >> L, R = 8 >> L, SK = 7 >> L, S = PW + SK;
    const int hl = G.h[L], wl = G.w[L], wp = G.wp[L];
    const int x0 = px * PW, y0 = band * R;
    const size_t rowf = (size_t)wp * 8;
    for (int idx = tid; idx < 8 * R * S * 8; idx += 256) {       // (divisions by compile-time constants)
        const int k = idx & 7, t = idx >> 3;
        const int s = t % S, t2 = t / S;
        const int yy = t2 % R, g = t2 / R;
        const int y = y0 + yy, xs = x0 + s;                       // destination row and x'
        const int Gi = g0 + g;
        if (Gi >= G.ngroups || y >= hl || xs >= wp) continue;
        const int x = xs - SK + (k >> L), xx = x - x0;
        float v;
        if (xx >= 0 && xx < PW && x < wl) v = src[(g * 8 + k) * pitch + yy * PW + xx];
        else if (x < 0 || x >= wl) v = 0.0f;
        else continue;                                            // inside the map but another patch's column
        lvl[((size_t)bz * G.ngroups + Gi) * hl * rowf + (size_t)y * rowf + (size_t)xs * 8 + k] = v;
    }
    if (px == G.npx - 1) {                                       // row padding right of the last patch's slots: zeros
        const int xe = x0 + S, ne = wp - xe;                      // x' in [xe, wp): x >= x0 + PW >= w_l for every k
        if (ne > 0) {
            for (int idx = tid; idx < 8 * R * ne * 8; idx += 256) {
                const int k = idx & 7, t = idx >> 3;
                const int s = t % ne, t2 = t / ne;
                const int yy = t2 % R, g = t2 / R;
                const int y = y0 + yy, Gi = g0 + g;
                if (Gi >= G.ngroups || y >= hl) continue;
                lvl[((size_t)bz * G.ngroups + Gi) * hl * rowf + (size_t)y * rowf + (size_t)(xe + s) * 8 + k] = 0.0f;
            }
        }
    }
}

// A: (b, K, mp) fmap1 in group order;  B: (b, K, np) fmap2 in patch order.  grid = (nbands, mp/128, b).
__global__ __launch_bounds__(256, 2) void k_corr_build(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ pyr,
                                                    int K, float scale, PyrGeom G) {
    // main loop: As[2][BK][BM] | Bs[2][BK][BN] (32 KB); epilogue (aliased): T[64][TP] | T1[64][33] | T2[64][9] | T3[64][3]
    __shared__ __attribute__((aligned(16))) float smem[64 * TP + 64 * 33 + 64 * 9 + 64 * 3];
    float (*As)[BK][BM] = (float (*)[BK][BM])smem;
    float (*Bs)[BK][BN] = (float (*)[BK][BN])(smem + 2 * BK * BM);
    float* T = smem;
    float* T1 = smem + 64 * TP;
    float* T2 = T1 + 64 * 33;
    float* T3 = T2 + 64 * 9;
    const int bz = blockIdx.z, band = blockIdx.x;
    const int m0 = blockIdx.y * BM;
    const int M = G.mp, N = G.np;
    const float* Ab = A + (size_t)bz * K * M;
    const float* Bb = B + (size_t)bz * K * N;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int lk = tid >> 5, lc = (tid & 31) * 4;                 // loader: (k = tid>>5 [+8], 4 consecutive columns)
    static_assert(BK == 16, "two loader rows per thread");
    const int nk = K / BK;

    for (int px = 0; px < G.npx; ++px) {
        const int n0 = (band * G.npx + px) * BN;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        float4 ra0, ra1, rb0, rb1;                                // (scalars: as arrays they are demoted to LDS)
        ra0 = *(const float4*)(Ab + (size_t)lk * M + m0 + lc);        ra1 = *(const float4*)(Ab + (size_t)(lk + 8) * M + m0 + lc);
        rb0 = *(const float4*)(Bb + (size_t)lk * N + n0 + lc);        rb1 = *(const float4*)(Bb + (size_t)(lk + 8) * N + n0 + lc);
        __syncthreads();                                          // the previous patch's epilogue is done with smem
        *(float4*)&As[0][lk][lc] = ra0; *(float4*)&As[0][lk + 8][lc] = ra1;
        *(float4*)&Bs[0][lk][lc] = rb0; *(float4*)&Bs[0][lk + 8][lc] = rb1;
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) {
                const int k1 = (kt + 1) * BK + lk;
                ra0 = *(const float4*)(Ab + (size_t)k1 * M + m0 + lc);    ra1 = *(const float4*)(Ab + (size_t)(k1 + 8) * M + m0 + lc);
                rb0 = *(const float4*)(Bb + (size_t)k1 * N + n0 + lc);    rb1 = *(const float4*)(Bb + (size_t)(k1 + 8) * N + n0 + lc);
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
                *(float4*)&As[nxt][lk][lc] = ra0; *(float4*)&As[nxt][lk + 8][lc] = ra1;
                *(float4*)&Bs[nxt][lk][lc] = rb0; *(float4*)&Bs[nxt][lk + 8][lc] = rb1;
            }
            __syncthreads();
        }
        // ---- epilogue, one half (64 queries = the rows of the waves with wm == hh) at a time.
        // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        for (int hh = 0; hh < 2; ++hh) {
            if (hh) __syncthreads();                              // half 0's scatter has read T..T3
            if (wm == hh) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int col = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                            T[row * TP + col] = acc[i][j][r] * scale;
                        }
                    }
            }
            __syncthreads();
            // 2x2 means in F.avg_pool2d's order: ((a + b) + c) + d, then * 0.25
            for (int idx = tid; idx < 64 * 32; idx += 256) {      // level 1: 4 x 8 cells per query
                const int q = idx >> 5, c1 = idx & 31;
                const float* s = T + q * TP + (2 * (c1 >> 3)) * 16 + 2 * (c1 & 7);
                T1[q * 33 + c1] = (((s[0] + s[1]) + s[16]) + s[17]) * 0.25f;
            }
            __syncthreads();
            for (int idx = tid; idx < 64 * 8; idx += 256) {       // level 2: 2 x 4
                const int q = idx >> 3, c2 = idx & 7;
                const float* s = T1 + q * 33 + (2 * (c2 >> 2)) * 8 + 2 * (c2 & 3);
                T2[q * 9 + c2] = (((s[0] + s[1]) + s[8]) + s[9]) * 0.25f;
            }
            __syncthreads();
            if (tid < 64 * 2) {                                   // level 3: 1 x 2
                const int q = tid >> 1, c3 = tid & 1;
                const float* s = T2 + q * 9 + 2 * c3;
                T3[q * 3 + c3] = (((s[0] + s[1]) + s[4]) + s[5]) * 0.25f;
            }
            __syncthreads();
            const int g0 = (m0 + hh * 64) >> 3;
            scatter_level<0>(T, TP, pyr + G.base[0], G, bz, g0, band, px, tid);
            if (G.levels > 1) scatter_level<1>(T1, 33, pyr + G.base[1], G, bz, g0, band, px, tid);
            if (G.levels > 2) scatter_level<2>(T2, 9, pyr + G.base[2], G, bz, g0, band, px, tid);
            if (G.levels > 3) scatter_level<3>(T3, 3, pyr + G.base[3], G, bz, g0, band, px, tid);
        }
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

#define LK_WAVES 4
#ifndef LK_RPP
#define LK_RPP 4                               // box rows staged per pass
#endif
#define LK_BOXW 16                             // x' slots per staged row (4 lines)
#define LK_BOXH 12                             // box rows (LK_BOXH / LK_RPP passes)
#define LK_GSTRIDE (LK_RPP * LK_BOXW * GQ + 8) // floats per group: +8 rotates the banks group to group (conflict-free 4-B reads
                                               // when the groups of a wave sit at the same column offset -- the usual case)
#define LK_WAVE_FLOATS (8 * LK_GSTRIDE)
#define LK_NI (4 * LK_RPP)                     // loader instructions per pass: 8 groups x RPP rows x 4 lines x 8 pieces / 64 lanes

__device__ __forceinline__ int group_min(int v) {
    v = min(v, __shfl_xor(v, 1, 64)); v = min(v, __shfl_xor(v, 2, 64)); v = min(v, __shfl_xor(v, 4, 64));
    return v;
}
__device__ __forceinline__ int group_max(int v) {
    v = max(v, __shfl_xor(v, 1, 64)); v = max(v, __shfl_xor(v, 2, 64)); v = max(v, __shfl_xor(v, 4, 64));
    return v;
}

// Consumer of one staged pass: box rows LK_RPP*pass .. +LK_RPP-1.  DEV = false: no tap of any lane of the wave deviates
// (the usual case away from exactly-integer coordinates): 10 columns, two weights per tap on both axes.
template <bool DEV>
__device__ __forceinline__ void consume_pass(const float* __restrict__ mine, int pass, const TapAxis& X, const TapAxis& Y, int yoff, bool store_ok,
                                             float* __restrict__ o, int nq, float (&hm2)[WIN], float (&hm1)[WIN]) {
#pragma unroll
    for (int rr = 0; rr < LK_RPP; ++rr) {
        const int rb = LK_RPP * pass + rr;                               // box row (compile time)
        float A[WIN + 2], hc[WIN];
#pragma unroll
        for (int c = 0; c < (DEV ? WIN + 2 : WIN + 1); ++c) A[c] = mine[rr * (LK_BOXW * GQ) + c * GQ];
#pragma unroll
        for (int i = 0; i < WIN; ++i) hc[i] = DEV ? A[i] * X.a0[i] + A[i + 1] * X.a1[i] + A[i + 2] * X.a2[i] : A[i] * X.a0[i] + A[i + 1] * X.a1[i];
        // A lane whose window starts at box row yoff (0 or 1) finishes its window row j = rb - 2 - yoff with rows rb-2..rb.
        const int ja = rb - 2, jb = rb - 3;                              // yoff = 0 / yoff = 1
        const bool has_a = ja >= 0 && ja < WIN, has_b = jb >= 0 && jb < WIN;
        if (has_a || has_b) {
            const float w0 = yoff ? (has_b ? Y.a0[has_b ? jb : 0] : 0.0f) : (has_a ? Y.a0[has_a ? ja : 0] : 0.0f);
            const float w1 = yoff ? (has_b ? Y.a1[has_b ? jb : 0] : 0.0f) : (has_a ? Y.a1[has_a ? ja : 0] : 0.0f);
            const float w2 = DEV ? (yoff ? (has_b ? Y.a2[has_b ? jb : 0] : 0.0f) : (has_a ? Y.a2[has_a ? ja : 0] : 0.0f)) : 0.0f;
            const int jl = rb - 2 - yoff;
            if (store_ok && jl >= 0 && jl < WIN) {
                float* oj = o + (size_t)jl * nq;
#pragma unroll
                for (int i = 0; i < WIN; ++i)                           // channel i*9+j: x offset i-r, y offset j-r
                    oj[(size_t)(i * WIN) * nq] = DEV ? hm2[i] * w0 + hm1[i] * w1 + hc[i] * w2 : hm2[i] * w0 + hm1[i] * w1;
            }
        }
#pragma unroll
        for (int i = 0; i < WIN; ++i) { hm2[i] = hm1[i]; hm1[i] = hc[i]; }
    }
}

// One workgroup = 4 independent waves; a wave = 8 consecutive groups (64 queries) of one (batch item, level).
__global__ __launch_bounds__(64 * LK_WAVES) void k_corr_lookup(const float* __restrict__ pyr, const float* __restrict__ coords,
                                                             float* __restrict__ out, PyrGeom G) {
    __shared__ __attribute__((aligned(16))) float stage[LK_WAVES * LK_WAVE_FLOATS];
    const int l = blockIdx.y, bz = blockIdx.z;
    const int nq = G.h8 * G.w8;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave_g0 = (blockIdx.x * LK_WAVES + wv) * 8;             // first group of this wave
    if (wave_g0 >= G.ngroups) return;                                 // (whole wave; there is no workgroup barrier below)
    const int grp = lane >> 3, k = lane & 7;
    const int Gi = wave_g0 + grp;
    const int qy = Gi / G.gx, qx = (Gi % G.gx) * GQ + k;
    const bool qok = Gi < G.ngroups && qx < G.w8;
    const int q = qok ? qy * G.w8 + qx : 0;
    const int hl = G.h[l], wl = G.w[l], wp = G.wp[l], sk = G.sk[l];
    const float inv = 1.0f / (float)(1 << l);
    const float cx = coords[((size_t)bz * 2 + 0) * nq + q] * inv;     // coords / 2**i  (exact)
    const float cy = coords[((size_t)bz * 2 + 1) * nq + q] * inv;
    TapAxis X, Y;
    make_taps(cx, wl, X);
    make_taps(cy, hl, Y);
    const float* lvl = pyr + G.base[l] + (size_t)bz * G.ngroups * hl * ((size_t)wp * GQ);
    float* o = out + ((size_t)bz * G.levels * WIN * WIN + (size_t)l * WIN * WIN) * nq + q;

    // ---- the group's staging box.  A lane whose window lies wholly outside its map (or whose coordinates are not
    // finite) needs no data: its output is zero (all Y weights are cleared; staged data is always finite).
    // (needw x needh = what is fetched; "fits" is tested against the full 11 x 11 because when any lane of the wave has a
    // deviating tap every lane reads 11 columns / rows of its window -- with zero weights, but from staged memory)
    const int needw = WIN + 1 + (X.dev ? 1 : 0), needh = WIN + 1 + (Y.dev ? 1 : 0);
    const bool empty = !qok || X.lo + WIN + 1 < 0 || X.lo >= wl || Y.lo + WIN + 1 < 0 || Y.lo >= hl;
    if (empty) {
#pragma unroll
        for (int i = 0; i < WIN; ++i) { Y.a0[i] = 0.0f; Y.a1[i] = 0.0f; Y.a2[i] = 0.0f; }
    }
    const int sx = X.lo + sk - (k >> l);                              // window start in skewed columns
    const int big = 0x3fffffff;
    const int gminx = group_min(empty ? big : sx), gminy = group_min(empty ? big : Y.lo);
    const bool gany = gminx != big;
    const int gx0 = gany ? (gminx >> 2) << 2 : 0, gy0 = gany ? gminy : 0;
    const int xoff_ = sx - gx0, yoff_ = Y.lo - gy0;
    const bool fits = !empty && xoff_ + WIN + 2 <= LK_BOXW && yoff_ <= 1 && yoff_ + WIN + 2 <= LK_BOXH;
    const bool slow = !empty && !fits;                               // rare: handled tap by tap below
    const int gmaxx = group_max(fits ? xoff_ + needw - 1 : -1), gmaxy = group_max(fits ? yoff_ + needh - 1 : -1);
    const int nlines = (gmaxx >> 2) + 1, nrows = gmaxy + 1;           // what the loader fetches (0 when no lane fits)
    const int xoff = fits ? xoff_ : 0, yoff = fits ? yoff_ : 0;
    const bool store_ok = qok && !slow;
    const bool any_dev = __any((X.dev | Y.dev) != 0 && fits);

    float* wstage = stage + wv * LK_WAVE_FLOATS;
    const float* mine = wstage + grp * LK_GSTRIDE + xoff * GQ + k;    // this lane's column 0 of box row 0 of a pass
    // loader role: lane -> (16-B piece of a line, line of the row, row/group selector)
    const int ld_piece = lane & 7, ld_line = (lane >> 3) & 3, ld_hi = lane >> 5;
    const size_t row_floats = (size_t)wp * GQ;

    float hm2[WIN], hm1[WIN];
#pragma unroll
    for (int i = 0; i < WIN; ++i) { hm2[i] = 0.0f; hm1[i] = 0.0f; }

#pragma unroll
    for (int pass = 0; pass < LK_BOXH / LK_RPP; ++pass) {
        f32x4 v[LK_NI];
        unsigned okbits = 0;
#pragma unroll
        for (int i = 0; i < LK_NI; ++i) {
            const int gr = 2 * i + ld_hi;                             // (group, row) pair served by this instruction half
            const int g = (2 * i) / LK_RPP;                           // group: wave-uniform, compile time
            const int row = gr % LK_RPP;
            const int s_gx0 = __builtin_amdgcn_readlane(gx0, g * 8), s_gy0 = __builtin_amdgcn_readlane(gy0, g * 8);
            const int s_nl = __builtin_amdgcn_readlane(nlines, g * 8), s_nr = __builtin_amdgcn_readlane(nrows, g * 8);
            const int rbx = LK_RPP * pass + row;
            const int y = s_gy0 + rbx, xs = s_gx0 + 4 * ld_line;
            // (bitwise &, not &&: short-circuit evaluation puts a branch in front of every load)
            const bool ok = (rbx < s_nr) & (y >= 0) & (y < hl) & (ld_line < s_nl) & (xs >= 0) & (xs + 4 <= wp) & (wave_g0 + g < G.ngroups);
            const size_t off = ok ? ((size_t)(wave_g0 + g) * hl + y) * row_floats + (size_t)xs * GQ + ld_piece * 4 : 0;
            v[i] = *(const f32x4*)(lvl + off);
            okbits |= ok ? (1u << i) : 0u;
        }
        __builtin_amdgcn_wave_barrier();                              // (compiler ordering only: the wave's LDS traffic is in order)
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < LK_NI; ++i) {
            const int gr = 2 * i + ld_hi, g = (2 * i) / LK_RPP, row = gr % LK_RPP;
            *(f32x4*)(wstage + g * LK_GSTRIDE + row * (LK_BOXW * GQ) + ld_line * 32 + ld_piece * 4) = ((okbits >> i) & 1) ? v[i] : zero;
        }
        __builtin_amdgcn_wave_barrier();
        if (any_dev) consume_pass<true>(mine, pass, X, Y, yoff, store_ok, o, nq, hm2, hm1);
        else consume_pass<false>(mine, pass, X, Y, yoff, store_ok, o, nq, hm2, hm1);
        __builtin_amdgcn_wave_barrier();
    }

    // ---- windows that do not fit the group's box (flow that jumps inside a group): bilinear taps straight from memory,
    // same tap positions and the same (horizontal, then vertical) order of operations
    if (__any(slow)) {
        if (slow) {
            const float* qmap = lvl + (size_t)Gi * hl * row_floats + (size_t)(sk - (k >> l)) * GQ + k;   // (y, x) -> qmap[y*row_floats + x*8]
            for (int i = 0; i < WIN; ++i) {
                const float px = rt_pos(rn_add(cx, (float)(i - RADIUS)), wl);
                float pfx; const int fx = safe_floor(px, pfx);
                const float wx1 = px - pfx, wx0 = (pfx + 1.0f) - px;
                for (int j = 0; j < WIN; ++j) {
                    const float py = rt_pos(rn_add(cy, (float)(j - RADIUS)), hl);
                    float pfy; const int fy = safe_floor(py, pfy);
                    const float wy1 = py - pfy, wy0 = (pfy + 1.0f) - py;
                    float t[2][2];
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            const int yy = fy + dy, xx = fx + dx;
                            const bool in = (yy >= 0) & (yy < hl) & (xx >= 0) & (xx < wl);
                            t[dy][dx] = in ? qmap[(size_t)yy * row_floats + (size_t)xx * GQ] : 0.0f;
                        }
                    const float h0 = t[0][0] * wx0 + t[0][1] * wx1, h1 = t[1][0] * wx0 + t[1][1] * wx1;
                    o[(size_t)(i * WIN + j) * nq] = h0 * wy0 + h1 * wy1;
                }
            }
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

// dense (b*nq, h_l, w_l) copy of one level (tests / debugging)
__global__ void k_corr_export(const float* __restrict__ pyr, float* __restrict__ dense, PyrGeom G, int l) {
    const int nq = G.h8 * G.w8;
    const long long qb = blockIdx.x;                          // (b, q) flattened, q row-major
    const int bz = (int)(qb / nq), q = (int)(qb % nq);
    const int qy = q / G.w8, qx = q % G.w8;
    const int Gi = qy * G.gx + (qx >> 3), k = qx & 7;
    const int h = G.h[l], w = G.w[l], wp = G.wp[l];
    const float* g = pyr + G.base[l] + ((size_t)bz * G.ngroups + Gi) * h * ((size_t)wp * GQ) + (size_t)(G.sk[l] - (k >> l)) * GQ + k;
    for (int p = threadIdx.x; p < h * w; p += blockDim.x) {
        int y = p / w, x = p - y * w;
        dense[qb * h * w + p] = g[(size_t)y * wp * GQ + (size_t)x * GQ];
    }
}

extern "C" int rpe_corr_build(const float* fmap1, const float* fmap2, int b, int c, int h8, int w8, int levels,
                              void* pyramid, void* stream) {
    PyrGeom G;
    if (!fmap1 || !fmap2 || !pyramid || c <= 0 || c > 256 || c % BK != 0 || !make_geom(b, h8, w8, levels, G)) return RPE_E_BADARG;
    if (((uintptr_t)pyramid) & 15) return RPE_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    float* pyr = (float*)pyramid;
    float* Ap = pyr + G.total;                               // scratch behind the pyramid: (b, c, mp) then (b, c, np)
    float* Bp = Ap + (size_t)b * c * G.mp;
    hipLaunchKernelGGL(k_permute_fmap, dim3(ceil_div(G.mp, 256), b * c), dim3(256), 0, s, fmap1, Ap, h8, w8, 0, G.gx, G.npx, G.mp);
    hipLaunchKernelGGL(k_permute_fmap, dim3(ceil_div(G.np, 256), b * c), dim3(256), 0, s, fmap2, Bp, h8, w8, 1, G.gx, G.npx, G.np);
    hipLaunchKernelGGL(k_corr_build, dim3(G.nbands, G.mp / BM, b), dim3(256), 0, s, (const float*)Ap, (const float*)Bp, pyr, c,
                       1.0f / sqrtf((float)c), G);
    return rpe_check_launch();
}

extern "C" int rpe_corr_lookup(const void* pyramid, const float* coords, int b, int h8, int w8, int levels, int radius,
                               float* out, void* stream) {
    PyrGeom G;
    if (!pyramid || !coords || !out || radius != RADIUS || !make_geom(b, h8, w8, levels, G)) return RPE_E_BADARG;
    hipLaunchKernelGGL(k_corr_lookup, dim3(ceil_div(G.ngroups, 8 * LK_WAVES), levels, b), dim3(64 * LK_WAVES), 0, (hipStream_t)stream,
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
