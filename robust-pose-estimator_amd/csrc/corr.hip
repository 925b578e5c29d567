// RAFT all-pairs correlation pyramid (build) and radius-4 window lookup for gfx950.
//
// Replaces CorrBlock of the reference's RAFT submodule (core/RAFT/core/corr.py: corr(), __init__ pyramid,
// __call__ lookup; call sites core/pose/pose_net.py:47,65,129).
//
// Pyramid layout (private to this file): GROUP-INTERLEAVED, SKEWED ROWS.
//   A *group* = 8 queries that are neighbours in x: g = (q.y, q.x >> 3), k = q.x & 7.  For batch item b, group g
//   and level l the eight (h_l x w_l) correlation maps of the group are stored together, query index fastest:
//       slot(y, x', k),  x' = x + sk_l - (k >> l),  sk_l = max(7 >> l, 1)  ("skew": query k's map is shifted left
//       addr = base_l + ((b*ngroups + g) * h_l + y) * wp_l*8 + x'*8 + k      by the distance its window centre moves)
//   with wp_l = (w_l + sk_l) rounded up to 4; slots whose x falls outside [0, w_l) hold zeros (they ARE the zero
//   padding of grid_sample; sk_l >= 1 makes slot x' = 0 of every row zero for every k, which gives the lookup's loader
//   an all-zero 16-B piece at offset 0 of every level region).  One 128-B line = 4 x' x 8 queries of one map row.
// Why: a query reads a 10x10 (11x11) window of ITS OWN map per level, so nothing is shared between queries -- unless
// the maps of neighbouring queries are interleaved.  Neighbouring queries have neighbouring window centres
// (centre = q + flow, and flow is smooth), so after the skew the 8 windows of a group coincide: the group needs
// ~10 rows x (10 + 3)/4 lines and every fetched line is used by all 8 queries.  Per query and level that is ~4 lines
// instead of ~7 with per-query 8x4-pixel lines (and ~15 row-major): the lookup is bound by HBM lines.  Rows are whole
// lines, so a loader pass over any subset of rows requests every line exactly once.  Flow that is NOT smooth inside a
// group costs bandwidth, never correctness: windows that do not fit one 12 x 16 staging box are served by further
// rounds of the same loader / consumer with the box re-anchored on the lanes that are left.
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
#include <type_traits>

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
            G.h[l] = h; G.w[l] = w; G.sk[l] = (7 >> l) > 1 ? (7 >> l) : 1; G.wp[l] = (w + G.sk[l] + 3) / 4 * 4;
            G.base[l] = off;
            off += (long long)b * G.ngroups * h * G.wp[l] * GQ;
            h /= 2; w /= 2;
        } else { G.h[l] = G.w[l] = G.sk[l] = G.wp[l] = 0; G.base[l] = off; }
    }
    G.total = off;
    return true;
}

// bytes of the permuted feature maps (GEMM operands, scratch behind the pyramid): 4 per element in f32, 2 in fp16; the RPE_F32X3
// experiment keeps both maps as three bf16 planes = 6
static size_t scratch_bytes(const PyrGeom& G, int c, int feature_dtype) {
    const size_t elems = (size_t)G.b * c * ((size_t)G.mp + G.np);
    return elems * (feature_dtype == RPE_F32X3 ? 6 : 4);       // (an fp16 build fits an f32-sized buffer: one size for both)
}

extern "C" size_t rpe_corr_pyramid_bytes_ex(int b, int h8, int w8, int levels, int feature_dtype) {
    PyrGeom G;
    if (!make_geom(b, h8, w8, levels, G)) return 0;
    if (feature_dtype != RPE_F32 && feature_dtype != RPE_F16 && feature_dtype != RPE_F32X3) return 0;
    return (size_t)G.total * 4 + scratch_bytes(G, 256, feature_dtype) + 256;   // (<= 256 feature channels)
}

extern "C" size_t rpe_corr_pyramid_bytes(int b, int h8, int w8, int levels) { return rpe_corr_pyramid_bytes_ex(b, h8, w8, levels, RPE_F32); }

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
    constexpr int PW = 16 >> L, R = 8 >> L, SK = (7 >> L) > 1 ? (7 >> L) : 1, S = PW + SK;
    const int hl = G.h[L], wl = G.w[L], wp = G.wp[L];
    const int x0 = px * PW, y0 = band * R;
    const unsigned rowB = (unsigned)wp * 32;                      // bytes per pyramid row; a level of one batch item is < 4 GB
    char* lv = (char*)lvl + (size_t)bz * G.ngroups * hl * rowB;
    // Item u = 2 * t + half: slot t = (g*R + yy)*S + s, queries k = 4*half .. 4*half+3 -- one 16-byte piece of the slot's 32 bytes.
    // Consecutive lanes take consecutive pieces (coalesced 16-byte stores); a piece whose four queries all have something to write
    // (data of this patch, or the zero padding outside the map) leaves as ONE store, the pieces at the patch's skewed edges --
    // where some of the four columns belong to the neighbouring patch -- as up to four 4-byte stores.  (Round 2 moved one float per
    // lane and iteration: 4x the iterations of this loop, whose index arithmetic -- not the stores -- is what the epilogue costs.)
    constexpr int NITEM = 2 * 8 * R * S, NIT = (NITEM + 255) / 256;
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
        const int u = tid + 256 * it;
        const int half = u & 1, t = u >> 1;
        const int row = t / S, sl = t - row * S;                  // (S is a compile-time constant: multiply-shift)
        const int g = row / R, yy = row % R;                      // (R is a power of two)
        const int y = y0 + yy, xs = x0 + sl, Gi = g0 + g;
        const bool valid = (u < NITEM) & (Gi < G.ngroups) & (y < hl) & (xs < wp);
        float v[4]; bool wr[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 4 * half + e;
            const int xx = sl - SK + (k >> L), x = x0 + xx;
            const bool data = (xx >= 0) & (xx < PW) & (x < wl), zero = (x < 0) | (x >= wl);
            v[e] = (valid & data) ? src[(g * 8 + k) * pitch + yy * PW + xx] : 0.0f;
            wr[e] = valid & (data | zero);                        // else: inside the map but another patch's column
        }
        float* dst = (float*)(lv + (size_t)(((unsigned)(Gi * hl + y) * (unsigned)wp + (unsigned)xs) * 32u + (unsigned)half * 16u));
        if (wr[0] & wr[1] & wr[2] & wr[3]) *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (wr[e]) dst[e] = v[e];
        }
    }
    if (px == G.npx - 1) {                                       // row padding right of the last patch's slots: zeros
        const int xe = x0 + S, ne = wp - xe;                      // x' in [xe, wp): x >= x0 + PW >= w_l for every k
        if (ne > 0) {
            for (int idx = tid; idx < 8 * R * ne * 8; idx += 256) {
                const int t = idx >> 3, k = idx & 7;
                const int se = t % ne, t2 = t / ne;
                const int yy = t2 % R, g = t2 / R;
                const int y = y0 + yy, Gi = g0 + g;
                if (Gi >= G.ngroups || y >= hl) continue;
                *(float*)(lv + (size_t)(((unsigned)(Gi * hl + y) * (unsigned)wp + (unsigned)(xe + se)) * 32u + (unsigned)k * 4u)) = 0.0f;
            }
        }
    }
}

// Level 0 with a CARRY (k_corr_build, f32 features): a workgroup walks the patches of its band left to right, and slot x' of a row
// holds x = x' - 7 + k for query k, so the first seven slots of a patch are completed by the LAST seven columns of the previous
// patch.  Those columns wait in LDS (carry[q][yy][7], zero in front of the first patch = the map's zero padding) and a patch writes
// slots [x0, x0 + 16) only -- whole 32-byte slots, each exactly once, as 16-byte pieces -- instead of leaving half-written sectors
// for the next patch to finish ~27 us later (after L2 had evicted most of them: 6.3 GB written for a 4.9 GB pyramid).  The last
// patch of the band also writes its trailing slots [x0 + 16, x0 + 23): every column they reference lies beyond the map (zeros).
#define CARRY_HALF (64 * 8 * 7)
template <bool LAST>
__device__ __forceinline__ void scatter_level0_carry(const float* __restrict__ T, const float* __restrict__ carry, float* __restrict__ lvl,
                                                     const PyrGeom& G, int bz, int g0, int band, int px, int tid) {
    constexpr int NS = LAST ? 23 : 16, NITEM = 2 * 64 * NS;
    const int hl = G.h[0], wp = G.wp[0];
    const int x0 = px * 16, y0 = band * 8;
    char* lv = (char*)lvl + (size_t)bz * G.ngroups * hl * ((unsigned)wp * 32);
    if (!LAST) {
        // 16 slots: a thread keeps its (piece, slot, row) for all eight groups of the half tile -- which LDS array each of its four
        // queries comes from, the addresses and the destination are set up once and advance by constants (as one flat index
        // decomposed per item this loop was ~60 vector instructions per 16-byte piece, a third of the epilogue's)
        const int half = tid & 1, sl = (tid >> 1) & 15, yy = tid >> 5;
        const int y = y0 + yy, xs = x0 + sl;
        if ((y >= hl) | (xs >= wp)) return;
        const float* src[4]; int step[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 4 * half + e, xx = sl - 7 + k;          // column of this patch; < 0: the previous patch's column 16 + xx
            src[e] = xx >= 0 ? T + k * TP + yy * 16 + xx : carry + (k * 8 + yy) * 7 + xx + 7;
            step[e] = xx >= 0 ? 8 * TP : 8 * 8 * 7;
        }
        char* dst = lv + (size_t)(((unsigned)(g0 * hl + y) * (unsigned)wp + (unsigned)xs) * 32u + (unsigned)half * 16u);
        const size_t dstep = (size_t)hl * ((unsigned)wp * 32u);
        const int ng = G.ngroups - g0 < 8 ? G.ngroups - g0 : 8;
#pragma unroll 1
        for (int g = 0; g < ng; ++g) {
            *(f32x4*)dst = (f32x4){*src[0], *src[1], *src[2], *src[3]};
            dst += dstep;
#pragma unroll
            for (int e = 0; e < 4; ++e) src[e] += step[e];
        }
        return;
    }
#pragma unroll 1
    for (int u = tid; u < NITEM; u += 256) {
        const int half = u & 1, t = u >> 1;
        const int row = t / NS, sl = t - row * NS;                // (compile-time divisor)
        const int g = row >> 3, yy = row & 7;
        const int y = y0 + yy, xs = x0 + sl, Gi = g0 + g;
        if ((Gi >= G.ngroups) | (y >= hl) | (xs >= wp)) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 4 * half + e, q = g * 8 + k;
            const int xx = sl - 7 + k;                            // column of this patch; < 0: the previous patch's column 16 + xx
            v[e] = xx >= 16 ? 0.0f : (xx >= 0 ? T[q * TP + yy * 16 + xx] : carry[(q * 8 + yy) * 7 + xx + 7]);
        }
        *(f32x4*)(lv + (size_t)(((unsigned)(Gi * hl + y) * (unsigned)wp + (unsigned)xs) * 32u + (unsigned)half * 16u)) = v;
    }
    if (LAST) {                                                   // row padding right of the last patch's slots: zeros
        const int xe = x0 + 23, ne = wp - xe;
        if (ne > 0) {
            for (int idx = tid; idx < 64 * ne * 8; idx += 256) {
                const int t = idx >> 3, k = idx & 7;
                const int se = t % ne, t2 = t / ne;
                const int yy = t2 & 7, g = t2 >> 3;
                const int y = y0 + yy, Gi = g0 + g;
                if (Gi >= G.ngroups || y >= hl) continue;
                *(float*)(lv + (size_t)(((unsigned)(Gi * hl + y) * (unsigned)wp + (unsigned)(xe + se)) * 32u + (unsigned)k * 4u)) = 0.0f;
            }
        }
    }
}

// Epilogue of one 128 x 128 tile: one half (64 queries = the rows of the waves with wm == hh) at a time through LDS
// (aliasing the operand tiles, which nobody reads after the K loop's last barrier), pooled to levels 1-3 and scattered.
// C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ void build_epilogue(f32x16 (&acc)[2][2], float* smem, float* __restrict__ pyr, const PyrGeom& G, float scale,
                                               int bz, int m0, int band, int px, int tid, float* carry = nullptr) {
    float* T = smem;
    float* T1 = smem + 64 * TP;
    float* T2 = T1 + 64 * 33;
    float* T3 = T2 + 64 * 9;
    const int lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
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
#ifndef CB_NO_SCATTER0
        if (carry) {
            if (px == G.npx - 1) scatter_level0_carry<true>(T, carry + hh * CARRY_HALF, pyr + G.base[0], G, bz, g0, band, px, tid);
            else scatter_level0_carry<false>(T, carry + hh * CARRY_HALF, pyr + G.base[0], G, bz, g0, band, px, tid);
        } else scatter_level<0>(T, TP, pyr + G.base[0], G, bz, g0, band, px, tid);
#endif
#ifndef CB_NO_SCATTER123
        if (G.levels > 1) scatter_level<1>(T1, 33, pyr + G.base[1], G, bz, g0, band, px, tid);
        if (G.levels > 2) scatter_level<2>(T2, 9, pyr + G.base[2], G, bz, g0, band, px, tid);
        if (G.levels > 3) scatter_level<3>(T3, 3, pyr + G.base[3], G, bz, g0, band, px, tid);
#endif
        if (carry) {                                              // this patch's last seven columns wait for the next patch
            __syncthreads();                                      // (every read of the old carry is done)
            float* ch = carry + hh * CARRY_HALF;
            for (int idx = tid; idx < CARRY_HALF; idx += 256) {
                const int j = idx % 7, qy = idx / 7;              // qy = q * 8 + yy
                ch[idx] = T[(qy >> 3) * TP + (qy & 7) * 16 + 9 + j];
            }
        }
    }
}

#define SMEM_FLOATS (64 * TP + 64 * 33 + 64 * 9 + 64 * 3)

// A: (b, K, mp) fmap1 in group order;  B: (b, K, np) fmap2 in patch order.  grid = (nbands, mp/128, b).
// K loop (round 3): both operand tiles are k-major rows of 128 consecutive floats in global memory, i.e. already the matrix
// instruction's operand order, so they arrive by LDS-DMA (global_load_lds_dwordx4, four 1 KB chunks per wave and step) into 3-deep
// rings issued two steps ahead and counted by hand (s_waitcnt vmcnt(4); see wino_common.h / conv1x1.hip), instead of passing
// through registers with a __syncthreads per step: the loop holds the 32 matrix instructions of a step, their 32 fragment reads,
// one wait, one barrier and four DMA instructions.
#define CB_TILE (BK * 128)
#define CB_BIAS 4096u
#define SMEM_MAIN (6 * CB_TILE)
__device__ __forceinline__ void cb_dma2(const float* base, unsigned v0, unsigned v1, unsigned lds_addr) {
    unsigned keep;
    const unsigned long long vb = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)vb), hi = __builtin_amdgcn_readfirstlane((unsigned)(vb >> 32));
    const float* sb = (const float*)(((unsigned long long)hi << 32) | lo);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "v"(v1), "s"(sb), "s"(lds_addr) : "memory");
}

#ifndef CB_CARRY
#define CB_CARRY 1
#endif
// CARRY = false: one workgroup per (band, patch) instead of one per band -- for launches whose band walk would leave most of the chip
// on a second, nearly empty round (sequential tracking: 2 pairs = 640 band workgroups on 512 slots; 3 200 patch workgroups fill every
// round).  No carry: level 0's skewed edge slots are completed by the neighbouring patch's workgroup, as levels 1-3's are (more
// partial-sector writes, irrelevant at this size); 48 KB of LDS, three workgroups per CU.  The values written are the same.
template <bool CARRY>
__global__ __launch_bounds__(256, CARRY ? 2 : 3) void k_corr_build(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ pyr,
                                                    int K, float scale, PyrGeom G) {
    // main loop: As[3][BK][BM] | Bs[3][BK][BN] (48 KB); epilogue (aliased): T[64][TP] | T1[64][33] | T2[64][9] | T3[64][3]
    __shared__ __attribute__((aligned(16))) float smem[SMEM_MAIN > SMEM_FLOATS ? SMEM_MAIN : SMEM_FLOATS];
    __shared__ float carry_s[CARRY ? 2 * CARRY_HALF : 1];                    // level 0: the previous patch's last seven columns (28 KB)
    float* carry = CARRY ? carry_s : nullptr;
    if (CARRY) for (int i = threadIdx.x; i < 2 * CARRY_HALF; i += 256) carry_s[i] = 0.0f;      // (published by the loop's first barrier)
    const int bz = blockIdx.z, band = CARRY ? blockIdx.x : blockIdx.x / G.npx;
    const int px_lo = CARRY ? 0 : blockIdx.x % G.npx, px_hi = CARRY ? G.npx : px_lo + 1;
    const int m0 = blockIdx.y * BM;
    const int M = G.mp, N = G.np;
    const float* Ab = A + (size_t)bz * K * M;
    const float* Bb = B + (size_t)bz * K * N;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    static_assert(BK == 16 && BM == 128 && BN == 128, "DMA roles: a wave moves four 512-byte rows of each tile per step");
    const int nk = K / BK;
    // DMA roles: wave wv moves rows k = 4 wv .. 4 wv + 3 of both tiles (two 1 KB chunks = two rows each)
    const int bl = lane & 31, bh = lane >> 5;
    unsigned aoff[2], boff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = 4 * wv + 2 * j + bh;
        aoff[j] = (unsigned)((size_t)k * M + m0 + 4 * bl) * 4u + CB_BIAS - 1024u * j;      // (16 rows of a step: < 4 GB for any map that fits the pyramid)
        boff[j] = (unsigned)((size_t)k * N + 4 * bl) * 4u + CB_BIAS - 1024u * j;
    }
    const unsigned a_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem + (unsigned)(4 * wv) * 512u;
    const unsigned b_lds = a_lds + 3u * CB_TILE * 4u;
    const float* asrc = Ab - CB_BIAS / 4;
    const int l31 = lane & 31, lh = lane >> 5;
    const float* a_l = smem + wm * 64 + l31 + lh * 128;
    const float* b_l = smem + 3 * CB_TILE + wn * 64 + l31 + lh * 128;

    for (int px = px_lo; px < px_hi; ++px) {
        const int n0 = (band * G.npx + px) * BN;
        const float* bsrc = Bb + n0 - CB_BIAS / 4;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        auto issue = [&](int s, int buf) {
            const int sc = s < nk ? s : nk - 1;                               // past the end: a harmless repeat keeps the DMA count per step constant
            cb_dma2(asrc + (size_t)sc * BK * M, aoff[0], aoff[1], a_lds + (unsigned)buf * (CB_TILE * 4u));
            cb_dma2(bsrc + (size_t)sc * BK * N, boff[0], boff[1], b_lds + (unsigned)buf * (CB_TILE * 4u));
        };
        __syncthreads();                                          // the previous patch's epilogue is done with smem
        issue(0, 0);
        issue(1, 1);
        auto step = [&](auto bufc, int s) {
            constexpr int BUF = decltype(bufc)::value, NB = (BUF + 2) % 3;
            __builtin_amdgcn_s_waitcnt(0x0F74);                               // vmcnt(4): own DMAs of step s have landed
            __builtin_amdgcn_s_barrier();                                     // ... everybody's have, and buffer (s + 2) % 3 is free
            issue(s + 2, NB);
            const float* a = a_l + BUF * CB_TILE;
            const float* b = b_l + BUF * CB_TILE;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
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
                step(I0{}, s); if (++s == nk) break;
                step(I1{}, s); if (++s == nk) break;
                step(I2{}, s); if (++s == nk) break;
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                                   // the repeats issued past the end have landed ...
        __syncthreads();                                                      // ... and every wave is done with the operand tiles the epilogue aliases
        build_epilogue(acc, smem, pyr, G, scale, bz, m0, band, px, tid, carry);
    }
}

// ---- BASELINE config 5 ("fp16 features + fp32 solve"): the feature maps are rounded to fp16 (what RAFT's mixed-precision
// encoders hand to corr.py before its .float()), the correlation runs on the 16-bit matrix cores with f32 accumulation
// (products of fp16 values are exact in f32) and the pyramid stays f32.  Operands are stored k4-interleaved,
// (b, K/4, N', 4) halves, so that a lane's four consecutive k of v_mfma_f32_32x32x8_f16 are one 8-byte LDS read.
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__global__ void k_permute_fmap_h(const float* __restrict__ f, half4* __restrict__ out, int C, int h8, int w8, int mode, int gx, int npx, int npad) {
    const int np = blockIdx.x * blockDim.x + threadIdx.x;         // grid.y = b * C/4
    if (np >= npad) return;
    int y, x;
    if (mode == 0) { const int g = np >> 3; y = g / gx; x = (g % gx) * 8 + (np & 7); }
    else { const int t = np >> 7, r = np & 127; y = (t / npx) * 8 + (r >> 4); x = (t % npx) * 16 + (r & 15); }
    const int bz = blockIdx.y / (C / 4), c4 = blockIdx.y % (C / 4);
    half4 v = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
    if (y < h8 && x < w8) {
        const float* src = f + ((size_t)bz * C + 4 * c4) * h8 * w8 + (size_t)y * w8 + x;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (_Float16)src[(size_t)e * h8 * w8];      // round to nearest even, like tensor.half()
    }
    out[(size_t)blockIdx.y * npad + np] = v;
}

__global__ __launch_bounds__(256, 2) void k_corr_build_h(const half4* __restrict__ A, const half4* __restrict__ B, float* __restrict__ pyr,
                                                         int K, float scale, PyrGeom G) {
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    half4 (*As)[BK / 4][BM] = (half4 (*)[BK / 4][BM])smem;                    // 2 x 4 x 128 x 8 B = 8 KB
    half4 (*Bs)[BK / 4][BN] = (half4 (*)[BK / 4][BN])(smem + 2 * (BK / 4) * BM * 2);
    const int bz = blockIdx.z, band = blockIdx.x;
    const int m0 = blockIdx.y * BM;
    const int M = G.mp, N = G.np, K4 = K / 4;
    const half4* Ab = A + (size_t)bz * K4 * M;
    const half4* Bb = B + (size_t)bz * K4 * N;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1, l31 = lane & 31, lh = lane >> 5;
    const int lk = tid >> 6, lc = (tid & 63) * 2;                 // loader: k-group lk, two consecutive columns (16 B)
    const int nk = K / BK;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    for (int px = 0; px < G.npx; ++px) {
        const int n0 = (band * G.npx + px) * BN;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        f32x4v ra = *(const f32x4v*)(Ab + (size_t)lk * M + m0 + lc), rb = *(const f32x4v*)(Bb + (size_t)lk * N + n0 + lc);
        __syncthreads();                                          // the previous patch's epilogue is done with smem
        *(f32x4v*)&As[0][lk][lc] = ra; *(f32x4v*)&Bs[0][lk][lc] = rb;
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) {
                const int k4 = (kt + 1) * (BK / 4) + lk;
                ra = *(const f32x4v*)(Ab + (size_t)k4 * M + m0 + lc); rb = *(const f32x4v*)(Bb + (size_t)k4 * N + n0 + lc);
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {                      // two k = 8 steps per BK = 16
                const half4 a0 = As[cur][2 * st + lh][wm * 64 + l31], a1 = As[cur][2 * st + lh][wm * 64 + 32 + l31];
                const half4 b0 = Bs[cur][2 * st + lh][wn * 64 + l31], b1 = Bs[cur][2 * st + lh][wn * 64 + 32 + l31];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x8f16(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x8f16(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x8f16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x8f16(a1, b1, acc[1][1], 0, 0, 0);
            }
            if (kt + 1 < nk) { *(f32x4v*)&As[cur ^ 1][lk][lc] = ra; *(f32x4v*)&Bs[cur ^ 1][lk][lc] = rb; }
            __syncthreads();
        }
        build_epilogue(acc, smem, pyr, G, scale, bz, m0, band, px, tid);
    }
}

// ---- RPE_F32X3: the f32 correlation with every f32 product evaluated as SIX bf16 products on the 16-bit matrix cores.
// x = hi + mid + lo EXACTLY (three bf16 pieces of 8 significand bits each: truncation split of the 24-bit significand), and
//   x y ~= hi hi' + (hi mid' + mid hi') + (hi lo' + lo hi' + mid mid'),   f32 accumulation;
// the dropped terms (mid lo', lo mid', lo lo') are below 2^-24 |x y|, i.e. below the rounding of the f32 product itself (measured,
// experiments/bf16x3_probe.hip: RMS error against f64 3.6e-7 vs 4.0e-7 for v_mfma_f32_32x32x2_f32 on the same data).  Six
// v_mfma_f32_32x32x16_bf16 (32 cycles each) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each) per 16 k: 3/8 of the matrix time.
// Operands: both maps pre-split by the permute pass into the tile-major layout  [b][tile of 128][step of 16 k][plane 3][k half 2]
// [128 pixels][8 bf16]  -- a step's operand tile is ONE contiguous 12 KB block (its LDS image: plain LDS-DMA copies, three 1 KB
// chunks per wave and operand) and a lane's eight consecutive k of one plane are one 16-byte LDS read.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
#define X3_TILE 6144                                          // u16 per operand tile of a step (12 KB)

__global__ void k_permute_fmap_x3(const float* __restrict__ f, unsigned short* __restrict__ out, int C, int h8, int w8, int mode, int gx, int npx, int npad) {
    const int np = blockIdx.x * blockDim.x + threadIdx.x;         // grid.y = b * C/8
    if (np >= npad) return;
    int y, x;
    if (mode == 0) { const int g = np >> 3; y = g / gx; x = (g % gx) * 8 + (np & 7); }
    else { const int t = np >> 7, r = np & 127; y = (t / npx) * 8 + (r >> 4); x = (t % npx) * 16 + (r & 15); }
    const int c8 = C / 8, bz = blockIdx.y / c8, cg = blockIdx.y % c8;
    u16x8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, mid = hi, lo = hi;
    if (y < h8 && x < w8) {
        const float* src = f + ((size_t)bz * C + 8 * cg) * h8 * w8 + (size_t)y * w8 + x;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = src[(size_t)e * h8 * w8];
            const unsigned u = __builtin_bit_cast(unsigned, v) & 0xFFFF0000u;                 // truncation: the pieces do not overlap
            const float r1 = v - __builtin_bit_cast(float, u);                                // exact
            const unsigned u1 = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
            const float r2 = r1 - __builtin_bit_cast(float, u1);                              // exact; at most 8 significant bits left
            hi[e] = (unsigned short)(u >> 16); mid[e] = (unsigned short)(u1 >> 16); lo[e] = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
        }
    }
    const int tile = np >> 7, r = np & 127, ks = cg >> 1, kg = cg & 1, nk = C / 16, ntiles = npad >> 7;
    unsigned short* dst = out + (((size_t)bz * ntiles + tile) * nk + ks) * X3_TILE + ((size_t)kg * 128 + r) * 8;
    *(u16x8*)(dst) = hi;
    *(u16x8*)(dst + 2 * 128 * 8) = mid;
    *(u16x8*)(dst + 4 * 128 * 8) = lo;
}

// three 1 KB chunks: global base + voff + 1024 j  ->  LDS lds_addr + 1024 j + lane * 16
__device__ __forceinline__ void x3_dma3(const void* base, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    const unsigned long long vb = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)vb), hi = __builtin_amdgcn_readfirstlane((unsigned)(vb >> 32));
    const void* sb = (const void*)(((unsigned long long)hi << 32) | lo);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sb), "s"(lds_addr) : "memory");
}

__global__ __launch_bounds__(256, 2) void k_corr_build_x3(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, float* __restrict__ pyr,
                                                          int K, float scale, PyrGeom G) {
    // main loop: 3 x (A tile | B tile) = 72 KB; the epilogue's staging (46 KB) aliases it
    __shared__ __attribute__((aligned(16))) unsigned short smem3[3 * 2 * X3_TILE];
    static_assert(3 * 2 * X3_TILE * 2 >= SMEM_FLOATS * 4, "epilogue staging fits the operand rings");
    const int bz = blockIdx.z, band = blockIdx.x, mt = blockIdx.y, m0 = mt * BM;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1, l31 = lane & 31, lh = lane >> 5;
    const int nk = K / 16, mtiles = G.mp / BM, ntiles = G.np / BN;
    const unsigned short* Ab = A + ((size_t)bz * mtiles + mt) * nk * X3_TILE;
    const unsigned voff = (unsigned)lane * 16u + (unsigned)wv * 3072u;            // this wave's three chunks of a 12 KB tile
    const unsigned a_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem3 + (unsigned)wv * 3072u;
    const unsigned b_lds = a_lds + X3_TILE * 2u;
    // fragment of (plane p, row block i): u16 index ((2 p + lh) * 128 + 64 w + 32 i + l31) * 8
    const unsigned short* a_l = smem3 + (lh * 128 + wm * 64 + l31) * 8;
    const unsigned short* b_l = smem3 + X3_TILE + (lh * 128 + wn * 64 + l31) * 8;
    for (int px = 0; px < G.npx; ++px) {
        const unsigned short* Bb = B + ((size_t)bz * ntiles + band * G.npx + px) * nk * X3_TILE;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        auto issue = [&](int s, int buf) {
            const int sc = s < nk ? s : nk - 1;                               // past the end: a harmless repeat keeps the DMA count per step constant
            x3_dma3(Ab + (size_t)sc * X3_TILE, voff, a_lds + (unsigned)buf * (4u * X3_TILE));
            x3_dma3(Bb + (size_t)sc * X3_TILE, voff, b_lds + (unsigned)buf * (4u * X3_TILE));
        };
        __syncthreads();                                          // the previous patch's epilogue is done with smem
        issue(0, 0);
        issue(1, 1);
        auto step = [&](auto bufc, int s) {
            constexpr int BUF = decltype(bufc)::value, NB = (BUF + 2) % 3;
            __builtin_amdgcn_s_waitcnt(0x0F76);                               // vmcnt(6): own DMAs of step s have landed
            __builtin_amdgcn_s_barrier();                                     // ... everybody's have, and buffer (s + 2) % 3 is free
            issue(s + 2, NB);
            const unsigned short* a = a_l + BUF * (2 * X3_TILE);
            const unsigned short* b = b_l + BUF * (2 * X3_TILE);
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[i][p] = *(const bf16x8*)(a + (p * 256 + i * 32) * 8);
                    fb[i][p] = *(const bf16x8*)(b + (p * 256 + i * 32) * 8);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {                                 // smallest terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
                }
        };
        {
            typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1; typedef std::integral_constant<int, 2> I2;
            int s = 0;
            while (true) {
                step(I0{}, s); if (++s == nk) break;
                step(I1{}, s); if (++s == nk) break;
                step(I2{}, s); if (++s == nk) break;
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                                   // the repeats issued past the end have landed ...
        __syncthreads();                                                      // ... and every wave is done with the operand tiles the epilogue aliases
        build_epilogue(acc, (float*)smem3, pyr, G, scale, bz, m0, band, px, tid);
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

#ifndef LK_WAVES
#define LK_WAVES 2
#endif
#ifndef LK_NT
#define LK_NT 5                                // bit 0: non-temporal loads of the pyramid, bit 1: non-temporal stores of the output,
                                               // bit 2: non-temporal stores in waves that retire every lane in their first round (full 256-B stores)
#endif
#ifndef LK_XCD
#define LK_XCD 1                               // 1: workgroups of one XCD take a contiguous range of (batch, level, query) slices
#endif
#ifndef LK_MINW
#define LK_MINW 2                              // waves per SIMD the register allocation must allow
#endif
#ifndef LK_RPP
#define LK_RPP 4                               // box rows staged per pass (even)
#endif
#ifndef LK_PREFETCH
#define LK_PREFETCH 0                          // 1: the next pass's loads are issued before the current pass is consumed
#endif
#define LK_BOXW 16                             // x' slots per staged row
#define LK_BOXH 12                             // box rows (LK_BOXH / LK_RPP passes)
#define LK_NPASS (LK_BOXH / LK_RPP)
#define LK_GSTRIDE (LK_RPP * LK_BOXW * GQ + 8) // floats per group: +8 rotates the banks group to group (conflict-free 4-B reads
                                               // when the groups of a wave sit at the same column offset -- the usual case)
#define LK_WAVE_FLOATS (8 * LK_GSTRIDE)
#define LK_NI (4 * LK_RPP)                     // loader instructions per pass: 8 groups x RPP rows x 16 slots x 2 halves / 64 lanes

__device__ __forceinline__ int group_min(int v) {
    v = min(v, __shfl_xor(v, 1, 64)); v = min(v, __shfl_xor(v, 2, 64)); v = min(v, __shfl_xor(v, 4, 64));
    return v;
}
__device__ __forceinline__ int group_max(int v) {
    v = max(v, __shfl_xor(v, 1, 64)); v = max(v, __shfl_xor(v, 2, 64)); v = max(v, __shfl_xor(v, 4, 64));
    return v;
}

// What a pass needs to know; the per-group words are wave-uniform (read once per round from a lane of each group).
struct LkCtx {
    const char* lvl;                             // this batch item's level region (bytes); its first 16 B are zeros (slot x' = 0)
    float* wstage; const float* mine; char* outb; // wave's LDS stage; this lane's column 0 of box row 0; batch item's output (bytes)
    unsigned row_bytes, obase, nq4;              // bytes per pyramid row; byte offset of (level, q) in outb; nq * 4
    unsigned gbase[8];                           // byte offset of box (row 0, slot 0) of each group [may wrap: only used when valid]
    unsigned gmask[8];                           // bits 0-11: box rows to fetch, bits 16-31: slots to fetch
    int ld_x, ld_half, ld_hi, yoff;
    bool store_ok, any_dev, nt_store;
    bool to_tile;                                // the destination is an LDS tile (fused lookup + convc1): plain stores
};

// Consumer of one staged pass: box rows LK_RPP*pass .. +LK_RPP-1.  DEV = false: no tap of any lane of the wave deviates
// (the usual case away from exactly-integer coordinates): 10 columns, two weights per tap on both axes.
template <bool DEV, int PASS>
__device__ __forceinline__ void consume_pass(const LkCtx& C, const TapAxis& X, const TapAxis& Y, float (&hm2)[WIN], float (&hm1)[WIN]) {
#pragma unroll
    for (int rr = 0; rr < LK_RPP; ++rr) {
        constexpr int dummy = 0; (void)dummy;
        const int rb = LK_RPP * PASS + rr;                               // box row (compile time)
        float A[WIN + 2], hc[WIN];
#pragma unroll
        for (int c = 0; c < (DEV ? WIN + 2 : WIN + 1); ++c) A[c] = C.mine[rr * (LK_BOXW * GQ) + c * GQ];
#pragma unroll
        for (int i = 0; i < WIN; ++i) hc[i] = DEV ? A[i] * X.a0[i] + A[i + 1] * X.a1[i] + A[i + 2] * X.a2[i] : A[i] * X.a0[i] + A[i + 1] * X.a1[i];
        // A lane whose window starts at box row yoff (0 or 1) finishes its window row j = rb - 2 - yoff with rows rb-2..rb.
        const int ja = rb - 2, jb = rb - 3;                              // yoff = 0 / yoff = 1
        const bool has_a = ja >= 0 && ja < WIN, has_b = jb >= 0 && jb < WIN;
        if (has_a || has_b) {
            const float w0 = C.yoff ? (has_b ? Y.a0[has_b ? jb : 0] : 0.0f) : (has_a ? Y.a0[has_a ? ja : 0] : 0.0f);
            const float w1 = C.yoff ? (has_b ? Y.a1[has_b ? jb : 0] : 0.0f) : (has_a ? Y.a1[has_a ? ja : 0] : 0.0f);
            const float w2 = DEV ? (C.yoff ? (has_b ? Y.a2[has_b ? jb : 0] : 0.0f) : (has_a ? Y.a2[has_a ? ja : 0] : 0.0f)) : 0.0f;
            const int jl = rb - 2 - C.yoff;
            if (C.store_ok && jl >= 0 && jl < WIN) {
                const unsigned oj = C.obase + (unsigned)jl * C.nq4;     // channel i*9+j: x offset i-r, y offset j-r
#pragma unroll
                for (int i = 0; i < WIN; ++i) {
                    const float val = DEV ? hm2[i] * w0 + hm1[i] * w1 + hc[i] * w2 : hm2[i] * w0 + hm1[i] * w1;
                    float* dst = (float*)(C.outb + (size_t)(oj + (unsigned)(i * WIN) * C.nq4));
                    if (!C.to_tile && ((LK_NT & 2) || ((LK_NT & 4) && C.nt_store))) __builtin_nontemporal_store(val, dst); else *dst = val;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < WIN; ++i) { hm2[i] = hm1[i]; hm1[i] = hc[i]; }
    }
}

// The loads of pass PASS: instruction i serves (group g = 2i / RPP, rows (2i + ld_hi) % RPP); lanes outside what the group
// needs read the zero piece at offset 0 instead (so what reaches LDS needs no select).
template <int PASS>
__device__ __forceinline__ void issue_loads(const LkCtx& C, f32x4 (&v)[LK_NI]) {
#pragma unroll
    for (int i = 0; i < LK_NI; ++i) {
        const int g = (2 * i) / LK_RPP;                                  // wave-uniform, compile time
        const int rbx = LK_RPP * PASS + (2 * i) % LK_RPP + C.ld_hi;      // box row of this lane
        const unsigned m = C.gmask[g];
        const bool ok = ((m >> rbx) & (m >> (16 + C.ld_x)) & 1u) != 0;
        const unsigned off = C.gbase[g] + (unsigned)rbx * C.row_bytes + (unsigned)(C.ld_x * 32 + C.ld_half * 16);
        const f32x4* src = (const f32x4*)(C.lvl + (size_t)(ok ? off : 0u));
        v[i] = (LK_NT & 1) ? __builtin_nontemporal_load(src) : *src;
    }
}

// One loader + consumer pass (PASS is a template parameter so that every row index is a compile-time constant: left to
// `#pragma unroll` the six-pass variant stayed a loop and the weight arrays went to scratch).
template <int PASS>
__device__ __forceinline__ void lookup_pass(const LkCtx& C, const TapAxis& X, const TapAxis& Y, float (&hm2)[WIN], float (&hm1)[WIN],
                                            f32x4 (&vcur)[LK_NI]) {
    f32x4 vnext[LK_NI];
    if constexpr (LK_PREFETCH && PASS + 1 < LK_NPASS) {
        issue_loads<PASS + 1>(C, vnext);
        __builtin_amdgcn_sched_barrier(0);                               // (keep them ahead of this pass's arithmetic)
    }
#pragma unroll
    for (int i = 0; i < LK_NI; ++i) {
        const int g = (2 * i) / LK_RPP, row = (2 * i) % LK_RPP;
        *(f32x4*)(C.wstage + g * LK_GSTRIDE + (row + C.ld_hi) * (LK_BOXW * GQ) + C.ld_x * GQ + C.ld_half * 4) = vcur[i];
    }
    __builtin_amdgcn_wave_barrier();                                     // (compiler ordering only: the wave's LDS traffic is in order)
    if (C.any_dev) consume_pass<true, PASS>(C, X, Y, hm2, hm1);
    else consume_pass<false, PASS>(C, X, Y, hm2, hm1);
    __builtin_amdgcn_wave_barrier();
    if constexpr (PASS + 1 < LK_NPASS) {
        if constexpr (!LK_PREFETCH) issue_loads<PASS + 1>(C, vnext);
        lookup_pass<PASS + 1>(C, X, Y, hm2, hm1, vnext);
    }
}

// The same with the loads of ALL passes of the round issued up front (3 x 12 x 16 bytes per lane in flight: one memory round trip per round
// instead of three).  For the fused lookup + convc1 kernel, whose four waves per CU have nothing else to hide the latency behind and 512
// registers each to hold the data.  (Measured: no gain -- 330 vs 325 us at batch 32 --: the kernel is bound by its matrix phase; kept, it is
// never slower.)
template <int PASS>
__device__ __forceinline__ void stage_and_consume(const LkCtx& C, const TapAxis& X, const TapAxis& Y, float (&hm2)[WIN], float (&hm1)[WIN],
                                                  f32x4 (&v)[LK_NPASS][LK_NI]) {
#pragma unroll
    for (int i = 0; i < LK_NI; ++i) {
        const int g = (2 * i) / LK_RPP, row = (2 * i) % LK_RPP;
        *(f32x4*)(C.wstage + g * LK_GSTRIDE + (row + C.ld_hi) * (LK_BOXW * GQ) + C.ld_x * GQ + C.ld_half * 4) = v[PASS][i];
    }
    __builtin_amdgcn_wave_barrier();
    if (C.any_dev) consume_pass<true, PASS>(C, X, Y, hm2, hm1);
    else consume_pass<false, PASS>(C, X, Y, hm2, hm1);
    __builtin_amdgcn_wave_barrier();
    if constexpr (PASS + 1 < LK_NPASS) stage_and_consume<PASS + 1>(C, X, Y, hm2, hm1, v);
}
template <int PASS>
__device__ __forceinline__ void issue_all(const LkCtx& C, f32x4 (&v)[LK_NPASS][LK_NI]) {
    issue_loads<PASS>(C, v[PASS]);
    if constexpr (PASS + 1 < LK_NPASS) issue_all<PASS + 1>(C, v);
}

// The round rule (shared by k_corr_lookup and the k_corr_rounds diagnostic): per group the staging box is anchored at the topmost
// pending window row and, among the lanes within one row of it, the leftmost window column; a lane is served in this round when
// its 11 x 11 window fits the 12 x 16 box.
struct LkRound { int gx0, gy0, xoff, yoff, nslots, nrows; bool fits; };
__device__ __forceinline__ LkRound plan_round(bool pending, int ylo, int sx, int needw, int needh) {
    const int big = 0x3fffffff;
    LkRound R;
    const int gy0m = group_min(pending ? ylo : big);
    const bool near_top = pending && ylo - gy0m <= 1;
    const int gx0m = group_min(near_top ? sx : big);
    const bool gany = gx0m != big;
    R.gx0 = gany ? gx0m : 0; R.gy0 = gany ? gy0m : 0;
    R.xoff = sx - R.gx0; R.yoff = ylo - R.gy0;
    R.fits = near_top && R.xoff >= 0 && R.xoff + WIN + 2 <= LK_BOXW;
    R.nslots = group_max(R.fits ? R.xoff + needw : 0); R.nrows = group_max(R.fits ? R.yoff + needh : 0);   // what the loader fetches
    return R;
}

// The lookup of ONE wave: 8 consecutive groups (64 queries, first group wave_g0) of one (batch item bz, level l), staged through the
// wave's own LDS stage.  The 81 values of a lane's query go to outb + obase + (i * 9 + j) * nq4 (bytes): channel planes of the (b, 324,
// h8, w8) output for k_corr_lookup, or rows of a workgroup's LDS tile for the fused lookup + convc1 kernel (a generic pointer serves both).
struct NoGate { __device__ __forceinline__ void operator()() const {} };
// ``gate`` is called once, after the first round's loads have been issued and before anything is stored to ``outb`` (the pipelined fused
// kernel waits there until the matrix waves are done with the tile rows this level overwrites).
template <bool DEEP, typename Gate = NoGate>
__device__ __forceinline__ void lookup_wave(const float* __restrict__ pyr, const float* __restrict__ coords, char* outb, unsigned nq4, int tile_q,
                                            float* wstage, const PyrGeom& G, int wave_g0, int l, int bz, Gate gate = Gate()) {
    const int nq = G.h8 * G.w8;
    const int lane = threadIdx.x & 63;
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

    // ---- A lane whose window lies wholly outside its map (or whose coordinates are not finite) needs no data: its
    // output is zero (all Y weights are cleared; staged data is always finite) and is stored in the first round.
    // needw x needh = what is fetched for a lane; "fits" is tested against the full 11 x 11 because when any lane of the
    // wave has a deviating tap every lane reads 11 columns / rows of its window (zero weights, but staged memory).
    const int needw = WIN + 1 + (X.dev ? 1 : 0), needh = WIN + 1 + (Y.dev ? 1 : 0);
    const bool empty = !qok || X.lo + WIN + 1 < 0 || X.lo >= wl || Y.lo + WIN + 1 < 0 || Y.lo >= hl;
    if (empty) {
#pragma unroll
        for (int i = 0; i < WIN; ++i) { Y.a0[i] = 0.0f; Y.a1[i] = 0.0f; Y.a2[i] = 0.0f; }
    }
    const int sx = X.lo + sk - (k >> l);                              // window start in skewed columns

    LkCtx C;
    C.row_bytes = (unsigned)wp * GQ * 4; C.nq4 = nq4;
    C.lvl = (const char*)(pyr + G.base[l]) + (size_t)bz * G.ngroups * hl * C.row_bytes;       // (a level of one batch item is < 4 GB)
    C.outb = outb;
    C.obase = ((unsigned)l * WIN * WIN * (nq4 / 4) + (unsigned)(tile_q >= 0 ? tile_q : q)) * 4;
    C.wstage = wstage;
    C.to_tile = tile_q >= 0;
    // loader role: lane -> (16-B piece: x' slot ld_x of the staged row, half ld_half; row selector ld_hi)
    C.ld_x = (lane >> 1) & 15; C.ld_half = lane & 1; C.ld_hi = lane >> 5;
    const unsigned my_rows = (unsigned)Gi * hl;                       // first pyramid row of this lane's group

    // ---- Rounds.  Per group the staging box is anchored at the topmost pending window row and, among the lanes within one
    // row of it, the leftmost window column: that lane always fits, so every round retires at least one lane per group.
    // Smooth flow retires all 8 lanes of every group in the first round; lanes of a group whose windows are more than
    // 5 columns / 1 row apart (flow discontinuities) take further rounds -- more traffic, same arithmetic.
    bool pending = !empty;
    bool first = true;
    for (;;) {
        const LkRound R = plan_round(pending, Y.lo, sx, needw, needh);
        const int gx0 = R.gx0, gy0 = R.gy0, xoff_ = R.xoff, yoff_ = R.yoff, nslots = R.nslots, nrows = R.nrows;
        const bool fits = R.fits;
        // box rows / slots that exist in the map and are needed, as bit masks (empty when nothing fits)
        int r_lo = -gy0 > 0 ? -gy0 : 0, r_hi = hl - gy0 < nrows ? hl - gy0 : nrows;
        int x_lo = -gx0 > 0 ? -gx0 : 0, x_hi = wp - gx0 < nslots ? wp - gx0 : nslots;
        r_lo = r_lo > 16 ? 16 : r_lo; x_lo = x_lo > 16 ? 16 : x_lo;
        r_hi = r_hi < r_lo ? r_lo : r_hi; x_hi = x_hi < x_lo ? x_lo : x_hi;
        const unsigned rmask = ((1u << r_hi) - 1u) & ~((1u << r_lo) - 1u), xmask = ((1u << x_hi) - 1u) & ~((1u << x_lo) - 1u);
        const unsigned my_mask = (Gi < G.ngroups ? (rmask & 0xfffu) : 0u) | (xmask << 16);
        const unsigned my_base = (my_rows + (unsigned)gy0) * C.row_bytes + (unsigned)gx0 * (GQ * 4);     // wraps when gy0/gx0 < 0: those pieces are masked
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            C.gbase[g] = (unsigned)__builtin_amdgcn_readlane((int)my_base, g * 8);
            C.gmask[g] = (unsigned)__builtin_amdgcn_readlane((int)my_mask, g * 8);
        }
        C.yoff = fits ? yoff_ : 0;
        C.store_ok = qok && (fits || (first && empty));
        C.any_dev = __any((X.dev | Y.dev) != 0 && fits);
        C.nt_store = first && !__any(pending && !fits);
        C.mine = C.wstage + grp * LK_GSTRIDE + (fits ? xoff_ : 0) * GQ + k;    // this lane's column 0 of box row 0 of a pass

        float hm2[WIN], hm1[WIN];
#pragma unroll
        for (int i = 0; i < WIN; ++i) { hm2[i] = 0.0f; hm1[i] = 0.0f; }
        if constexpr (DEEP) {
            f32x4 v[LK_NPASS][LK_NI];
            issue_all<0>(C, v);
            __builtin_amdgcn_sched_barrier(0);
            if (first) gate();
            stage_and_consume<0>(C, X, Y, hm2, hm1, v);
        } else {
            f32x4 v0[LK_NI];
            issue_loads<0>(C, v0);
            if (first) gate();
            lookup_pass<0>(C, X, Y, hm2, hm1, v0);
        }
        pending = pending && !fits;
        first = false;
        if (!__any(pending)) break;
    }
}

// One workgroup = LK_WAVES independent waves; a wave = 8 consecutive groups (64 queries) of one (batch item, level).
__global__ __launch_bounds__(64 * LK_WAVES, LK_MINW) void k_corr_lookup(const float* __restrict__ pyr, const float* __restrict__ coords,
                                                                      float* __restrict__ out, PyrGeom G) {
    __shared__ __attribute__((aligned(16))) float stage[LK_WAVES * LK_WAVE_FLOATS];
    int bx = blockIdx.x, l = blockIdx.y, bz = blockIdx.z;
    if (LK_XCD) {   // consecutive workgroup ids go round-robin over the 8 XCDs: give each XCD a contiguous range of the grid
        const unsigned total = gridDim.x * gridDim.y * gridDim.z, L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned c = L & 7, sidx = L >> 3, q8 = total >> 3, r8 = total & 7;
        const unsigned nl = c * q8 + (c < r8 ? c : r8) + sidx;
        bx = nl % gridDim.x; l = (nl / gridDim.x) % gridDim.y; bz = nl / (gridDim.x * gridDim.y);
    }
    const int nq = G.h8 * G.w8;
    const int wv = threadIdx.x >> 6;
    const int wave_g0 = (bx * LK_WAVES + wv) * 8;                     // first group of this wave
    if (wave_g0 >= G.ngroups) return;                                 // (whole wave; there is no workgroup barrier below)
    lookup_wave<false>(pyr, coords, (char*)(out + (size_t)bz * G.levels * WIN * WIN * nq), (unsigned)nq * 4, -1, stage + wv * LK_WAVE_FLOATS, G, wave_g0, l, bz);
}

// ------------------------------------------------------------------------------------------------ lookup fused into convc1
// BasicMotionEncoder.forward starts with cor = relu(convc1(corr)) (upstream core/RAFT/core/update.py, restated in oracle/raft.py; the
// reference's call site is core/pose/pose_net.py:65): the 324-channel lookup result exists only to be contracted to 256 channels by a 1x1
// convolution.  k_lookup_conv1x1 never writes it: a workgroup = 4 waves looks up the four levels of 64 queries (wave = level; the loader /
// consumer of k_corr_lookup, lookup_wave) into an LDS tile [336 channels][64 queries] (84 KB; rows 324..335 are zeros: the K padding of
// the last 16-channel step), and after one barrier the same four waves run the GEMM out[256][64] = W[256][324] tile on the f32 matrix
// cores: wave = 64 output channels x 64 queries (2 x 2 blocks of v_mfma_f32_32x32x2_f32), B fragments from the tile, A fragments straight
// from global memory (the 332 KB of packed weights live in L2; each lane's 16 values of a step are one contiguous 64-byte piece,
// fetched one step ahead).  The products are added in k_conv1x1's order (per 16-channel step, matrix instruction j adds channels j and
// 8 + j): the result is BIT-IDENTICAL to rpe_corr_lookup followed by rpe_conv1x1 / rpe_conv_fused (tests/test_gpu_corr.py).
// Measured (MI355X, 640x512; tools/bench_lookup_conv.py): 2 pairs (a tracker frame) 35.3 us against 16.1 + 30.9 = 45 us back to back;
// 32 pairs (the bench step) 325-330 us against 308-312: with 150 KB of LDS one workgroup owns a CU, so its lookup phase (memory) and its
// matrix phase never overlap, which two kernels at 3+ workgroups per CU do -- the host side uses it for small passes only (raft.py).
// Tried on top, both measured: all three passes' loads of a round in flight at once (lookup_wave<DEEP>: 330 vs 325, kept); the 21 steps
// written out with the next step's fragments requested ahead of the matrix instructions (361 us: the compiler hoists 48 weight loads in
// front of the lookup; dropped).
#define FZ_COUT 256
#define FZ_CIN (MAX_LEVELS * WIN * WIN)                            // 324
#define FZ_STEPS ((FZ_CIN + 15) / 16)                              // 21
#define FZ_TILE_FLOATS (FZ_STEPS * 16 * 64)
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct FzP {
    const float* pyr; const float* coords; const float* wp; const float* bias;
    float* out; long long obs; float* out2; long long o2bs;
    int relu;
};

template <bool OUT2>
__global__ __launch_bounds__(256, 1) void k_lookup_conv1x1(FzP P, PyrGeom G) {
    __shared__ __attribute__((aligned(16))) float tile[FZ_TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float stage[4 * LK_WAVE_FLOATS];
    __shared__ float bias_s[FZ_COUT];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bz = blockIdx.y;
    const int wave_g0 = blockIdx.x * 8;                                 // the workgroup's 8 groups = 64 consecutive queries (w8 % 8 == 0)
    const int nq = G.h8 * G.w8;
    for (int i = tid; i < (FZ_STEPS * 16 - FZ_CIN) * 64; i += 256) tile[FZ_CIN * 64 + i] = 0.0f;
    bias_s[tid] = P.bias ? P.bias[tid] : 0.0f;
    // A fragments of step 0 are requested before the lookup: they arrive during it
    const f32x4* wsrc = (const f32x4*)P.wp + ((size_t)wv * FZ_STEPS * 64 + lane) * 4;
    f32x4 a_cur[4], a_nxt[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) a_cur[c] = wsrc[c];
    lookup_wave<true>(P.pyr, P.coords, (char*)tile, 64u * 4u, lane, stage + wv * LK_WAVE_FLOATS, G, wave_g0, wv, bz);
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int l31 = lane & 31, lh = lane >> 5;
    const float* b_l = tile + (8 * lh) * 64 + l31;
#pragma unroll 1
    for (int s = 0; s < FZ_STEPS; ++s) {
        if (s + 1 < FZ_STEPS) {
#pragma unroll
            for (int c = 0; c < 4; ++c) a_nxt[c] = wsrc[(size_t)(s + 1) * 256 + c];
        }
        const float* b = b_l + s * (16 * 64);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a0 = a_cur[j >> 1][(j & 1) * 2], a1 = a_cur[j >> 1][(j & 1) * 2 + 1];
            const float b0 = b[j * 64], b1 = b[j * 64 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) a_cur[c] = a_nxt[c];
    }
    // ---- epilogue (k_conv1x1's): C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float* ob = P.out + (size_t)bz * P.obs;
    float* ob2 = OUT2 ? P.out2 + (size_t)bz * P.o2bs : nullptr;
    const int co_w = wv * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = wave_g0 * GQ + j * 32 + l31;
        const bool qok = q < nq;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_w + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[i][j][r] + bias_s[co];
                if (P.relu) v = v < 0.0f ? 0.0f : v;                            // NaN stays NaN, like torch.relu
                if (qok) {
                    ob[(size_t)co * nq + q] = v;
                    if (OUT2) ob2[(size_t)co * nq + q] = v;
                }
            }
    }
}

// The same for passes of MORE than one round of workgroups: a PERSISTENT workgroup of 8 waves walks its tiles (64 queries each) with the two
// phases pipelined through the ONE tile the LDS has room for.  Waves 0-3 (matrix waves, one per SIMD) contract tile n; waves 4-7 (lookup
// waves, wave = level) look tile n + 1 up meanwhile: their loads go out at once, and a level's rows of the tile are overwritten as soon as
// the matrix waves have passed the last 16-channel step that reads them (level l occupies rows 81 l .. 81 l + 80: free after steps 5, 10, 15,
// 20).  The matrix waves in turn wait for a level only at the step that first reads it (steps 0, 5, 10, 15) -- by then the lookup of that
// level has long finished, so the matrix pipe runs tile after tile and the lookup costs no time of its own.  Progress words in LDS (a
// step counter per matrix wave, a tile counter per level), polled; no workgroup barrier after the prologue.  Same products in the same
// order as k_lookup_conv1x1: bit-identical.
struct TileGate {
    volatile int* prog; int need;                                   // the four matrix waves' step counters; steps that must be complete
    __device__ __forceinline__ void operator()() const {
        if (need <= 0) return;
        for (int spin = 0; spin < (1 << 24); ++spin) {
            const int a = prog[0], b = prog[1], c = prog[2], d = prog[3];
            if (a >= need && b >= need && c >= need && d >= need) break;
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
};
__device__ __forceinline__ void wait_level(volatile int* done, int need) {
    for (int spin = 0; spin < (1 << 24); ++spin) {
        if (*done >= need) break;
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <bool OUT2>
__global__ __launch_bounds__(512, 1) void k_lookup_conv1x1_pipe(FzP P, PyrGeom G, int tiles_per_item, int total_tiles) {
    __shared__ __attribute__((aligned(16))) float tile[FZ_TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float stage[4 * LK_WAVE_FLOATS];
    __shared__ float bias_s[FZ_COUT];
    __shared__ volatile int prog[4], done[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = G.h8 * G.w8;
    for (int i = tid; i < (FZ_STEPS * 16 - FZ_CIN) * 64; i += 512) tile[FZ_CIN * 64 + i] = 0.0f;
    if (tid < FZ_COUT) bias_s[tid] = P.bias ? P.bias[tid] : 0.0f;
    if (tid < 4) { prog[tid] = 0; done[tid] = 0; }
    __syncthreads();
    const int ntiles = total_tiles > (int)blockIdx.x ? (total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;   // tiles blockIdx.x, + gridDim.x, ...

    if (wv >= 4) {
        // ---------------- lookup wave: level l of every tile of this workgroup
        const int l = wv - 4;
        for (int k = 0; k < ntiles; ++k) {
            const int t = blockIdx.x + k * gridDim.x;
            const int bz = t / tiles_per_item, wave_g0 = (t % tiles_per_item) * 8;
            // rows 81 l .. 81 l + 80 are last read by step (81 l + 80) / 16 of the previous tile
            const TileGate gate{prog, k == 0 ? 0 : (k - 1) * FZ_STEPS + (81 * l + 80) / 16 + 1};
            lookup_wave<false>(P.pyr, P.coords, (char*)tile, 64u * 4u, lane, stage + l * LK_WAVE_FLOATS, G, wave_g0, l, bz, gate);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // (the wave's LDS stores are complete and ordered before the flag)
            if (lane == 0) done[l] = k + 1;
        }
        return;
    }
    // -------------------- matrix wave: output channels 64 wv .. 64 wv + 63 of every tile
    const f32x4* wsrc = (const f32x4*)P.wp + ((size_t)wv * FZ_STEPS * 64 + lane) * 4;
    const int l31 = lane & 31, lh = lane >> 5;
    const float* b_l = tile + (8 * lh) * 64 + l31;
    for (int k = 0; k < ntiles; ++k) {
        const int t = blockIdx.x + k * gridDim.x;
        const int bz = t / tiles_per_item, wave_g0 = (t % tiles_per_item) * 8;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        f32x4 a_cur[4], a_nxt[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) a_cur[c] = wsrc[c];
#pragma unroll 1
        for (int s = 0; s < FZ_STEPS; ++s) {
            // step s reads rows 16 s .. 16 s + 15: it is the first to read level (16 s + 15) / 81 when that differs from the step before
            const int lv = (16 * s + 15) / 81;
            if (s == 0 || lv != (16 * s - 1) / 81) wait_level(&done[lv < 4 ? lv : 3], k + 1);
            if (s + 1 < FZ_STEPS) {
#pragma unroll
                for (int c = 0; c < 4; ++c) a_nxt[c] = wsrc[(size_t)(s + 1) * 256 + c];
            }
            const float* b = b_l + s * (16 * 64);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a0 = a_cur[j >> 1][(j & 1) * 2], a1 = a_cur[j >> 1][(j & 1) * 2 + 1];
                const float b0 = b[j * 64], b1 = b[j * 64 + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) a_cur[c] = a_nxt[c];
            // this step's fragment reads have returned (the matrix instructions consumed them): its rows may be overwritten
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) prog[wv] = k * FZ_STEPS + s + 1;
        }
        float* ob = P.out + (size_t)bz * P.obs;
        float* ob2 = OUT2 ? P.out2 + (size_t)bz * P.o2bs : nullptr;
        const int co_w = wv * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = wave_g0 * GQ + j * 32 + l31;
            const bool qok = q < nq;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co_w + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v = acc[i][j][r] + bias_s[co];
                    if (P.relu) v = v < 0.0f ? 0.0f : v;
                    if (qok) {
                        ob[(size_t)co * nq + q] = v;
                        if (OUT2) ob2[(size_t)co * nq + q] = v;
                    }
                }
        }
    }
}

// weight (256, 324, 1, 1) -> [wave = co / 64][step = ci / 16][lane][16]: value 2 j + i of lane (l31, lh) = W[64 wave + 32 i + l31][16 step + j + 8 lh]
// (zero beyond ci = 323): a lane's fragments of a step are one 64-byte piece
__global__ void k_lookup_conv_pack(const float* __restrict__ w, float* __restrict__ wp) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 4 * FZ_STEPS * 64 * 16) return;
    const int v = e & 15, lane = (e >> 4) & 63, step = (e >> 10) % FZ_STEPS, wave = (e >> 10) / FZ_STEPS;
    const int j = v >> 1, i = v & 1, l31 = lane & 31, lh = lane >> 5;
    const int co = 64 * wave + 32 * i + l31, ci = 16 * step + j + 8 * lh;
    wp[e] = ci < FZ_CIN ? w[(size_t)co * FZ_CIN + ci] : 0.0f;
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

// Diagnostic: how many rounds k_corr_lookup takes per (batch item, level, group) for these coordinates, and how many 128-B lines
// its loader requests -- the same make_taps / plan_round / mask arithmetic, no loads.  rounds[(b*levels + l)*ngroups + g], lines likewise.
__global__ void k_corr_rounds(const float* __restrict__ coords, int32_t* __restrict__ rounds, int32_t* __restrict__ lines, PyrGeom G) {
    const int l = blockIdx.y, bz = blockIdx.z;
    const int nq = G.h8 * G.w8;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave_g0 = (blockIdx.x * (blockDim.x >> 6) + wv) * 8;
    if (wave_g0 >= G.ngroups) return;
    const int grp = lane >> 3, k = lane & 7;
    const int Gi = wave_g0 + grp;
    const int qy = Gi / G.gx, qx = (Gi % G.gx) * GQ + k;
    const bool qok = Gi < G.ngroups && qx < G.w8;
    const int q = qok ? qy * G.w8 + qx : 0;
    const int hl = G.h[l], wl = G.w[l], wp = G.wp[l], sk = G.sk[l];
    const float inv = 1.0f / (float)(1 << l);
    TapAxis X, Y;
    make_taps(coords[((size_t)bz * 2 + 0) * nq + q] * inv, wl, X);
    make_taps(coords[((size_t)bz * 2 + 1) * nq + q] * inv, hl, Y);
    const int needw = WIN + 1 + (X.dev ? 1 : 0), needh = WIN + 1 + (Y.dev ? 1 : 0);
    const bool empty = !qok || X.lo + WIN + 1 < 0 || X.lo >= wl || Y.lo + WIN + 1 < 0 || Y.lo >= hl;
    const int sx = X.lo + sk - (k >> l);
    bool pending = !empty;
    int nr = 0, nl = 0;
    for (;;) {
        const LkRound R = plan_round(pending, Y.lo, sx, needw, needh);
        int r_lo = -R.gy0 > 0 ? -R.gy0 : 0, r_hi = hl - R.gy0 < R.nrows ? hl - R.gy0 : R.nrows;
        int x_lo = -R.gx0 > 0 ? -R.gx0 : 0, x_hi = wp - R.gx0 < R.nslots ? wp - R.gx0 : R.nslots;
        r_lo = r_lo > 16 ? 16 : r_lo; x_lo = x_lo > 16 ? 16 : x_lo;
        r_hi = r_hi < r_lo ? r_lo : r_hi; x_hi = x_hi < x_lo ? x_lo : x_hi;
        r_hi = r_hi > 12 ? 12 : r_hi;
        // 128-B lines = 4 slots: count the distinct lines the [x_lo, x_hi) slot run of each fetched row touches
        const int s0 = R.gx0 + x_lo, s1 = R.gx0 + x_hi;
        const int lines_row = x_hi > x_lo ? ((s1 + 3) >> 2) - (s0 >> 2) : 0;
        if (group_max(R.fits ? 1 : 0)) { nr += 1; nl += lines_row * (r_hi - r_lo); }
        pending = pending && !R.fits;
        if (!__any(pending)) break;
    }
    if (k == 0 && Gi < G.ngroups) {
        const size_t o = ((size_t)bz * G.levels + l) * G.ngroups + Gi;
        rounds[o] = nr; lines[o] = nl;
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

extern "C" int rpe_corr_build_ex(const float* fmap1, const float* fmap2, int b, int c, int h8, int w8, int levels, int feature_dtype,
                                 void* pyramid, void* stream) {
    PyrGeom G;
    if (feature_dtype != RPE_F32 && feature_dtype != RPE_F16 && feature_dtype != RPE_F32X3) return RPE_E_BADARG;
    if (!fmap1 || !fmap2 || !pyramid || c <= 0 || c > 256 || c % BK != 0 || !make_geom(b, h8, w8, levels, G)) return RPE_E_BADARG;
    if (((uintptr_t)pyramid) & 15) return RPE_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    float* pyr = (float*)pyramid;
    float* Ap = pyr + G.total;                               // scratch behind the pyramid: (b, c, mp) then (b, c, np)
    float* Bp = Ap + (size_t)b * c * G.mp;
    if (feature_dtype == RPE_F16) {                           // same scratch, half the bytes: (b, c/4, mp, 4) then (b, c/4, np, 4) halves
        half4* Ah = (half4*)Ap;
        half4* Bh = Ah + (size_t)b * (c / 4) * G.mp;
        hipLaunchKernelGGL(k_permute_fmap_h, dim3(ceil_div(G.mp, 256), b * (c / 4)), dim3(256), 0, s, fmap1, Ah, c, h8, w8, 0, G.gx, G.npx, G.mp);
        hipLaunchKernelGGL(k_permute_fmap_h, dim3(ceil_div(G.np, 256), b * (c / 4)), dim3(256), 0, s, fmap2, Bh, c, h8, w8, 1, G.gx, G.npx, G.np);
        hipLaunchKernelGGL(k_corr_build_h, dim3(G.nbands, G.mp / BM, b), dim3(256), 0, s, (const half4*)Ah, (const half4*)Bh, pyr, c,
                           1.0f / sqrtf((float)c), G);
        return rpe_check_launch();
    }
    if (feature_dtype == RPE_F32X3) {                         // f32 features, f32 products as six bf16 products (k_corr_build_x3)
        unsigned short* A3 = (unsigned short*)Ap;
        unsigned short* B3 = A3 + (size_t)b * c * G.mp * 3;
        hipLaunchKernelGGL(k_permute_fmap_x3, dim3(ceil_div(G.mp, 256), b * (c / 8)), dim3(256), 0, s, fmap1, A3, c, h8, w8, 0, G.gx, G.npx, G.mp);
        hipLaunchKernelGGL(k_permute_fmap_x3, dim3(ceil_div(G.np, 256), b * (c / 8)), dim3(256), 0, s, fmap2, B3, c, h8, w8, 1, G.gx, G.npx, G.np);
        hipLaunchKernelGGL(k_corr_build_x3, dim3(G.nbands, G.mp / BM, b), dim3(256), 0, s, (const unsigned short*)A3, (const unsigned short*)B3, pyr, c,
                           1.0f / sqrtf((float)c), G);
        return rpe_check_launch();
    }
    // group order is the map's own row-major order when a row is whole groups (w8 % 8 == 0), and no padding is needed when h8 * w8 is a
    // multiple of the 128-query tile: fmap1 is then the GEMM's A operand as it stands (16-byte aligned rows for the LDS-DMA)
    const bool a_in_place = (w8 % 8 == 0) && G.mp == h8 * w8 && (((uintptr_t)fmap1) & 15) == 0;
    if (!a_in_place) hipLaunchKernelGGL(k_permute_fmap, dim3(ceil_div(G.mp, 256), b * c), dim3(256), 0, s, fmap1, Ap, h8, w8, 0, G.gx, G.npx, G.mp);
    hipLaunchKernelGGL(k_permute_fmap, dim3(ceil_div(G.np, 256), b * c), dim3(256), 0, s, fmap2, Bp, h8, w8, 1, G.gx, G.npx, G.np);
#ifndef CB_SMALL_WG
#define CB_SMALL_WG 4096                          /* band workgroups below which the launch is split per patch (tools/build_variant.sh -DCB_SMALL_WG=0: never): 2 pairs 412 -> 366 us, 6 pairs 1058 -> 1030, 12 pairs 1895 -> 1862, 32 pairs equal (the carry then saves a quarter of the writes) */
#endif
    if (!CB_CARRY || (long long)G.nbands * (G.mp / BM) * b < CB_SMALL_WG) {   // few band workgroups: whole rounds of patch workgroups instead of a nearly empty last round of band walks
        hipLaunchKernelGGL(k_corr_build<false>, dim3(G.nbands * G.npx, G.mp / BM, b), dim3(256), 0, s, a_in_place ? fmap1 : (const float*)Ap, (const float*)Bp, pyr, c,
                           1.0f / sqrtf((float)c), G);
        return rpe_check_launch();
    }
    hipLaunchKernelGGL(k_corr_build<true>, dim3(G.nbands, G.mp / BM, b), dim3(256), 0, s, a_in_place ? fmap1 : (const float*)Ap, (const float*)Bp, pyr, c,
                       1.0f / sqrtf((float)c), G);
    return rpe_check_launch();
}

extern "C" int rpe_corr_build(const float* fmap1, const float* fmap2, int b, int c, int h8, int w8, int levels,
                              void* pyramid, void* stream) {
    return rpe_corr_build_ex(fmap1, fmap2, b, c, h8, w8, levels, RPE_F32, pyramid, stream);
}

extern "C" int rpe_corr_lookup(const void* pyramid, const float* coords, int b, int h8, int w8, int levels, int radius,
                               float* out, void* stream) {
    PyrGeom G;
    if (!pyramid || !coords || !out || radius != RADIUS || !make_geom(b, h8, w8, levels, G)) return RPE_E_BADARG;
    hipLaunchKernelGGL(k_corr_lookup, dim3(ceil_div(G.ngroups, 8 * LK_WAVES), levels, b), dim3(64 * LK_WAVES), 0, (hipStream_t)stream,
                       (const float*)pyramid, coords, out, G);
    return rpe_check_launch();
}

extern "C" size_t rpe_corr_lookup_conv1x1_packed_floats(int cout, int cin) {
    return (cout == FZ_COUT && cin == FZ_CIN) ? (size_t)4 * FZ_STEPS * 64 * 16 : 0;
}

extern "C" int rpe_corr_lookup_conv1x1_pack(const float* weight, int cout, int cin, float* packed, void* stream) {
    if (!weight || !packed) return RPE_E_BADARG;
    if (cout != FZ_COUT || cin != FZ_CIN) return RPE_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_lookup_conv_pack, dim3(ceil_div(4 * FZ_STEPS * 64 * 16, 256)), dim3(256), 0, (hipStream_t)stream, weight, packed);
    return rpe_check_launch();
}

extern "C" int rpe_corr_lookup_conv1x1(const void* pyramid, const float* coords, int b, int h8, int w8, int levels, int radius, const float* packed,
                                       const float* bias, int relu, float* out, long long out_batch_stride, float* out2, long long out2_batch_stride,
                                       void* stream) {
    PyrGeom G;
    if (!pyramid || !coords || !packed || !out || radius != RADIUS || !make_geom(b, h8, w8, levels, G)) return RPE_E_BADARG;
    // four levels (324 channels), rows of whole groups (the workgroup's 64 queries are 64 consecutive pixels), at most 65535 batch items
    if (levels != MAX_LEVELS || (w8 % GQ) != 0 || b > 65535) return RPE_E_UNSUPPORTED;
    FzP P{(const float*)pyramid, coords, packed, bias, out, out_batch_stride, out2, out2_batch_stride, relu};
    const int tiles_per_item = ceil_div(G.ngroups, 8);
    const long long total = (long long)tiles_per_item * b;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (total > cus && total < (1ll << 31)) {
        // more than one round of workgroups: one persistent 8-wave workgroup per CU, lookup of tile n + 1 under the matrix phase of tile n
        if (out2) hipLaunchKernelGGL(k_lookup_conv1x1_pipe<true>, dim3(cus), dim3(512), 0, (hipStream_t)stream, P, G, tiles_per_item, (int)total);
        else hipLaunchKernelGGL(k_lookup_conv1x1_pipe<false>, dim3(cus), dim3(512), 0, (hipStream_t)stream, P, G, tiles_per_item, (int)total);
        return rpe_check_launch();
    }
    const dim3 grid(tiles_per_item, b);
    if (out2) hipLaunchKernelGGL(k_lookup_conv1x1<true>, grid, dim3(256), 0, (hipStream_t)stream, P, G);
    else hipLaunchKernelGGL(k_lookup_conv1x1<false>, grid, dim3(256), 0, (hipStream_t)stream, P, G);
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

extern "C" int rpe_corr_lookup_rounds(const float* coords, int b, int h8, int w8, int levels, int32_t* rounds, int32_t* lines,
                                      void* stream) {
    PyrGeom G;
    if (!coords || !rounds || !lines || !make_geom(b, h8, w8, levels, G)) return RPE_E_BADARG;
    hipLaunchKernelGGL(k_corr_rounds, dim3(ceil_div(G.ngroups, 8 * LK_WAVES), levels, b), dim3(64 * LK_WAVES), 0, (hipStream_t)stream,
                       coords, rounds, lines, G);
    return rpe_check_launch();
}

extern "C" int rpe_corr_export_level(const void* pyramid, int b, int h8, int w8, int levels, int level, float* dense,
                                     void* stream) {
    PyrGeom G;
    if (!pyramid || !dense || level < 0 || level >= levels || !make_geom(b, h8, w8, levels, G)) return RPE_E_BADARG;
    hipLaunchKernelGGL(k_corr_export, dim3(b * h8 * w8), dim3(256), 0, (hipStream_t)stream, (const float*)pyramid, dense, G, level);
    return rpe_check_launch();
}
