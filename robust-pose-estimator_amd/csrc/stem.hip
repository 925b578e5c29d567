// The encoders' first layer (core/RAFT/core/extractor.py BasicEncoder.conv1 + norm1 + relu1; input normalisation of
// core/RAFT/core/raft.py RAFT.forward: image = 2 * (image / 255) - 1): a 7x7 stride-2 pad-3 convolution of the three image
// channels to 64, as an implicit GEMM on the f32 matrix cores that reads the RAW 0..255 image.
//
// K = 3*7*7 = 147 has no 16-channel chunks to stage, so the operand is built the other way round: a workgroup stages the
// normalised input PATCH of its 32 x 8 output pixels once (3 x 21 x 69 floats, zero outside the image = the reference's
// zero padding of the normalised image) and every MFMA B-operand element is a 4-byte LDS read of
// patch[ci][2*ty + dy][2*tx + dx]; the k -> (ci, dy, dx) offsets come from a 160-entry table.  Weights ([64][164], k
// contiguous, 42 KB) sit in LDS for the whole workgroup; within a 16-tap block lane half lh of k2-step j takes k = 8*lh + j,
// so a lane's eight weights and eight tap offsets are two ds_read_b128 each.  Epilogue as the residual blocks': bias, folded batch norm (cnet) or per-tile
// partial sums for rpe_instnorm_apply (fnet), ReLU.
#include "rpe_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SPX 32
#define SPY 8
// The same kernel serves the motion encoder's convf1 (core/RAFT/core/update.py: 7x7 stride 1 on the 2 flow channels -> 128,
// bias + ReLU): template <CIN, STRIDE>, 64 output channels per workgroup (blockIdx.y picks the channel tile).
template <int CIN, int STRIDE> struct StemGeo {
    static constexpr int TAPS = CIN * 49;
    static constexpr int SK = (TAPS + 15) / 16 * 16;            // taps padded to 16-tap blocks (zero weights beyond): 160 / 112
    static constexpr int SKA = SK + 4;                          // weight row stride 164 / 116: 16 consecutive rows tile the 64 banks
    static constexpr int PROWS = STRIDE * SPY + 7 - STRIDE;     // 21 / 14
    static constexpr int PCOLS = STRIDE * SPX + 7 - STRIDE;     // 69 / 38
    static constexpr int PSTR = (PCOLS + 7) / 8 * 8;            // 72 / 40
};

struct StemP {
    const float* x; int H, W, Ho, Wo, cout;     // (b,CIN,H,W) input; output map; total output channels
    float div, mul, sub;                        // xn = mul * (x / div) - sub
    const float* wk;                            // [64][SKA]
    const float* bias; const float* scale;      // v = acc * scale[co] + bias[co]   (scale may be null)
    int relu, snp;                              // snp: patches per workgroup (weights are staged once for all of them)
    float* out; float* stats;                   // (b,cout,Ho,Wo); [b][cout][tiles][3] (n, mean, M2) or null
};

__device__ __forceinline__ float half_wave_sum_s(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, true));
    return v;
}

// Epilogue of one patch for one wave.  sb[co][2] = (scale | 1, bias | 0) of the workgroup's channels; element offsets within the
// (batch item, channel tile) block are 32-bit (the launcher checks the size).
template <int COB, bool STATS, bool INSIDE>
__device__ __forceinline__ void stem_epilogue(f32x16 (&acc)[COB][2], const StemP& P, float* __restrict__ ob, const float* sb, float* red,
                                              int x0, int y0, int wv, int l31, int lh) {
    const unsigned hw = (unsigned)(P.Ho * P.Wo);
    const int ya = y0 + 2 * wv, x = x0 + l31;
    const bool ok0 = INSIDE || ((ya < P.Ho) & (x < P.Wo)), ok1 = INSIDE || ((ya + 1 < P.Ho) & (x < P.Wo));
    const unsigned lane_off = (unsigned)(ya * P.Wo + x) + (unsigned)(4 * lh) * hw, wo = (unsigned)P.Wo;
    const bool relu = P.relu != 0;
    const float* sbl = sb + 8 * lh;
#pragma unroll
    for (int i = 0; i < COB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cc = i * 32 + (r & 3) + 8 * (r >> 2);                // + 4 * lh: this lane's channel of the tile
            const float2 s2 = *(const float2*)(sbl + 2 * cc);
            float v0 = fmaf(acc[i][0][r], s2.x, s2.y), v1 = fmaf(acc[i][1][r], s2.x, s2.y);
            if (STATS) {                                                      // sums about a pivot (the wave's first output of this channel), see k_conv_igemm
                const float piv = __builtin_bit_cast(float, lh ? __builtin_amdgcn_readlane(__builtin_bit_cast(int, v0), 32)
                                                               : __builtin_amdgcn_readlane(__builtin_bit_cast(int, v0), 0));
                const float d0 = ok0 ? v0 - piv : 0.0f, d1 = ok1 ? v1 - piv : 0.0f;
                const float ssum = half_wave_sum_s(d0 + d1), ssq = half_wave_sum_s(fmaf(d1, d1, d0 * d0));
                if (l31 == 31) { float* rd = red + ((wv * 64) + cc + 4 * lh) * 3; rd[0] = ssum; rd[1] = ssq; rd[2] = piv; }
            }
            if (relu) { v0 = v0 < 0.0f ? 0.0f : v0; v1 = v1 < 0.0f ? 0.0f : v1; }
            const unsigned e = lane_off + (unsigned)cc * hw;
            if (ok0) ob[e] = v0;
            if (ok1) ob[e + wo] = v1;
        }
}

// COB = 32-channel blocks per workgroup: 2 -> 64 output channels; 1 -> 32, for launches of a few dozen workgroups (the motion encoder's
// convf1 of one frame pair: 80 workgroups of 64 channels on 256 CUs, each a chain of weight staging -> patch staging -> 224 matrix
// instructions per wave): twice the workgroups, half the weights to stage and half the matrix instructions each.  Same products in
// the same order: bit-identical outputs.
template <int CIN, int STRIDE, int COB>
__global__ __launch_bounds__(256, 2) void k_stem7x7(StemP P) {
    typedef StemGeo<CIN, STRIDE> G;
    constexpr int TCO = 32 * COB;
    constexpr int SK = G::SK, SKA = G::SKA, PROWS = G::PROWS, PCOLS = G::PCOLS, PSTR = G::PSTR, TAPS = G::TAPS;
    const int SNP = P.snp;
    __shared__ __attribute__((aligned(16))) float As[TCO][SKA];
    __shared__ float patch[CIN][PROWS][PSTR];
    __shared__ __attribute__((aligned(16))) int koff[SK];
    __shared__ float red[4][64][3];                          // per wave and channel: sum(v - p), sum((v - p)^2), pivot p
    __shared__ __attribute__((aligned(8))) float sb[TCO][2];  // (scale | 1, bias | 0) per channel of the tile
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int bz = blockIdx.z, cbase = blockIdx.y * TCO;
    const int tiles_x = (P.Wo + SPX - 1) / SPX;
    const int ntiles = tiles_x * ((P.Ho + SPY - 1) / SPY);
    const size_t hw_in = (size_t)P.H * P.W;
    const float* xb = P.x + (size_t)bz * CIN * hw_in;
    // ---- stage weights and tap table once; then SNP patches one after the other
    for (int i = tid; i < TCO * SKA / 4; i += 256) ((float4*)&As[0][0])[i] = ((const float4*)(P.wk + (size_t)cbase * SKA))[i];
    if (tid < TCO) { sb[tid][0] = P.scale ? P.scale[cbase + tid] : 1.0f; sb[tid][1] = P.bias ? P.bias[cbase + tid] : 0.0f; }
    for (int k = tid; k < SK; k += 256) {
        const int kk = k < TAPS ? k : 0, ci = kk / 49, dy = (kk % 49) / 7, dx = kk % 7;
        koff[k] = (ci * PROWS + dy) * PSTR + dx;
    }
    for (int pi = 0; pi < SNP; ++pi) {
    const int tile = blockIdx.x * SNP + pi;
    if (tile >= ntiles) break;                               // (uniform)
    const int x0 = (tile % tiles_x) * SPX, y0 = (tile / tiles_x) * SPY;
    if (pi) __syncthreads();                                 // every wave is done reading the previous patch
    {   // all loads of the patch first (clamped addresses, no branches), then normalise + store: as a load-use-store loop
        // the 17 round trips per thread serialise and cost twice the MFMA time of the workgroup
        constexpr int NI = (CIN * PROWS * PCOLS + 255) / 256;
        float raw[NI]; unsigned okm = 0;
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = tid + 256 * u;
            const int ci = i / (PROWS * PCOLS), rem = i - ci * (PROWS * PCOLS), r = rem / PCOLS, c = rem - r * PCOLS;
            const int yi = STRIDE * y0 - 3 + r, xi = STRIDE * x0 - 3 + c;
            const bool ok = (i < CIN * PROWS * PCOLS) & (yi >= 0) & (yi < P.H) & (xi >= 0) & (xi < P.W);
            raw[u] = xb[ok ? ci * hw_in + (size_t)yi * P.W + xi : 0];
            okm |= ok ? (1u << u) : 0u;
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = tid + 256 * u;
            if (i >= CIN * PROWS * PCOLS) break;
            const int ci = i / (PROWS * PCOLS), rem = i - ci * (PROWS * PCOLS), r = rem / PCOLS, c = rem - r * PCOLS;
            const float v = rn_sub(rn_mul(P.mul, rn_div(raw[u], P.div)), P.sub);          // 2 * (x / 255) - 1, as torch rounds it
            patch[ci][r][c] = (okm >> u) & 1 ? v : 0.0f;
        }
    }
    __syncthreads();
    // ---- 64 (co) x 256 (px) per workgroup; wave wv owns output rows 2*wv, 2*wv+1 of the patch (32 px each)
    f32x16 acc[COB][2];
#pragma unroll
    for (int i = 0; i < COB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const float* pbase = &patch[0][0][0];
    const int lane0 = (STRIDE * (2 * wv)) * PSTR + STRIDE * l31, lane1 = (STRIDE * (2 * wv + 1)) * PSTR + STRIDE * l31;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    for (int blk = 0; blk < SK / 16; ++blk) {
        const int kb = 16 * blk + 8 * lh;
        f32x4 a0[2], a1[2]; i32x4 ko[2];
        a0[0] = *(const f32x4*)&As[l31][kb];      a0[1] = *(const f32x4*)&As[l31][kb + 4];
        if (COB == 2) { a1[0] = *(const f32x4*)&As[TCO - 32 + l31][kb]; a1[1] = *(const f32x4*)&As[TCO - 32 + l31][kb + 4]; }
        else { a1[0] = a0[0]; a1[1] = a0[1]; }
        ko[0] = *(const i32x4*)&koff[kb];         ko[1] = *(const i32x4*)&koff[kb + 4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int o = ko[j >> 2][j & 3];
            const float b0 = pbase[o + lane0], b1 = pbase[o + lane1];
            const float fa0 = a0[j >> 2][j & 3], fa1 = a1[j >> 2][j & 3];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, b1, acc[0][1], 0, 0, 0);
            if (COB == 2) {
                acc[COB - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, b0, acc[COB - 1][0], 0, 0, 0);
                acc[COB - 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, b1, acc[COB - 1][1], 0, 0, 0);
            }
        }
    }
    // ---- epilogue (C/D layout: col = lane&31 = tx, row = (r&3) + 8*(r>>2) + 4*lh): v = acc * scale + bias from the LDS table,
    // moments about a pivot, ReLU, store.  Compile-time shapes (moments or not | patch inside the map or on its border): as one
    // loop with every feature behind a run-time test it was 64 dependent scalar-bias loads, 500 branches and 1 660 vector
    // instructions per wave and patch -- as long as the patch's 320 matrix instructions.
    {
        const size_t hw = (size_t)P.Ho * P.Wo;
        float* ob = P.out + ((size_t)bz * P.cout + cbase) * hw;
        const bool inside = (y0 + SPY <= P.Ho) & (x0 + SPX <= P.Wo);
        if (P.stats) { if (inside) stem_epilogue<COB, true, true>(acc, P, ob, &sb[0][0], &red[0][0][0], x0, y0, wv, l31, lh);
                       else stem_epilogue<COB, true, false>(acc, P, ob, &sb[0][0], &red[0][0][0], x0, y0, wv, l31, lh); }
        else { if (inside) stem_epilogue<COB, false, true>(acc, P, ob, &sb[0][0], &red[0][0][0], x0, y0, wv, l31, lh);
               else stem_epilogue<COB, false, false>(acc, P, ob, &sb[0][0], &red[0][0][0], x0, y0, wv, l31, lh); }
    }
    if (P.stats) {
        __syncthreads();
        if (tid < TCO) {
            StatAcc A;
            int ncols = P.Wo - x0; ncols = ncols > SPX ? SPX : ncols;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                int nrows = P.Ho - (y0 + 2 * w2); nrows = nrows < 0 ? 0 : (nrows > 2 ? 2 : nrows);
                A.add_pivoted(ncols * nrows, red[w2][tid][0], red[w2][tid][1], red[w2][tid][2]);
            }
            float* st = P.stats + (((size_t)bz * P.cout + cbase + tid) * ntiles + tile) * 3;
            st[0] = (float)A.n; st[1] = (float)A.mean; st[2] = (float)A.m2;
        }
    }
    }
}

// weight (cout, cin, 7, 7) -> [co][k = (ci*7 + dy)*7 + dx], rows padded to ska floats, zero beyond the last tap
__global__ void k_stem_pack(const float* __restrict__ w, float* __restrict__ wk, int cout, int taps, int ska) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cout * ska) return;
    const int co = e / ska, k = e % ska;
    wk[e] = k < taps ? w[co * taps + k] : 0.0f;
}

static int stem_ska(int cin) { return (cin * 49 + 15) / 16 * 16 + 4; }
static bool stem_ok(int cin, int stride, int cout) { return ((cin == 3 && stride == 2) || (cin == 2 && stride == 1)) && cout > 0 && cout % 64 == 0; }

extern "C" int rpe_stem_tiles(int h, int w, int stride) {
    if (h <= 0 || w <= 0 || (stride != 1 && stride != 2) || (stride == 2 && ((h & 1) || (w & 1)))) return 0;
    return ceil_div(w / stride, SPX) * ceil_div(h / stride, SPY);
}

extern "C" size_t rpe_stem_packed_floats(int cout, int cin) { return (cout > 0 && (cin == 2 || cin == 3)) ? (size_t)cout * stem_ska(cin) : 0; }

extern "C" int rpe_stem_pack(const float* weight, int cout, int cin, float* packed, void* stream) {
    if (!weight || !packed || cout <= 0 || (cin != 2 && cin != 3)) return RPE_E_BADARG;
    hipLaunchKernelGGL(k_stem_pack, dim3(ceil_div((int64_t)cout * stem_ska(cin), 256)), dim3(256), 0, (hipStream_t)stream, weight, packed, cout,
                       cin * 49, stem_ska(cin));
    return rpe_check_launch();
}

extern "C" int rpe_stem_conv(const float* image, int b, int cin, int h, int w, int stride, float div, float mul, float sub, const float* packed,
                             int cout, const float* bias, const float* scale, int relu, float* out, float* stats, void* stream) {
    if (!image || !packed || !out || b <= 0 || h <= 0 || w <= 0) return RPE_E_BADARG;
    if (!stem_ok(cin, stride, cout) || (stride == 2 && ((h & 1) || (w & 1))) || (((uintptr_t)packed) & 15)) return RPE_E_UNSUPPORTED;
    if ((long long)(h / stride) * (w / stride) * 65 * 4 >= (1ll << 32)) return RPE_E_UNSUPPORTED;      // 32-bit offsets within a 64-channel output block
    StemP P;
    P.x = image; P.H = h; P.W = w; P.Ho = h / stride; P.Wo = w / stride; P.cout = cout; P.div = div; P.mul = mul; P.sub = sub; P.wk = packed;
    P.bias = bias; P.scale = scale; P.relu = relu; P.out = out; P.stats = stats;
    // several patches per workgroup share one staging of the weights -- when the launch still fills the chip that way
    // (bench: 15 360 stem patches; one frame of sequential tracking: 640)
    const long long patches = (long long)rpe_stem_tiles(h, w, stride) * (cout / 64) * b;
    P.snp = patches >= 5 * 2048 ? 5 : patches >= 2 * 2048 ? 2 : 1;
#ifndef STEM_SMALL_PATCHES
#define STEM_SMALL_PATCHES 256
#endif
    if (patches < STEM_SMALL_PATCHES) {            // fewer workgroups than CUs: 32-channel workgroups (same arithmetic, bit-identical)
        dim3 g1(rpe_stem_tiles(h, w, stride), cout / 32, b);
        if (cin == 3) hipLaunchKernelGGL((k_stem7x7<3, 2, 1>), g1, dim3(256), 0, (hipStream_t)stream, P);
        else hipLaunchKernelGGL((k_stem7x7<2, 1, 1>), g1, dim3(256), 0, (hipStream_t)stream, P);
        return rpe_check_launch();
    }
    dim3 grid(ceil_div(rpe_stem_tiles(h, w, stride), P.snp), cout / 64, b);
    if (cin == 3) hipLaunchKernelGGL((k_stem7x7<3, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL((k_stem7x7<2, 1, 2>), grid, dim3(256), 0, (hipStream_t)stream, P);
    return rpe_check_launch();
}
