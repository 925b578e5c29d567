// Stride-1 "same" convolutions of RAFT's update block as implicit GEMMs on the f32 matrix cores, with the
// element-wise work that follows each of them fused into the epilogue.
//
// Replaces (reference's RAFT submodule, call sites core/pose/pose_net.py:47,65,129):
//   core/RAFT/core/update.py  BasicMotionEncoder.forward  (convc1, convc2, convf2, conv: conv + bias + ReLU + cat)
//                             SepConvGRU.forward          (convz|convr -> sigmoid, r*h ; convq -> tanh, h blend)
//                             FlowHead.conv1              (conv + bias + ReLU)
// The conv library runs these NCHW tensors through NCHW<->NHWC transposes around its GEMM kernel and leaves bias,
// activation, gate arithmetic and concatenation to separate passes; here the convolution reads NCHW directly and
// the accumulator tile goes through bias / context addend / activation / gate blend before its only store.
//
// GEMM view, per batch item:   out[co][p] = sum_{ci,dy,dx} Wt[co][ci][dy][dx] * x[ci][p + dy*W + dx]
//   M = output channels (A operand = packed weights), N = pixels (B operand = the input planes, contiguous in p),
//   K = cin * kh * kw walked as steps of 16 input channels x one tap.
// Workgroup = 4 waves, tile 128(co) x 128(px) or 64(co) x 256(px); each wave owns 64 x 64 = 2 x 2 MFMA 32x32x2 blocks.
// LDS: weights tile [16][BM] per step; input tile [16][BN + 8] per (channel chunk, dy): the kw taps of a row are
// served from the same staged tile by reading it at a column offset (4-B LDS reads have no alignment rule), so the
// input crosses L2->LDS once per dy instead of once per tap.  Both tiles are double buffered; one barrier per step.
// Zero padding: rows (y+dy outside the map) are zero-filled by the loader (whole float4s, W % 4 == 0); columns
// (x+dx outside the row) are masked per lane when the B operand is read.
#include "rpe_common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK 16
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)     /* nothing is scheduled across this point */

struct ConvP {
    const float* x; long long xbs;            // first input channel of the slice; batch stride (floats)
    const float* wp;                          // packed weights [step][16][coP]
    int cin, cout, coP, H, W, hw, kh;         // H, W, hw: OUTPUT map
    int Hin, Win;                             // input map (= H, W unless stride 2)
    const float* bias;                        // [cout] or null
    const float* add; long long abs_;         // (b, cout, hw) pre-activation addend or null
    int mode;
    float* out; long long obs;                // channel 0 of the destination slice; batch stride
    float* out2; long long o2bs;              // second destination (RPE_CONV_RELU/LINEAR: copy; GATE_ZR: r*h)
    const float* h; long long hbs;            // hidden state, channels [0, c)
    const float* z; long long zbs;            // update gate (GATE_H)
    int cgate;
    const float* scale;                       // [cout] or null: v = acc * scale + ...
    const float* res; long long rbs;          // residual added after the activation, then ReLU again (encoder blocks)
    float* stats;                             // [b][cout][tiles_n][2] partial (sum, sum of squares) of v, or null
    const float* pre;                         // [b][cin][2] (mean, 1/std) or null: the input is normalised + ReLU'd as it is staged
};


// Sum over each 32-lane half of the wave with DPP moves (VALU rate, no LDS traffic): quad butterflies, row half-mirror,
// row mirror, then lane 15 of each even 16-lane row is broadcast into the odd row.  Lanes 31 and 63 hold the totals.
__device__ __forceinline__ float half_wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, true));   // row_bcast:15 into rows 1, 3
    return v;
}

// LDS layouts are K-contiguous: As[m][KS], Bs[4 + n][KS] with KS = 20 floats (16 used): a lane fetches the 8 k-values
// it feeds to 8 consecutive MFMAs with two ds_read_b128 (rows 80 B apart: 16 consecutive rows tile the 64 banks
// exactly, so reads, the float4 weight stores and the transposing 4-B input stores are all conflict-free).  The MFMA
// sums over k in any order, so lane half lh of k2-step j supplies k = 8*lh + j for both operands.
// ENC = false: update-block epilogues (bias / addend / ReLU / GRU gates).  ENC = true: encoder epilogues (folded batch
// norm scale+shift, ReLU, residual, partial statistics for instance norm) and output tiles whose upper 32 rows may be
// missing (cout = 96).  Two instantiations keep the update-block kernels free of the encoder's registers and branches.
// T = 32x32 MFMA blocks per wave in each direction: 2 -> 64x64 per wave (tiles 128x128 / 64x256), 1 -> 32x32 per wave
// (tile 64x64, for launches that would not fill the chip with the large tiles: batch-1 tracking).
// VERT: KW then counts VERTICAL taps (a KW x 1 kernel: the GRU's 5x1 convolutions).  The pixel tile is a 16 x 8 patch staged
// with KW/2 halo rows above and below, and tap t reads it 16*t LDS rows further down -- the vertical twin of the dx
// offset, so a 5x1 convolution stages one input tile per channel chunk like a 1x5 one (as a k x 1 kernel walked with
// one flattened 128-pixel tile per dy it staged five, and ran 10 % slower).  Zero rows come from the loader; no masks.
// S2: stride 2 (the encoders' 3x3 pad-1 and 1x1 down-sampling convolutions).  The staged tile holds, per output pixel n
// of the tile, the even input pixel E[n] = in[yi][2x] (LDS rows 0..BN-1) and the odd one O[n] = in[yi][2x+1] (rows
// BN+1+n; row BN = O[-1] of the tile's first pixel), yi = 2y + dy - pad: the loader's aligned float4 (E,O,E,O of two
// neighbouring outputs) is de-interleaved by the 4-byte LDS stores it needs anyway.  Taps: in[2x-1] = O[n-1], in[2x] =
// E[n], in[2x+1] = O[n] -- plain row offsets again.
template <int KW, int WM, bool ENC, int T, bool VERT, bool S2>
__global__ __launch_bounds__(256, 3) void k_conv_igemm(ConvP P) {
    constexpr int WN = 4 / WM, WT = 32 * T, BM = WT * WM, BN = WT * WN, PW = VERT ? 0 : KW / 2;
    constexpr int VTX = 16, VTY = 8, VROWS = VTY + KW - 1;       // VERT patch: 16 x 8 pixels, VROWS staged rows of 16
    constexpr int KS = 20;
    constexpr int NLA = BM / 64;                                 // float4 per thread of the [4 k4][BM] weights tile
    constexpr int NLB = VERT ? (VROWS * VTX / 4 * CK + 255) / 256 : S2 ? BN / 32 : BN / 64;   // float4 per thread of the input tile
    __shared__ __attribute__((aligned(16))) float As[2][BM][KS];
    __shared__ __attribute__((aligned(16))) float Bs[2][VERT ? VROWS * VTX : S2 ? 2 * BN + 8 : BN + 8][KS];
    const int bz = blockIdx.z, m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int vtiles_x = (P.W + VTX - 1) / VTX;                  // VERT: blockIdx.x -> patch (vx0, vy0)
    const int vx0 = VERT ? (blockIdx.x % vtiles_x) * VTX : 0, vy0 = VERT ? (blockIdx.x / vtiles_x) * VTY : 0;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wm = wv / WN, wn = wv % WN;
    const int W = P.W, hw = P.hw, ph = P.kh / 2;
    // encoder epilogues: (scale | 1, bias | 0) of the tile's channel rows, staged now, read after the K loop (whose barriers publish it)
    __shared__ __attribute__((aligned(8))) float sbt[ENC ? BM : 1][2];
    if (ENC && tid < BM) {
        const int co = m0 + tid;
        sbt[tid][0] = (P.scale && co < P.cout) ? P.scale[co] : 1.0f;
        sbt[tid][1] = (P.bias && co < P.cout) ? P.bias[co] : 0.0f;
    }
    const int hw_in = S2 ? P.Hin * P.Win : hw;
    // weights loader: thread -> (m = tid % BM, k4 = tid / BM [+ 256/BM * u]);  packed as [step][k4][coP][4]
    const int a_m = tid % BM, a_k4 = tid / BM;
    constexpr int A_K4STEP = 256 / BM;
    // input loader: lanes 4j..4j+3 fetch 64 contiguous bytes of channel row k = j; thread -> (k = (tid/4) % 16,
    // pixel float4 n4 = tid % 4 + 4 * wave [+ 16 u]).  Halo (threads 0..31): k = tid % 16, side = tid / 16.
    const int b_k = (tid >> 2) & 15, b_n4 = (tid & 3) + 4 * wv;
    const int h_k = tid & 15, h_side = tid >> 4;
    const float* xrow = P.x + (size_t)bz * P.xbs + (size_t)b_k * hw_in;       // channel row b_k of chunk 0
    const float* xrow_h = P.x + (size_t)bz * P.xbs + (size_t)h_k * hw_in;

    f32x16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // x coordinate of this lane's two B columns: decides which taps fall off the row
    const int xq0 = (n0 + wn * WT + l31) % W, xq1 = (n0 + wn * WT + 32 + l31) % W;

    const int nchunk = (P.cin + CK - 1) / CK;
    const int G = VERT ? nchunk : nchunk * P.kh;                 // groups = staged input tiles; each serves KW steps
    float4 ra0, ra1;                                             // (scalars: as an array it is demoted to LDS)
    struct RB { float4 v[NLB]; float4 halo; unsigned ok; float pm, pi, hm, hi; };   // one staged input tile (this thread's part); ok bits:
                                                                 // u (and NLB = halo): lane's float4 is inside the map
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* wnext = P.wp + ((size_t)a_k4 * P.coP + m0 + a_m) * 4;        // this thread's float4 of step 0
    const size_t wstep = (size_t)4 * P.coP * 4, wk4 = (size_t)A_K4STEP * P.coP * 4;
    int lc = 0, ld = 0;                                          // (chunk, dy index) of the next group to load

    auto load_a = [&]() {                                        // next step's weights; advances the pointer
        ra0 = *(const float4*)wnext;
        if (NLA == 2) ra1 = *(const float4*)(wnext + wk4);
        wnext += wstep;
    };
    // next group's input tile; advances (lc, ld).  Issued unconditionally (groups past the end read the dummy address):
    // a load under a branch makes the compiler's s_waitcnt placement assume the worst on every path.
    auto load_b = [&](RB& R) {
        R.ok = 0;
        if (VERT) {                                              // thread -> (k, float4 q4 of the 16-px row, rows rg + 4u)
            const int q4 = tid & 3, rg = tid >> 6;
            const bool cok = lc * CK + b_k < P.cin && vx0 + 4 * q4 + 3 < W;
            const float* src = xrow + (size_t)lc * CK * hw + vx0 + 4 * q4;
#pragma unroll
            for (int u = 0; u < NLB; ++u) {
                const int r = rg + 4 * u, y = vy0 + r - KW / 2;
                const bool ok = cok && r < VROWS && y >= 0 && y < P.H;
                R.v[u] = *(const float4*)(ok ? src + (size_t)y * W : P.x);
                R.ok |= ok ? (1u << u) : 0u;
            }
            ++lc;
            return;
        }
        if (S2) {                                                // thread -> (k, output pixel pair n = 2q, 2q+1)
            const bool cok = lc * CK + b_k < P.cin;
            const float* src = xrow + (size_t)lc * CK * hw_in;
#pragma unroll
            for (int u = 0; u < NLB; ++u) {
                const int na = n0 + 2 * (b_n4 + 16 * u);
                const int y = na / W, x = na - y * W, yi = 2 * y + ld - ph;
                const bool ok = cok && na < hw && yi >= 0 && yi < P.Hin;
                R.v[u] = *(const float4*)(ok ? src + (size_t)yi * P.Win + 2 * x : P.x);
                R.ok |= ok ? (1u << u) : 0u;
            }
            if (KW > 1 && tid < 16) {                            // O[-1]: the input pixel left of the tile's first one
                const int y = n0 / W, x = n0 - y * W, yi = 2 * y + ld - ph;
                const bool ok = lc * CK + h_k < P.cin && x > 0 && yi >= 0 && yi < P.Hin;
                R.halo.x = *(ok ? xrow_h + (size_t)lc * CK * hw_in + (size_t)yi * P.Win + 2 * x - 1 : P.x);
                R.ok |= ok ? (1u << NLB) : 0u;
            }
            if (++ld == P.kh) { ld = 0; ++lc; }
            return;
        }
        const bool cok = lc * CK + b_k < P.cin;
        const int sh = (ld - ph) * W;
        const float* src = xrow + (size_t)lc * CK * hw;
        if (ENC && P.pre) {                                      // instance norm of the INPUT (previous convolution's raw output)
            const float* pp = P.pre + ((size_t)bz * P.cin + (cok ? lc * CK + b_k : 0)) * 2;
            R.pm = pp[0]; R.pi = pp[1];
            if (tid < 32) { const float* ph2 = P.pre + ((size_t)bz * P.cin + (lc * CK + h_k < P.cin ? lc * CK + h_k : 0)) * 2; R.hm = ph2[0]; R.hi = ph2[1]; }
        }
        // out-of-map lanes read a valid dummy address; they are zeroed when the tile is written to LDS (a conditional
        // load is split into four branchy dword loads, and a select right here would wait for the data at once)
#pragma unroll
        for (int u = 0; u < NLB; ++u) {
            const int p = n0 + 4 * (b_n4 + 16 * u) + sh;
            const bool ok = cok && p >= 0 && p + 3 < hw;
            R.v[u] = *(const float4*)(ok ? src + p : P.x);
            R.ok |= ok ? (1u << u) : 0u;
        }
        if (KW > 1 && tid < 32) {
            const int p = n0 + (h_side ? BN : -4) + sh;
            const bool ok = lc * CK + h_k < P.cin && p >= 0 && p + 3 < hw;
            R.halo = *(const float4*)(ok ? xrow_h + (size_t)lc * CK * hw + p : P.x);
            R.ok |= ok ? (1u << NLB) : 0u;
        }
        if (++ld == P.kh) { ld = 0; ++lc; }
    };
    auto store_a = [&](int buf) {
        *(float4*)&As[buf][a_m][4 * a_k4] = ra0;
        if (NLA == 2) *(float4*)&As[buf][a_m][4 * (a_k4 + A_K4STEP)] = ra1;
    };
    auto store_b = [&](const RB& R, int buf) {
        if (VERT) {
            const int q4 = tid & 3, rg = tid >> 6;
#pragma unroll
            for (int u = 0; u < NLB; ++u) {
                const int r = rg + 4 * u;
                if (r >= VROWS) continue;
                const int n = r * VTX + 4 * q4;
                const float4 v = (R.ok >> u) & 1 ? R.v[u] : zero4;
                Bs[buf][n + 0][b_k] = v.x; Bs[buf][n + 1][b_k] = v.y; Bs[buf][n + 2][b_k] = v.z; Bs[buf][n + 3][b_k] = v.w;
            }
            return;
        }
        if (S2) {
#pragma unroll
            for (int u = 0; u < NLB; ++u) {
                const int n = 2 * (b_n4 + 16 * u);
                const float4 v = (R.ok >> u) & 1 ? R.v[u] : zero4;
                Bs[buf][n][b_k] = v.x; Bs[buf][BN + 1 + n][b_k] = v.y; Bs[buf][n + 1][b_k] = v.z; Bs[buf][BN + 2 + n][b_k] = v.w;
            }
            if (KW > 1 && tid < 16) Bs[buf][BN][h_k] = (R.ok >> NLB) & 1 ? R.halo.x : 0.0f;
            return;
        }
        const bool pre = ENC && P.pre != nullptr;
        auto prep = [&](float4 v, float m, float iv) -> float4 {  // relu((x - mean) / std): padding stays zero (applied to valid lanes only)
            v.x = (v.x - m) * iv; v.y = (v.y - m) * iv; v.z = (v.z - m) * iv; v.w = (v.w - m) * iv;
            v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
            return v;
        };
#pragma unroll
        for (int u = 0; u < NLB; ++u) {
            const int n = 4 + 4 * (b_n4 + 16 * u);
            const float4 v = (R.ok >> u) & 1 ? (pre ? prep(R.v[u], R.pm, R.pi) : R.v[u]) : zero4;
            Bs[buf][n + 0][b_k] = v.x; Bs[buf][n + 1][b_k] = v.y; Bs[buf][n + 2][b_k] = v.z; Bs[buf][n + 3][b_k] = v.w;
        }
        if (KW > 1 && tid < 32) {
            const int n = h_side ? 4 + BN : 0;
            const float4 v = (R.ok >> NLB) & 1 ? (pre ? prep(R.halo, R.hm, R.hi) : R.halo) : zero4;
            Bs[buf][n + 0][h_k] = v.x; Bs[buf][n + 1][h_k] = v.y; Bs[buf][n + 2][h_k] = v.z; Bs[buf][n + 3][h_k] = v.w;
        }
    };
    // One step = 16 k-values of one tap = two halves of 16 MFMAs.  Half 1 (k = 8*lh + 0..3) runs on fragments that
    // were read from LDS during the previous step; half 2 (k = 8*lh + 4..7) on fragments read at the top of this
    // step.  The next tile goes to LDS and the workgroup barrier sits BETWEEN the halves, while 16 MFMAs are in
    // flight, and the first-half fragments of the next step are read right after it: the matrix pipe never waits
    // for an LDS round trip or for the barrier.  (Hazards: buffer (s+1)%2 is written before barrier(s); its last
    // readers, halves of step s-1, sit before barrier(s-1).  Half-2 reads of step s precede barrier(s); the buffer
    // they read is next written in step s+1, after barrier(s).)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 a0h1, a1h1, b0h1, b1h1, a0h2, a1h2, b0h2, b1h2;
    const bool hi_rows = T == 1 || m0 + wm * WT + 32 < P.cout;
    // (update-block kernels) false: this wave's output rows lie beyond cout.  Scalar on purpose: as a per-lane condition it
    // puts an exec-mask save/branch around every MFMA group of every step.
    const bool wave_rows = ENC || __builtin_amdgcn_readfirstlane(m0 + wm * WT) < P.cout;                // false: this wave's 64 output rows lie beyond cout (cout = 192 on 128-row tiles)
    auto read_h1 = [&](int bufA, int bufB, int dx) {
        const float* arow = &As[bufA][wm * WT + l31][8 * lh];
        const float* brow = &Bs[bufB][VERT ? wn * WT + l31 + VTX * dx : S2 ? wn * WT + l31 + (dx == 0 ? 0 : dx < 0 ? BN : BN + 1) : 4 + wn * WT + l31 + dx][8 * lh];
        a0h1 = *(const f32x4*)(arow); b0h1 = *(const f32x4*)(brow);
        if (T == 2) { a1h1 = *(const f32x4*)(arow + 32 * KS); b1h1 = *(const f32x4*)(brow + 32 * KS); }
    };
    auto read_h2 = [&](int bufA, int bufB, int dx) {
        const float* arow = &As[bufA][wm * WT + l31][8 * lh + 4];
        const float* brow = &Bs[bufB][VERT ? wn * WT + l31 + VTX * dx : S2 ? wn * WT + l31 + (dx == 0 ? 0 : dx < 0 ? BN : BN + 1) : 4 + wn * WT + l31 + dx][8 * lh + 4];
        a0h2 = *(const f32x4*)(arow); b0h2 = *(const f32x4*)(brow);
        if (T == 2) { a1h2 = *(const f32x4*)(arow + 32 * KS); b1h2 = *(const f32x4*)(brow + 32 * KS); }
    };
    auto mma_half = [&](const f32x4& a0, const f32x4& a1, const f32x4& b0, const f32x4& b1, int dx) {
        const bool v0 = KW == 1 || VERT || (S2 ? (dx >= 0 || xq0 > 0) : (unsigned)(xq0 + dx) < (unsigned)W);
        const bool v1 = KW == 1 || VERT || (S2 ? (dx >= 0 || xq1 > 0) : (unsigned)(xq1 + dx) < (unsigned)W);
        float fb0[4], fb1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { fb0[j] = v0 ? b0[j] : 0.0f; fb1[j] = v1 ? b1[j] : 0.0f; }   // column mask at use, not at the read
        if (!ENC && !wave_rows) return;                          // (wave-uniform; the wave still loads, stores and synchronises)
        if (T == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], fb0[j], acc[0][0], 0, 0, 0);
        } else if (!ENC || hi_rows) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], fb0[j], acc[0][0], 0, 0, 0);
                acc[0][T - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], fb1[j], acc[0][T - 1], 0, 0, 0);
                acc[T - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], fb0[j], acc[T - 1][0], 0, 0, 0);
                acc[T - 1][T - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], fb1[j], acc[T - 1][T - 1], 0, 0, 0);
            }
        } else {                                                 // wave-uniform: rows 32..63 of this tile do not exist
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], fb0[j], acc[0][0], 0, 0, 0);
                acc[0][T - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], fb1[j], acc[0][T - 1], 0, 0, 0);
            }
        }
    };

    // The input planes stream from HBM (each tile is read by only cout/BM * kh workgroups), so their loads are
    // issued a whole group (KW > 1: KW steps) or three steps (KW == 1, three register sets) ahead of use; the
    // weights are shared by every workgroup (L2 hits) and are fetched one step ahead.
    if (KW > 1 || S2) {
        RB R;
        load_a(); load_b(R);
        store_a(0); store_b(R, 0);
        __syncthreads();
        read_h1(0, 0, -PW);
        int step = 0;
        for (int g = 0; g < G; ++g) {
            const int curB = g & 1;
#pragma unroll
            for (int t = 0; t < KW; ++t, ++step) {
                const int curA = step & 1;
                const bool last = (g + 1 == G) && (t + 1 == KW);
                load_a();                                         // weights first: their wait must not cover the
                if (t == 0) load_b(R);                            // input loads issued after them (in-order return)
                read_h2(curA, curB, t - PW);
                SCHED_FENCE();                                    // (keep the order written here: left alone, the
                mma_half(a0h1, a1h1, b0h1, b1h1, t - PW);         //  scheduler sinks the global loads to just before
                SCHED_FENCE();                                    //  the LDS stores and reads all fragments at once)
                if (!last) store_a(curA ^ 1);
                if (t == KW - 1 && g + 1 < G) store_b(R, curB ^ 1);
                __syncthreads();
                if (!last) read_h1(curA ^ 1, t == KW - 1 ? curB ^ 1 : curB, t == KW - 1 ? -PW : t + 1 - PW);
                SCHED_FENCE();
                mma_half(a0h2, a1h2, b0h2, b1h2, t - PW);
                SCHED_FENCE();
            }
        }
    } else {
        RB R0, R1, R2;
        load_b(R0); load_b(R1); load_b(R2);
        load_a();
        store_a(0); store_b(R0, 0);
        __syncthreads();
        read_h1(0, 0, 0);
        auto body = [&](int g, RB& Rload, const RB& Rstore) {   // Rload: free (its tile is in LDS); Rstore: group g+1
            const int cur = g & 1;
            load_a();
            load_b(Rload);
            read_h2(cur, cur, 0);
            SCHED_FENCE();
            mma_half(a0h1, a1h1, b0h1, b1h1, 0);
            SCHED_FENCE();
            if (g + 1 < G) { store_a(cur ^ 1); store_b(Rstore, cur ^ 1); }
            __syncthreads();
            if (g + 1 < G) read_h1(cur ^ 1, cur ^ 1, 0);
            SCHED_FENCE();
            mma_half(a0h2, a1h2, b0h2, b1h2, 0);
            SCHED_FENCE();
        };
        for (int g = 0; g < G; g += 3) {
            body(g, R0, R1);
            if (g + 1 < G) body(g + 1, R1, R2);
            if (g + 2 < G) body(g + 2, R2, R0);
        }
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int mode = P.mode, cg = P.cgate;
    if (!ENC) {
        const float* addb0 = P.add ? P.add + (size_t)bz * P.abs_ : nullptr;
        const float* hb = P.h ? P.h + (size_t)bz * P.hbs : nullptr;
        const float* zb = P.z ? P.z + (size_t)bz * P.zbs : nullptr;
        float* outb0 = P.out + (size_t)bz * P.obs;
        float* out2b0 = P.out2 ? P.out2 + (size_t)bz * P.o2bs : nullptr;
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const int nl = wn * WT + j * 32 + l31;               // column of the tile -> pixel
                const int px = VERT ? (vy0 + nl / VTX) * W + vx0 + nl % VTX : n0 + nl;
                const bool pok = VERT ? (vy0 + nl / VTX < P.H && vx0 + nl % VTX < W) : px < hw;
                const int co0 = m0 + wm * WT + i * 32 + 4 * lh;
#pragma unroll
                for (int rb = 0; rb < 16; rb += 4) {                 // four rows per batch: loads together, then arithmetic, then masked stores
                    size_t e[4]; bool ok[4]; float av[4], bv[4], xv[4], yv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = co0 + r + 8 * (rb >> 2);
                        ok[r] = pok && co < P.cout;
                        e[r] = ok[r] ? (size_t)co * hw + px : 0;
                        bv[r] = P.bias ? P.bias[ok[r] ? co : 0] : 0.0f;
                        av[r] = addb0 ? addb0[e[r]] : 0.0f;
                    }
                    const bool rrows = co0 + 8 * (rb >> 2) >= cg;    // (GATE_ZR; uniform per batch when cg % 4 == 0)
                    if (mode == RPE_CONV_GATE_ZR) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) xv[r] = hb[(rrows && ok[r]) ? e[r] - (size_t)cg * hw : 0];
                    } else if (mode == RPE_CONV_GATE_H) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { xv[r] = zb[e[r]]; yv[r] = hb[e[r]]; }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[i][j][rb + r] + av[r] + bv[r];
                        if (mode == RPE_CONV_GATE_ZR) {
                            const float sg = sigmoid_f(v);
                            if (ok[r]) { if (rrows) out2b0[e[r] - (size_t)cg * hw] = sg * xv[r]; else outb0[e[r]] = sg; }
                        } else if (mode == RPE_CONV_GATE_H) {
                            if (ok[r]) outb0[e[r]] = (1.0f - xv[r]) * yv[r] + xv[r] * tanh_f(v);
                        } else {
                            if (mode == RPE_CONV_RELU) v = v < 0.0f ? 0.0f : v;    // NaN stays NaN, like torch.relu
                            else if (mode == RPE_CONV_TANH) v = tanh_f(v);
                            if (ok[r]) { outb0[e[r]] = v; if (out2b0) out2b0[e[r]] = v; }
                        }
                    }
                }
            }
        return;
    }
    float* red = &As[0][0][0];                                   // [WN * NS][BM][3] partial statistics (sum d, sum d^2, pivot) (LDS is free now:
                                                                 // nothing reads the tiles after the last barrier)
    constexpr int NS = S2 ? T : 1;                               // statistics blocks per wave
    static_assert(WN * NS * 3 <= 2 * KS, "statistics scratch must fit the weights tiles");
    // Four channel rows x T column blocks per batch: the residual loads of a batch are issued back to back from clamped
    // in-range addresses (per-element validity branches made the compiler emit load - wait - store per element), then
    // the arithmetic, then stores under the lane mask.  Compile-time shapes (moments or not | residual / addend / second
    // output or none of them): with every feature behind a run-time test the stride-2 layers' epilogue was ~600 scalar branches
    // and 64 dependent scale / bias loads per wave; (scale, bias) now come from the LDS table staged before the K loop.
    auto epilogue = [&](auto statsc, auto extrac) {
        constexpr bool STATS = decltype(statsc)::value, EXTRA = decltype(extrac)::value;
        const float* resb = EXTRA && P.res ? P.res + (size_t)bz * P.rbs : nullptr;
        const float* addb = EXTRA && P.add ? P.add + (size_t)bz * P.abs_ : nullptr;
        float* outb = P.out + (size_t)bz * P.obs;
        float* out2b = EXTRA && P.out2 ? P.out2 + (size_t)bz * P.o2bs : nullptr;
        const bool relu = mode == RPE_CONV_RELU;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int row0 = wm * WT + i * 32 + 4 * lh;
#pragma unroll
            for (int rb = 0; rb < 16; rb += 4) {
                size_t e[T][4]; bool ok[T][4]; float rv[T][4], av[T][4], sc[4], bi[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + r + 8 * (rb >> 2), co = m0 + row;
                    const bool cok = co < P.cout;
                    const float2 s2 = *(const float2*)&sbt[row][0];
                    sc[r] = s2.x; bi[r] = s2.y;
#pragma unroll
                    for (int j = 0; j < T; ++j) {
                        const int px = n0 + wn * WT + j * 32 + l31;
                        ok[j][r] = cok && px < hw;
                        e[j][r] = ok[j][r] ? (size_t)co * hw + px : 0;
                    }
                }
                if (EXTRA) {
#pragma unroll
                    for (int j = 0; j < T; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            rv[j][r] = resb ? resb[e[j][r]] : 0.0f;
                            av[j][r] = addb ? addb[e[j][r]] : 0.0f;
                        }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float vv[T];
#pragma unroll
                    for (int j = 0; j < T; ++j) {
                        if (EXTRA) { float v = acc[i][j][rb + r] * sc[r]; v += av[j][r]; v += bi[r]; vv[j] = v; }
                        else vv[j] = fmaf(acc[i][j][rb + r], sc[r], bi[r]);
                    }
                    // Instance-norm statistics are taken about a PIVOT (this wave's first column of the channel row), not about
                    // zero: sum(v - p) and sum((v - p)^2) keep their digits when |mean| >> std (a large conv bias is pure shift;
                    // E[v^2] - mean^2 from f32 sums loses mean^2/var * 1e-7 of the variance).
                    // Stride 2 (NS = T): one record per 32-pixel block of the plane, whatever the tile class -- a 64 x 64-tile launch
                    // (small batches) and a 128 x 128-tile one leave the SAME records, so the statistics of an image do not depend on
                    // how many images share the launch (the chunked sequence tracker relies on that).
                    float piv[NS], ssum[NS], ssq[NS];
                    if (STATS) {
#pragma unroll
                        for (int s2 = 0; s2 < NS; ++s2) {
                            const int vb = __builtin_bit_cast(int, vv[S2 ? s2 : 0]);
                            piv[s2] = __builtin_bit_cast(float, lh ? __builtin_amdgcn_readlane(vb, 32) : __builtin_amdgcn_readlane(vb, 0));
                            ssum[s2] = 0.0f; ssq[s2] = 0.0f;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < T; ++j) {
                        float v = vv[j];
                        if (STATS) { const float dv = ok[j][r] ? v - piv[S2 ? j : 0] : 0.0f; ssum[S2 ? j : 0] += dv; ssq[S2 ? j : 0] = fmaf(dv, dv, ssq[S2 ? j : 0]); }
                        if (relu) v = v < 0.0f ? 0.0f : v;
                        if (EXTRA && resb) { v = rv[j][r] + v; v = v < 0.0f ? 0.0f : v; }
                        if (ok[j][r]) { outb[e[j][r]] = v; if (EXTRA && out2b) out2b[e[j][r]] = v; }
                    }
                    if (STATS) {                                 // sum over the 32 lanes that share this channel row
                        const int row = row0 + r + 8 * (rb >> 2);
#pragma unroll
                        for (int s2 = 0; s2 < NS; ++s2) {
                            const float a = half_wave_sum(ssum[s2]), q = half_wave_sum(ssq[s2]);
                            float* rd = red + ((wn * NS + s2) * BM + row) * 3;
                            if (l31 == 31) { rd[0] = a; rd[1] = q; rd[2] = piv[s2]; }
                        }
                    }
                }
            }
        }
    };
    {
        typedef std::integral_constant<bool, true> Yes; typedef std::integral_constant<bool, false> No;
        const bool extra = P.res || P.add || P.out2;
        if (P.stats) { if (extra) epilogue(Yes{}, Yes{}); else epilogue(Yes{}, No{}); }
        else { if (extra) epilogue(No{}, Yes{}); else epilogue(No{}, No{}); }
    }
    if (P.stats) {                                               // combine the WN waves that cover the same channels
        __syncthreads();
        if (S2) {
            const int nrec = (hw + 31) / 32;
            if (tid < BM && m0 + tid < P.cout) {
#pragma unroll
                for (int w2 = 0; w2 < WN; ++w2)
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2) {
                        const int blk = (n0 + w2 * WT + s2 * 32) >> 5;
                        if (blk >= nrec) continue;
                        int nw = hw - blk * 32; nw = nw > 32 ? 32 : nw;                    // valid columns of the block (>= 1)
                        const float* rd = red + ((w2 * NS + s2) * BM + tid) * 3;
                        StatAcc A;
                        A.add_pivoted(nw, rd[0], rd[1], rd[2]);
                        float* st = P.stats + (((size_t)bz * P.cout + m0 + tid) * nrec + blk) * 3;
                        st[0] = (float)A.n; st[1] = (float)A.mean; st[2] = (float)A.m2;
                    }
            }
        } else if (tid < BM && m0 + tid < P.cout) {
            StatAcc A;
#pragma unroll
            for (int w2 = 0; w2 < WN; ++w2) {
                int nw = hw - (n0 + w2 * WT); nw = nw < 0 ? 0 : (nw > WT ? WT : nw);       // valid columns of wave w2
                A.add_pivoted(nw, red[(w2 * BM + tid) * 3], red[(w2 * BM + tid) * 3 + 1], red[(w2 * BM + tid) * 3 + 2]);
            }
            float* st = P.stats + (((size_t)bz * P.cout + m0 + tid) * gridDim.x + blockIdx.x) * 3;
            st[0] = (float)A.n; st[1] = (float)A.mean; st[2] = (float)A.m2;
        }
    }
}

// (mean, 1/sqrt(var + eps)) per (b, c) plane from the per-tile (n, mean, M2) records: what k_conv_igemm's loader-side
// normalisation reads.  Tiles are combined in f64 with the parallel-variance formula, in a fixed order.
// Record layouts: tiles > 0: (b, c, tiles, 3) (k_conv_igemm, k_stem7x7); tiles < 0: (b, |tiles|, c, 3) (k_conv_wino: a workgroup's
// records of all its channels are contiguous -- 12-B records strided by the tile count cost it 13 % as partial-sector writes).
__device__ __forceinline__ void plane_moments(const float* __restrict__ partials, int tiles_signed, int C, int plane, int nthreads, double* sh, double& mean, double& var) {
    const int tiles = tiles_signed < 0 ? -tiles_signed : tiles_signed;
    const int ts = tiles_signed < 0 ? 3 * C : 3;                                      // floats between consecutive tiles of one plane
    const float* pp = tiles_signed < 0 ? partials + ((size_t)(plane / C) * tiles * C + (plane % C)) * 3 : partials + (size_t)plane * tiles * 3;
    double n = 0.0, a = 0.0;
    for (int i = threadIdx.x; i < tiles; i += nthreads) { n += (double)pp[(size_t)ts * i]; a += (double)pp[(size_t)ts * i] * (double)pp[(size_t)ts * i + 1]; }
    n = wave_sum(n); a = wave_sum(a);
    const int nw = nthreads >> 6, wv = threadIdx.x >> 6;
    if (nw > 1) {
        if ((threadIdx.x & 63) == 0) { sh[wv] = n; sh[4 + wv] = a; }
        __syncthreads();
        n = 0.0; a = 0.0;
        for (int w = 0; w < nw; ++w) { n += sh[w]; a += sh[4 + w]; }
        __syncthreads();
    }
    mean = a / n;
    double q = 0.0;
    for (int i = threadIdx.x; i < tiles; i += nthreads) { const double d = (double)pp[(size_t)ts * i + 1] - mean; q += (double)pp[(size_t)ts * i + 2] + (double)pp[(size_t)ts * i] * d * d; }
    q = wave_sum(q);
    if (nw > 1) {
        if ((threadIdx.x & 63) == 0) sh[wv] = q;
        __syncthreads();
        q = 0.0;
        for (int w = 0; w < nw; ++w) q += sh[w];
    }
    var = q / n;
    var = var < 0.0 ? 0.0 : var;
}

// (256 threads: the merge is two passes of dependent-latency loads, 1 280 records per plane behind a 256 x 320 Winograd layer -- with one
// wave per plane the layer-1 launches took 138 us)
__global__ __launch_bounds__(256) void k_instnorm_finalize(const float* __restrict__ partials, int tiles, int C, int hw, float eps, float* __restrict__ mi) {
    const int plane = blockIdx.x;
    __shared__ double sh[8];
    double mean, var;
    plane_moments(partials, tiles, C, plane, 256, sh, mean, var);
    if (threadIdx.x == 0) { mi[(size_t)plane * 2] = (float)mean; mi[(size_t)plane * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps)); }
    (void)hw;
}

// The same for rpe_conv_wino's TILE-MAJOR records (b, tiles, C, 3): a plane's records are 12 bytes every 12 C bytes, so the per-plane
// merge above pulls a whole 64-byte sector per 12-byte record (1 280 records per plane behind a 256 x 320 layer: 130 us per launch).
// Here a workgroup takes FOUR neighbouring channels of one batch item -- 48 contiguous bytes per tile -- as 64 tile slices x 4 channels;
// the slices are combined pairwise in a fixed tree.  Two passes (mean, then the deviations about it) in f64 like plane_moments.
__global__ __launch_bounds__(256) void k_instnorm_finalize_tm(const float* __restrict__ partials, int tiles, int C, float eps, float* __restrict__ mi) {
    const int b = blockIdx.y, cl = threadIdx.x & 3, sl = threadIdx.x >> 2, c = blockIdx.x * 4 + cl;
    __shared__ double sh[2][64][4];
    const bool ok = c < C;
    const float* pp = partials + ((size_t)b * tiles * C + (ok ? c : 0)) * 3;
    const size_t ts = (size_t)3 * C;
    auto tree = [&](double (*v)[4]) {                         // v[0][cl] <- sum over the 64 slices, pairwise, fixed order
#pragma unroll
        for (int h = 32; h >= 1; h >>= 1) {
            __syncthreads();
            if (sl < h) v[sl][cl] += v[sl + h][cl];
        }
        __syncthreads();
    };
    // a thread's records (every 64th tile) are requested at once and kept for the second pass: as two loops of load - accumulate the
    // launch was two chains of ten dependent memory round trips (58 us behind a 256 x 320 layer)
    constexpr int RMAX = 20;                               // (1 280 records per plane behind a 256 x 320 layer = 20 per thread)
    float rn[RMAX], rm[RMAX], rq[RMAX];
    const bool held = tiles <= 64 * RMAX;
    if (held) {
#pragma unroll
        for (int j = 0; j < RMAX; ++j) {
            const int t = sl + 64 * j;
            const float* r = pp + ts * (size_t)(t < tiles ? t : 0);
            const float v0 = r[0], v1 = r[1], v2 = r[2];
            rn[j] = t < tiles ? v0 : 0.0f; rm[j] = v1; rq[j] = v2;
        }
    }
    double n = 0.0, a = 0.0;
    if (held) {
#pragma unroll
        for (int j = 0; j < RMAX; ++j) if (sl + 64 * j < tiles) { const double nt = (double)rn[j]; n += nt; a += nt * (double)rm[j]; }
    } else {
        for (int t = sl; t < tiles; t += 64) { const double nt = (double)pp[ts * t]; n += nt; a += nt * (double)pp[ts * t + 1]; }
    }
    sh[0][sl][cl] = n; sh[1][sl][cl] = a;
    tree(sh[0]); tree(sh[1]);
    n = sh[0][0][cl]; a = sh[1][0][cl];
    const double mean = a / n;
    double q = 0.0;
    if (held) {
#pragma unroll
        for (int j = 0; j < RMAX; ++j) if (sl + 64 * j < tiles) { const double d = (double)rm[j] - mean; q += (double)rq[j] + (double)rn[j] * d * d; }
    } else {
        for (int t = sl; t < tiles; t += 64) { const double d = (double)pp[ts * t + 1] - mean; q += (double)pp[ts * t + 2] + (double)pp[ts * t] * d * d; }
    }
    __syncthreads();
    sh[0][sl][cl] = q;
    tree(sh[0]);
    if (sl == 0 && ok) {
        double var = sh[0][0][cl] / n;
        var = var < 0.0 ? 0.0 : var;
        float* o = mi + ((size_t)b * C + c) * 2;
        o[0] = (float)mean; o[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// Instance norm from the per-tile (n, mean, M2) records k_conv_igemm / k_stem7x7 left behind: one workgroup per (b, c)
// plane combines them in f64 (biased variance), then normalises in ONE read + write pass:
//   y = (x - mean) / sqrt(var + eps); if (relu) y = max(y, 0); if (residual) y = max(residual + y, 0)
// ``res_mi`` (b, C, 2) = (mean, 1/std) of the residual's own instance norm: the residual is then the RAW output of an earlier
// convolution and relu((r - mean) * inv) is applied to it here (the stem's normalised output never exists as a tensor).
__global__ __launch_bounds__(256) void k_instnorm_apply(const float* __restrict__ x, const float* __restrict__ partials, int tiles, int C, int hw,
                                                        float eps, int relu, const float* __restrict__ residual, const float* __restrict__ res_mi,
                                                        float* __restrict__ out, int split) {
    // tiles == 0: ``partials`` = (b, C, 2) (mean, 1/std) pairs of rpe_instnorm_finalize, and ``split`` workgroups share a plane, consecutive
    // workgroups taking consecutive 16-20 KB slices of memory.  (With the records merged here -- 640 twelve-byte records per plane behind
    // a 256 x 320 Winograd layer, one 64-byte sector each, twice -- every workgroup began with ~10 us of dependent loads: 654 us per
    // layer-1 launch against 572 with the pairs given and 530 with them given and 16 workgroups per plane.)
    const int plane = blockIdx.x / split, part = blockIdx.x % split;
    __shared__ double sh[8];
    float mean, inv;
    if (tiles == 0) { mean = partials[(size_t)plane * 2]; inv = partials[(size_t)plane * 2 + 1]; }
    else {
        double dmean, dvar;
        plane_moments(partials, tiles, C, plane, 256, sh, dmean, dvar);
        mean = (float)dmean; inv = (float)(1.0 / sqrt(dvar + (double)eps));
    }
    const float4* xp = (const float4*)(x + (size_t)plane * hw);
    const float4* rp = residual ? (const float4*)(residual + (size_t)plane * hw) : nullptr;
    float4* op = (float4*)(out + (size_t)plane * hw);
    const bool rnorm = res_mi != nullptr;
    const float rmean = rnorm ? res_mi[(size_t)plane * 2] : 0.0f, rinv = rnorm ? res_mi[(size_t)plane * 2 + 1] : 1.0f;
    const bool yrelu = relu & 1, rrelu = !(relu & 2);         // bit 1: the raw residual's own norm has no ReLU (a stride-2 block's shortcut)
    auto fin = [&](float v, float r) -> float {
        float y = (v - mean) * inv;
        if (yrelu) y = y < 0.0f ? 0.0f : y;
        if (rp) {
            if (rnorm) { r = (r - rmean) * rinv; if (rrelu) r = r < 0.0f ? 0.0f : r; }
            y = r + y; y = y < 0.0f ? 0.0f : y;
        }
        return y;
    };
    // four 16-byte loads of each operand in flight per thread, non-temporal on both sides (every byte is touched once: the raw tensor and
    // the residual are 1 GB each at layer 1 of a 48-image pass): 698 -> 655 us there (4.3 -> 4.6 TB/s); unrolling alone 684, eight-fold 668
#ifndef INA_UNROLL
#define INA_UNROLL 4
#endif
#ifndef INA_NT
#define INA_NT 3                                /* bit 0: loads, bit 1: stores */
#endif
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int n4all = hw >> 2;
    const int per = ((n4all + split - 1) / split + 3) & ~3;                   // this workgroup's slice of the plane (whole 64-byte pieces)
    const int n4 = (part + 1) * per < n4all ? (part + 1) * per : n4all;
    int i = part * per + threadIdx.x;
    for (; i + (INA_UNROLL - 1) * (int)blockDim.x < n4; i += INA_UNROLL * blockDim.x) {
        f4 v[INA_UNROLL], r[INA_UNROLL];
#pragma unroll
        for (int u = 0; u < INA_UNROLL; ++u) {
            const int j = i + u * blockDim.x;
            v[u] = INA_NT ? __builtin_nontemporal_load((const f4*)xp + j) : ((const f4*)xp)[j];
            r[u] = rp ? (INA_NT ? __builtin_nontemporal_load((const f4*)rp + j) : ((const f4*)rp)[j]) : (f4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < INA_UNROLL; ++u) {
            const f4 o = {fin(v[u][0], r[u][0]), fin(v[u][1], r[u][1]), fin(v[u][2], r[u][2]), fin(v[u][3], r[u][3])};
            if (INA_NT & 2) __builtin_nontemporal_store(o, (f4*)op + i + u * blockDim.x); else ((f4*)op)[i + u * blockDim.x] = o;
        }
    }
    for (; i < n4; i += blockDim.x) {
        const float4 v = xp[i];
        const float4 r = rp ? rp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        op[i] = make_float4(fin(v.x, r.x), fin(v.y, r.y), fin(v.z, r.z), fin(v.w, r.w));
    }
}

// (cout, cin, kh, kw) -> [step = (chunk*kh + dy)*kw + dx][k4 = 0..3][coP][4]: element (step, k = 4*k4 + e, co) is
// weight[co][chunk*16 + k][dy][dx], zero beyond cin / cout.  One 16-B load per (k4, co) fills an As row segment.
__global__ void k_conv_pack(const float* __restrict__ w, float* __restrict__ wp, int cout, int cin, int kh, int kw, int coP,
                            long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int ke = (int)(e & 3);
    const int co = (int)((e >> 2) % coP);
    const long long rest = (e >> 2) / coP;
    const int k4 = (int)(rest & 3);
    const long long s = rest >> 2;
    const int dx = (int)(s % kw), dy = (int)((s / kw) % kh), c = (int)(s / ((long long)kw * kh));
    const int ci = c * CK + 4 * k4 + ke;
    wp[e] = (co < cout && ci < cin) ? w[(((size_t)co * cin + ci) * kh + dy) * kw + dx] : 0.0f;
}

static inline int conv_cop(int cout) { return (cout + 127) / 128 * 128; }
// Stride-1 launches use 64(co) x 256(px) tiles when cout is 64 / 96 / 192-like (cout % 128 in 1..96), else 128 x 128.
// ONE definition: the launcher, rpe_conv_stats_tiles and the statistics consumers must agree on the pixel-tile width.
static inline bool conv_wide(int cout) { return (cout % 128) != 0 && (cout % 128) <= 96; }
// stride 2: fewer than two rounds of 128 x 128 workgroups on 256 CUs -> 64 x 64 tiles
static inline bool conv_s2_small(int cout, int hw_out, int b) { return (long long)ceil_div(hw_out, 128) * ceil_div(cout, 128) * b < 512; }

extern "C" size_t rpe_conv_packed_floats(int cout, int cin, int kh, int kw) {
    if (cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0) return 0;
    // one zero step of padding: the kernel fetches the weights of step s+1 unconditionally (a load under a branch spoils
    // the compiler's s_waitcnt placement on every path)
    return ((size_t)((cin + CK - 1) / CK) * kh * kw + 1) * CK * conv_cop(cout);
}

extern "C" int rpe_conv_pack(const float* weight, int cout, int cin, int kh, int kw, float* packed, void* stream) {
    if (!weight || !packed || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0) return RPE_E_BADARG;
    const long long total = (long long)rpe_conv_packed_floats(cout, cin, kh, kw);
    hipLaunchKernelGGL(k_conv_pack, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, packed, cout, cin, kh, kw,
                       conv_cop(cout), total);
    return rpe_check_launch();
}

static_assert(sizeof(rpe_conv_desc) == 200, "rpe_conv_desc layout is part of the ABI (ctypes mirror in _lib.py)");
static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

extern "C" int rpe_conv_fused(const rpe_conv_desc* d, void* stream) {
    if (!d || !d->x || !d->packed || !d->out || d->b <= 0 || d->cin <= 0 || d->cout <= 0 || d->h <= 0 || d->w <= 0) return RPE_E_BADARG;
    if (d->kh < 1 || !(d->kh & 1) || (d->kw != 1 && d->kw != 3 && d->kw != 5)) return RPE_E_UNSUPPORTED;
    if ((d->w & 3) || !al16(d->x) || (d->x_batch_stride & 3)) return RPE_E_UNSUPPORTED;     // 16-B input loads
    if (d->mode < RPE_CONV_LINEAR || d->mode > RPE_CONV_TANH) return RPE_E_BADARG;
    if (d->mode == RPE_CONV_TANH && (d->scale || d->residual || d->stats || d->pre_norm || (d->stride != 0 && d->stride != 1) ||
                                     ((d->cout % 64) != 0 && (d->cout % 64) <= 32 && d->kw == 3))) return RPE_E_UNSUPPORTED;   // plain epilogue only
    if (d->mode == RPE_CONV_GATE_ZR && (!d->out2 || !d->hidden || d->gate_channels <= 0 || d->cout != 2 * d->gate_channels)) return RPE_E_BADARG;
    if (d->mode == RPE_CONV_GATE_H && (!d->hidden || !d->zgate)) return RPE_E_BADARG;
    if ((d->mode == RPE_CONV_GATE_ZR || d->mode == RPE_CONV_GATE_H) && (d->scale || d->residual || d->stats)) return RPE_E_BADARG;
    const int stride = d->stride ? d->stride : 1;
    if (stride != 1 && stride != 2) return RPE_E_UNSUPPORTED;
    if (d->pre_norm && (stride != 1 || d->kw != 3 || d->mode > RPE_CONV_RELU)) return RPE_E_UNSUPPORTED;   // loader-side norm: 3x3 stride 1 only
    if (stride == 2 && ((d->h & 1) || (d->w & 1) || d->mode > RPE_CONV_RELU || !((d->kh == 3 && d->kw == 3) || (d->kh == 1 && d->kw == 1))))
        return RPE_E_UNSUPPORTED;                    // stride 2: 3x3 (pad 1) or 1x1 (pad 0) on even maps
    ConvP P;
    P.x = d->x; P.xbs = d->x_batch_stride; P.wp = d->packed;
    P.cin = d->cin; P.cout = d->cout; P.coP = conv_cop(d->cout); P.kh = d->kh;
    P.Hin = d->h; P.Win = d->w; P.H = d->h / stride; P.W = d->w / stride; P.hw = P.H * P.W;
    P.bias = d->bias; P.add = d->add; P.abs_ = d->add_batch_stride; P.mode = d->mode;
    P.out = d->out; P.obs = d->out_batch_stride; P.out2 = d->out2; P.o2bs = d->out2_batch_stride;
    P.h = d->hidden; P.hbs = d->hidden_batch_stride; P.z = d->zgate; P.zbs = d->zgate_batch_stride; P.cgate = d->gate_channels;
    P.scale = d->scale; P.res = d->residual; P.rbs = d->residual_batch_stride; P.stats = d->stats; P.pre = d->pre_norm;
    hipStream_t s = (hipStream_t)stream;
    if (stride == 2) {
        // 128 x 128 tiles, or 64 x 64 for launches that would leave most CUs with at most one workgroup (sequential tracking's 2-3 image
        // batches: layer3 of a 640x512 frame is 40 tiles of 128 pixels).  Statistics are one record per 32 output pixels in BOTH classes
        // (rpe_conv_stats_tiles), bit-identical between them; desc->stats_tiles, when given, must be that count.
        const int t128 = ceil_div(P.hw, 128), t64 = ceil_div(P.hw, 64);
        const bool small = conv_s2_small(d->cout, P.hw, d->b);
        if (d->stats && d->stats_tiles != 0 && d->stats_tiles != ceil_div(P.hw, 32)) return RPE_E_BADARG;
        if (small) {
            dim3 g2(t64, ceil_div(d->cout, 64), d->b);
            if (d->kw == 3) hipLaunchKernelGGL((k_conv_igemm<3, 2, true, 1, false, true>), g2, dim3(256), 0, s, P);
            else hipLaunchKernelGGL((k_conv_igemm<1, 2, true, 1, false, true>), g2, dim3(256), 0, s, P);
            return rpe_check_launch();
        }
        dim3 g2(t128, ceil_div(d->cout, 128), d->b);
        if (d->kw == 3) hipLaunchKernelGGL((k_conv_igemm<3, 2, true, 2, false, true>), g2, dim3(256), 0, s, P);
        else hipLaunchKernelGGL((k_conv_igemm<1, 2, true, 2, false, true>), g2, dim3(256), 0, s, P);
        return rpe_check_launch();
    }
    bool wide = conv_wide(d->cout);                  // 64-row tiles for cout 64 / 96 / 192; 126 runs on one 128-row tile
                                                     // (192 on 128-row tiles with two idle waves was measured: 1.47 vs 1.20 ms)
    int BM = wide ? 64 : 128, BN = wide ? 256 : 128;
    const bool half_tile = (d->cout % 64) != 0 && (d->cout % 64) <= 32 && d->kw == 3 && d->mode <= RPE_CONV_RELU;   // cout = 96
    const bool enc = d->scale || d->residual || d->stats || d->pre_norm || half_tile;
    // Launches that would leave most of the 256 CUs with at most one workgroup use 64x64 tiles instead (4x the
    // workgroups, 6+ of them resident per CU): the batch-1 / batch-2 maps of sequential tracking.
    const bool small = !enc && (long long)ceil_div(P.hw, BN) * ceil_div(d->cout, BM) * d->b < 512;
    if (small) { BM = 64; BN = 64; }
    dim3 grid(ceil_div(P.hw, BN), ceil_div(d->cout, BM), d->b), block(256);
#define LAUNCH(KW_, WM_, ENC_, T_) hipLaunchKernelGGL((k_conv_igemm<KW_, WM_, ENC_, T_, false, false>), grid, block, 0, s, P)
    if (enc) {                                       // encoder epilogues exist for the encoders' 3x3 convolutions only
        if (d->kw != 3 || d->mode > RPE_CONV_RELU) return RPE_E_UNSUPPORTED;
        if (wide) LAUNCH(3, 1, true, 2); else LAUNCH(3, 2, true, 2);
    } else if (small) { if (d->kw == 1) LAUNCH(1, 2, false, 1); else if (d->kw == 3) LAUNCH(3, 2, false, 1); else LAUNCH(5, 2, false, 1); }
    else if (wide) { if (d->kw == 1) LAUNCH(1, 1, false, 2); else if (d->kw == 3) LAUNCH(3, 1, false, 2); else LAUNCH(5, 1, false, 2); }
    else if (d->kw == 1 && d->kh == 5) {             // 5x1 (GRU, vertical half): 16 x 8 pixel patches, taps along y
        dim3 vgrid(ceil_div(d->w, 16) * ceil_div(d->h, 8), grid.y, grid.z);
        hipLaunchKernelGGL((k_conv_igemm<5, 2, false, 2, true, false>), vgrid, block, 0, s, P);
    } else         { if (d->kw == 1) LAUNCH(1, 2, false, 2); else if (d->kw == 3) LAUNCH(3, 2, false, 2); else LAUNCH(5, 2, false, 2); }
#undef LAUNCH
    return rpe_check_launch();
}

extern "C" int rpe_conv_stats_tiles(int cout, int h, int w, int stride) {
    if (cout <= 0 || h <= 0 || w <= 0 || (stride != 1 && stride != 2)) return 0;
    if (stride == 2) return ceil_div((int64_t)(h / 2) * (w / 2), 32);  // one record per 32 output pixels, in either tile class
    return ceil_div((int64_t)h * w, conv_wide(cout) ? 256 : 128);     // (a stride-1 statistics launch is never a "small" 64x64 one)
}

// Kept for callers of the round-3 ABI: the record count no longer depends on the batch.
extern "C" int rpe_conv_stats_tiles_batch(int cout, int h, int w, int stride, int b) {
    if (b <= 0) return 0;
    return rpe_conv_stats_tiles(cout, h, w, stride);
}

extern "C" int rpe_instnorm_apply_ex(const float* x, const float* partials, int tiles, int b, int c, int hw, float eps, int relu,
                                     const float* residual, const float* residual_mean_inv, float* out, void* stream) {
    if (!x || !partials || !out || b <= 0 || c <= 0 || hw <= 0 || (residual_mean_inv && !residual)) return RPE_E_BADARG;
    if ((hw & 3) || !al16(x) || !al16(out) || (residual && !al16(residual))) return RPE_E_UNSUPPORTED;
    int split = 1;                                   // tiles == 0 (moments given): slices of >= 16 KB, at most 16 per plane
#ifndef INA_MAXSPLIT
#define INA_MAXSPLIT 16
#endif
#ifndef INA_MINBYTES
#define INA_MINBYTES 16384
#endif
    if (tiles == 0) while (split < INA_MAXSPLIT && (long long)hw * 4 / (split * 2) >= INA_MINBYTES) split *= 2;
    hipLaunchKernelGGL(k_instnorm_apply, dim3(b * c * split), dim3(256), 0, (hipStream_t)stream, x, partials, tiles, c, hw, eps, relu, residual,
                       residual_mean_inv, out, split);
    return rpe_check_launch();
}

extern "C" int rpe_instnorm_apply(const float* x, const float* partials, int tiles, int b, int c, int hw, float eps, int relu,
                                  const float* residual, float* out, void* stream) {
    return rpe_instnorm_apply_ex(x, partials, tiles, b, c, hw, eps, relu, residual, nullptr, out, stream);
}

extern "C" int rpe_instnorm_finalize(const float* partials, int tiles, int b, int c, int hw, float eps, float* mean_inv, void* stream) {
    if (!partials || !mean_inv || tiles == 0 || b <= 0 || c <= 0 || hw <= 0) return RPE_E_BADARG;
    if (tiles < 0) hipLaunchKernelGGL(k_instnorm_finalize_tm, dim3(ceil_div(c, 4), b), dim3(256), 0, (hipStream_t)stream, partials, -tiles, c, eps, mean_inv);
    else hipLaunchKernelGGL(k_instnorm_finalize, dim3(b * c), dim3(256), 0, (hipStream_t)stream, partials, tiles, c, hw, eps, mean_inv);
    return rpe_check_launch();
}
