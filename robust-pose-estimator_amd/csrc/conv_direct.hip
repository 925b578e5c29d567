// Generic convolution for the map sizes the tuned kernels refuse (odd maps, rows that are not whole 16-byte quads): any kernel up to
// 7 x 7, stride 1 or 2, any zero padding, any map size, NCHW channel slices in and out -- as an implicit GEMM on the f32 matrix cores
// with element-wise (4-byte, bounds-checked) operand gathers.
//
// Replaces the torch.nn.functional.conv2d calls (MIOpen) the host code used as its fallback route for such shapes: every convolution
// of the reference's RAFT (core/RAFT/core/extractor.py, update.py; call sites core/pose/pose_net.py:47,65,129) now runs in this
// library whatever the image size (the reference accepts any img_size, configuration/infer_f2f.yaml:13).  It is the ROBUST route, not
// the fast one: the tuned kernels (conv.hip, conv_wino*.hip, conv1x1.hip, stem.hip) serve every size whose 1/8 map has an even
// height and a width that is a multiple of 4 -- all of the reference's configurations -- and are 2-3x faster.
//
// GEMM view per batch item: out[co][p] = sum_k W[co][k] * X[k][p], k = (ci, ky, kx) flattened, p = output pixel.  Workgroup = 4 waves =
// 64 output channels x 64 pixels, wave = one 32 x 32 block of v_mfma_f32_32x32x2_f32, K in steps of 16 through LDS (k-major tiles: lane
// l of a matrix instruction reads row k0 + l / 32, column l % 32 -- consecutive lanes, consecutive floats); the next step's operands
// are gathered into registers while the current step's products run.  Sums over k in increasing order, two k per instruction.
#include "rpe_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CD_M 64
#define CD_N 64
#define CD_K 16
#define CD_LD (CD_M + 4)          // LDS row pitch (floats): the transposing weight stores of four k rows land on different banks

struct CDP {
    const float* x; long long xbs;
    const float* w; const float* bias;
    float* out; long long obs;
    int cin, cout, H, W, Ho, Wo, kh, kw, stride, ph, pw, relu, K;
};

__global__ __launch_bounds__(256) void k_conv_direct(CDP P) {
    __shared__ float As[2][CD_K][CD_LD];                      // [k][co]
    __shared__ float Bs[2][CD_K][CD_LD];                      // [k][pixel]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.y * CD_M, n0 = blockIdx.x * CD_N, bz = blockIdx.z;
    const int hwo = P.Ho * P.Wo, hwi = P.H * P.W, khw = P.kh * P.kw;
    const float* xb = P.x + (size_t)bz * P.xbs;
    // weights role: thread -> (co = tid / 4, four consecutive k); input role: thread -> (pixel = tid % 64, k = tid / 64 + 4 j)
    const int a_m = tid >> 2, a_k = (tid & 3) * 4;
    const int a_co = m0 + a_m;
    const float* wrow = P.w + (size_t)(a_co < P.cout ? a_co : 0) * P.K;
    const int b_n = tid & 63, b_k = tid >> 6;
    const int px = n0 + b_n;
    const bool pok = px < hwo;
    const int oy = pok ? px / P.Wo : 0, ox = pok ? px - oy * P.Wo : 0;
    const int iy0 = oy * P.stride - P.ph, ix0 = ox * P.stride - P.pw;

    float ra[4], rb[4];
    auto gather = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + a_k + j;
            ra[j] = (a_co < P.cout && k < P.K) ? wrow[k] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + b_k + 4 * j;
            float v = 0.0f;
            if (pok && k < P.K) {
                const int ci = k / khw, t = k - ci * khw, ky = t / P.kw, kx = t - ky * P.kw;
                const int iy = iy0 + ky, ix = ix0 + kx;
                if (iy >= 0 && iy < P.H && ix >= 0 && ix < P.W) v = xb[(size_t)ci * hwi + (size_t)iy * P.W + ix];
            }
            rb[j] = v;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { As[buf][a_k + j][a_m] = ra[j]; Bs[buf][b_k + 4 * j][b_n] = rb[j]; }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int nsteps = (P.K + CD_K - 1) / CD_K;
    gather(0);
    stash(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gather((s + 1) * CD_K);           // in flight beside this step's matrix instructions
#pragma unroll
        for (int kk = 0; kk < CD_K; kk += 2) {
            const float a = As[buf][kk + lh][wm * 32 + l31], b = Bs[buf][kk + lh][wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (s + 1 < nsteps) stash(buf ^ 1);                   // (the other buffer: last read before the previous barrier)
        __syncthreads();
    }
    // C/D layout of the 32 x 32 block: column = lane % 32, row = (r % 4) + 8 (r / 4) + 4 (lane / 32)
    const int opx = n0 + wn * 32 + l31;
    if (opx >= hwo) return;
    float* ob = P.out + (size_t)bz * P.obs;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < P.cout) {
            float v = acc[r] + (P.bias ? P.bias[co] : 0.0f);
            if (P.relu) v = v < 0.0f ? 0.0f : v;              // NaN stays NaN, like torch.relu
            ob[(size_t)co * hwo + opx] = v;
        }
    }
}

extern "C" int rpe_conv_direct(const float* x, long long x_batch_stride, const float* weight, const float* bias, int b, int cin, int cout,
                               int h, int w, int kh, int kw, int stride, int pad_h, int pad_w, int relu, float* out,
                               long long out_batch_stride, void* stream) {
    if (!x || !weight || !out || b <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || kh <= 0 || kw <= 0 || pad_h < 0 || pad_w < 0) return RPE_E_BADARG;
    if (kh > 7 || kw > 7 || (stride != 1 && stride != 2)) return RPE_E_UNSUPPORTED;
    const int ho = (h + 2 * pad_h - kh) / stride + 1, wo = (w + 2 * pad_w - kw) / stride + 1;      // torch.nn.functional.conv2d's output size
    if (h + 2 * pad_h < kh || w + 2 * pad_w < kw || ho <= 0 || wo <= 0) return RPE_E_BADARG;
    if ((long long)cin * kh * kw >= (1ll << 31) || (long long)ho * wo >= (1ll << 31) || b > 65535) return RPE_E_UNSUPPORTED;
    CDP P;
    P.x = x; P.xbs = x_batch_stride; P.w = weight; P.bias = bias; P.out = out; P.obs = out_batch_stride;
    P.cin = cin; P.cout = cout; P.H = h; P.W = w; P.Ho = ho; P.Wo = wo; P.kh = kh; P.kw = kw; P.stride = stride; P.ph = pad_h; P.pw = pad_w;
    P.relu = relu; P.K = cin * kh * kw;
    const dim3 grid(ceil_div((int64_t)ho * wo, CD_N), ceil_div(cout, CD_M), b);
    hipLaunchKernelGGL(k_conv_direct, grid, dim3(256), 0, (hipStream_t)stream, P);
    return rpe_check_launch();
}
