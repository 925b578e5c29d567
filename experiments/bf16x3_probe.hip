// EXPERIMENT (not part of librpe_hip.so, never in bench.py's headline): is a 3-way bf16 split of f32 operands -- 6 bf16 products per
// f32 product on v_mfma_f32_32x32x16_bf16, f32 accumulation -- a way past the f32 matrix pipe for this path's GEMM-shaped layers?
//
//   x = hi + mid + lo exactly (three bf16 of 8 significand bits each, truncation split), x*y ~ hi*hi + hi*mid + mid*hi + mid*mid +
//   hi*lo + lo*hi  (dropped: mid*lo, lo*mid, lo*lo <= 2^-23 |xy|)
//
// Measures, on a convc1-sized GEMM (M = 256 output channels, K = 320, N = 163 840 pixels: 26.8 GFLOP):
//   (a) time of the f32 loop (v_mfma_f32_32x32x2_f32, operands k-major in LDS) and of the 6-product bf16 loop with the SAME tiling
//       (128 x 128 per workgroup, 64 x 64 per wave), both with operands pre-split / pre-laid-out in global memory in the layout each
//       instruction wants (the favourable case: a real layer would have to split its activations on the fly);
//   (b) error against an f64 evaluation: f32 matrix pipe vs the 6-product split vs the 3-product split (hi*hi + hi*mid + mid*hi).
// Build + run:  hipcc -O3 --offload-arch=gfx950 experiments/bf16x3_probe.hip -o experiments/bf16x3_probe && experiments/bf16x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

// ---------------------------------------------------------------- f32 reference loop: A [K][M], B [K][N] (k-major), C [M][N]
__global__ __launch_bounds__(256, 3) void k_f32(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    __shared__ float As[2][16][128], Bs[2][16][128];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int lk = tid >> 5, lc = (tid & 31) * 4;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float4 ra0, ra1, rb0, rb1;
    auto ld = [&](int kt) {
        ra0 = *(const float4*)(A + (size_t)(kt * 16 + lk) * M + m0 + lc); ra1 = *(const float4*)(A + (size_t)(kt * 16 + lk + 8) * M + m0 + lc);
        rb0 = *(const float4*)(B + (size_t)(kt * 16 + lk) * N + n0 + lc); rb1 = *(const float4*)(B + (size_t)(kt * 16 + lk + 8) * N + n0 + lc);
    };
    auto st = [&](int buf) {
        *(float4*)&As[buf][lk][lc] = ra0; *(float4*)&As[buf][lk + 8][lc] = ra1; *(float4*)&Bs[buf][lk][lc] = rb0; *(float4*)&Bs[buf][lk + 8][lc] = rb1;
    };
    const int nk = K / 16;
    ld(0); st(0); __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) ld(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const int kr = kk + (lane >> 5);
            const float a0 = As[cur][kr][wm * 64 + (lane & 31)], a1 = As[cur][kr][wm * 64 + 32 + (lane & 31)];
            const float b0 = Bs[cur][kr][wn * 64 + (lane & 31)], b1 = Bs[cur][kr][wn * 64 + 32 + (lane & 31)];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) st(cur ^ 1);
        __syncthreads();
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = n0 + wn * 64 + j * 32 + (lane & 31);
        C[(size_t)row * N + col] = acc[i][j][r];
    }
}

// ---------------------------------------------------------------- bf16 split loop: A planes [3][M][K], B planes [3][N][K] (k contiguous: the
// operand order of v_mfma_f32_32x32x16_bf16, lane = (row, 8 consecutive k)); NPROD = 6 or 3 products
template <int NPROD>
__global__ __launch_bounds__(256, 2) void k_split(const u16* __restrict__ A, const u16* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    // per step of 16 k: three planes of 128 rows x 16 bf16 (32 B; padded to 48 B: conflict-free 16-byte reads) per operand, double buffered
    constexpr int RP = 24;                                          // u16 per row incl. padding
    __shared__ __attribute__((aligned(16))) u16 As[2][3][128 * RP], Bs[2][3][128 * RP];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int lr = tid >> 1, lh = tid & 1;                          // loader: row, 16-byte half
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    uint4 ra[3], rb[3];
    auto ld = [&](int kt) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            ra[p] = *(const uint4*)(A + ((size_t)p * M + m0 + lr) * K + kt * 16 + lh * 8);
            rb[p] = *(const uint4*)(B + ((size_t)p * N + n0 + lr) * K + kt * 16 + lh * 8);
        }
    };
    auto st = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 3; ++p) { *(uint4*)&As[buf][p][lr * RP + lh * 8] = ra[p]; *(uint4*)&Bs[buf][p][lr * RP + lh * 8] = rb[p]; }
    };
    const int nk = K / 16;
    ld(0); st(0); __syncthreads();
    const int l31 = lane & 31, lk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) ld(kt + 1);
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i][p] = *(const bf16x8*)&As[cur][p][(wm * 64 + i * 32 + l31) * RP + lk * 8];
                b[i][p] = *(const bf16x8*)&Bs[cur][p][(wn * 64 + i * 32 + l31) * RP + lk * 8];
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // smallest terms first
                if (NPROD == 6) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
            }
        if (kt + 1 < nk) st(cur ^ 1);
        __syncthreads();
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = n0 + wn * 64 + j * 32 + (lane & 31);
        C[(size_t)row * N + col] = acc[i][j][r];
    }
}

static void split3(float x, u16* hi, u16* mid, u16* lo) {           // truncation split: x = hi + mid + lo exactly (24 = 8 + 8 + 8 bits)
    union FU { float f; uint32_t u; };
    auto trunc16 = [](float v) { FU t; t.f = v; t.u &= 0xFFFF0000u; return t.f; };
    auto bits = [](float v) { FU t; t.f = v; return (u16)(t.u >> 16); };
    const float h = trunc16(x), r1 = x - h, m = trunc16(r1), l = trunc16(r1 - m);
    *hi = bits(h); *mid = bits(m); *lo = bits(l);
}

int main() {
    const int M = 256, K = 320, N = 163840;
    std::vector<float> A((size_t)K * M), B((size_t)K * N);
    srand(1);
    auto rnd = []() { float u = 0; for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX; return u - 6.0f; };   // ~N(0,1)
    for (auto& v : A) v = 0.05f * rnd();
    for (auto& v : B) v = fmaxf(rnd() + 1.0f, 0.0f);                  // post-ReLU-like, positive mean
    std::vector<u16> As((size_t)3 * M * K), Bs((size_t)3 * N * K);
    for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) {
        u16 h, mi, l; split3(A[(size_t)k * M + m], &h, &mi, &l);
        As[((size_t)0 * M + m) * K + k] = h; As[((size_t)1 * M + m) * K + k] = mi; As[((size_t)2 * M + m) * K + k] = l;
    }
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) {
        u16 h, mi, l; split3(B[(size_t)k * N + n], &h, &mi, &l);
        Bs[((size_t)0 * N + n) * K + k] = h; Bs[((size_t)1 * N + n) * K + k] = mi; Bs[((size_t)2 * N + n) * K + k] = l;
    }
    float *dA, *dB, *dC; u16 *dAs, *dBs;
    CHECK(hipMalloc(&dA, A.size() * 4)); CHECK(hipMalloc(&dB, B.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
    CHECK(hipMalloc(&dAs, As.size() * 2)); CHECK(hipMalloc(&dBs, Bs.size() * 2));
    CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dAs, As.data(), As.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dBs, Bs.data(), Bs.size() * 2, hipMemcpyHostToDevice));
    // f64 reference on a sample of outputs
    const int SM = 64, SN = 256;
    std::vector<double> ref((size_t)SM * SN);
    for (int i = 0; i < SM; ++i) for (int j = 0; j < SN; ++j) {
        const int m = (i * 37) % M, n = (int)(((size_t)j * 6151) % N);
        double s = 0; for (int k = 0; k < K; ++k) s += (double)A[(size_t)k * M + m] * (double)B[(size_t)k * N + n];
        ref[(size_t)i * SN + j] = s;
    }
    std::vector<float> C((size_t)M * N);
    dim3 grid(N / 128, M / 128), block(256);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CHECK(hipDeviceSynchronize());
        float best = 1e9f, tot = 0;
        for (int i = 0; i < 10; ++i) { CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms); tot += ms; }
        CHECK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        double emax = 0, e2 = 0, rmax = 0;
        for (int i = 0; i < SM; ++i) for (int j = 0; j < SN; ++j) {
            const int m = (i * 37) % M, n = (int)(((size_t)j * 6151) % N);
            const double d = fabs((double)C[(size_t)m * N + n] - ref[(size_t)i * SN + j]);
            emax = fmax(emax, d); e2 += d * d; rmax = fmax(rmax, fabs(ref[(size_t)i * SN + j]));
        }
        printf("%-34s %8.1f us (min %8.1f)  %6.1f TFLOP/s f32-equivalent   err vs f64: max %.3e  rms %.3e  (|out| up to %.1f)\n", name, tot * 100, best * 1000,
               2.0 * M * N * K / (tot / 10 * 1e-3) / 1e12, emax, sqrt(e2 / (SM * SN)), rmax);
    };
    run("f32  v_mfma_f32_32x32x2_f32", [&]() { hipLaunchKernelGGL(k_f32, grid, block, 0, 0, dA, dB, dC, M, N, K); });
    run("bf16 split, 6 products (x16_bf16)", [&]() { hipLaunchKernelGGL(k_split<6>, grid, block, 0, 0, dAs, dBs, dC, M, N, K); });
    run("bf16 split, 3 products (x16_bf16)", [&]() { hipLaunchKernelGGL(k_split<3>, grid, block, 0, 0, dAs, dBs, dC, M, N, K); });
    return 0;
}
