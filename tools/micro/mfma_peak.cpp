// Peak rate of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 on gfx950 as a function of waves per SIMD and of the
// number of independent accumulators -- the ceiling k_conv_igemm / k_corr_gemm can be priced against.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.cpp -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k32(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + u, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k16(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + u, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
static void run(const char* name, K kern, int nacc, int waves_per_simd, double flop_per_mfma) {
    float* out; hipMalloc(&out, 256 * 4 * 8 * 64 * 4 * 4);
    const int iters = 2000;
    dim3 grid(256 * waves_per_simd), block(256);          // 4 waves per block -> one per SIMD per resident block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid.x * 4 * iters * 8 * nacc * flop_per_mfma;
    printf("%-14s acc=%d waves/SIMD=%d : %7.1f TFLOP/s\n", name, nacc, waves_per_simd, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 3; ++w) {
        run("32x32x2 f32", k32<1>, 1, w, 4096.0);
        run("32x32x2 f32", k32<2>, 2, w, 4096.0);
        run("32x32x2 f32", k32<4>, 4, w, 4096.0);
        run("16x16x4 f32", k16<4>, 4, w, 2048.0);
        run("16x16x4 f32", k16<8>, 8, w, 2048.0);
    }
    return 0;
}
