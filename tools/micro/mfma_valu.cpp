// Can the f32 matrix pipe and the packed-f32 vector pipe of a CU run at full rate at the same time?
// Kernel A: v_mfma_f32_32x32x2_f32 stream.  Kernel B: v_pk_fma_f32 stream.  Each alone, then both concurrently on two
// streams with workgroups of both kinds resident on every CU (and, variant C, both instruction kinds in one wave).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu.cpp -o mfma_valu && ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + u, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_valu(float* out, int iters) {
    f32x2 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x2{0.f, 0.f};
    f32x2 a = {threadIdx.x * 1e-3f, 0.5f}, b = {1.0f, 2.0f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] = __builtin_elementwise_fma(a, b, acc[i]);
    }
    f32x2 s = {0.f, 0.f};
    for (int i = 0; i < 32; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1];
}
__global__ __launch_bounds__(256) void k_both(float* out, int iters) {      // both kinds interleaved in one wave
    f32x16 acc[4]; f32x2 vacc[32];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 32; ++i) vacc[i] = f32x2{0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f; f32x2 va = {a, 0.5f}, vb = {1.0f, 2.0f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + u, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 8; ++v) vacc[(i * 8 + v) & 31] = __builtin_elementwise_fma(va, vb, vacc[(i * 8 + v) & 31]);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 32; ++i) s += vacc[i][0] + vacc[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *o1, *o2; hipMalloc(&o1, 4096 * 256 * 4); hipMalloc(&o2, 4096 * 256 * 4);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipEvent_t e0, e1, f0, f1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f0); hipEventCreate(&f1);
    const int itm = 4000, itv = 8000, blocks = 512;            // 2 workgroups of each kind per CU
    const double fm = (double)blocks * 4 * itm * 32 * 4096.0, fv = (double)blocks * 256 * (double)itv * 64 * 4.0;
    float ms;
    hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, s1, o1, 10); hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, s2, o2, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, s1); hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, s1, o1, itm); hipEventRecord(e1, s1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("MFMA alone : %7.3f ms  %6.1f TFLOP/s\n", ms, fm / ms / 1e9);
    hipEventRecord(f0, s2); hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, s2, o2, itv); hipEventRecord(f1, s2); hipEventSynchronize(f1);
    hipEventElapsedTime(&ms, f0, f1); printf("VALU alone : %7.3f ms  %6.1f TFLOP/s\n", ms, fv / ms / 1e9);
    hipDeviceSynchronize();
    hipEventRecord(e0, s1); hipEventRecord(f0, s2);
    hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, s1, o1, itm); hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, s2, o2, itv);
    hipEventRecord(e1, s1); hipEventRecord(f1, s2); hipEventSynchronize(e1); hipEventSynchronize(f1);
    float ma, mb; hipEventElapsedTime(&ma, e0, e1); hipEventElapsedTime(&mb, f0, f1);
    printf("concurrent : MFMA %7.3f ms (%6.1f TF)  VALU %7.3f ms (%6.1f TF)  sum over max time %6.1f TFLOP/s\n", ma, fm / ma / 1e9, mb, fv / mb / 1e9,
           (fm + fv) / (ma > mb ? ma : mb) / 1e9);
    const double fb = (double)768 * 4 * 2000 * 32 * 4096.0 + (double)768 * 256 * 2000.0 * 8 * 4 * 8 * 4.0;
    hipLaunchKernelGGL(k_both, dim3(768), dim3(256), 0, s1, o1, 10); hipDeviceSynchronize();
    hipEventRecord(e0, s1); hipLaunchKernelGGL(k_both, dim3(768), dim3(256), 0, s1, o1, 2000); hipEventRecord(e1, s1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("one wave, both kinds (8 pk_fma per MFMA): %7.3f ms  %6.1f TFLOP/s\n", ms, fb / ms / 1e9);
    return 0;
}
