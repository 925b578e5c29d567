// What shader clock does the board hold while every SIMD issues f32 matrix instructions back to back?
// s_memtime counts shader-clock cycles, s_memrealtime the constant 100 MHz reference: their ratio over a long loop is the clock.
//   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip && ./clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>                                    // 0: dependent scalar adds only (light load)  1: 16x16x4 f32 MFMA chains on every wave
__global__ __launch_bounds__(256) void k_probe(unsigned long long* out, int iters, float seed) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        } else {
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (s == 1234.5f) out[0] = 0;
}

int main() {
    const int nblk = 256 * 8;                          // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    unsigned long long* d; hipMalloc(&d, nblk * 16);
    unsigned long long* h = new unsigned long long[2 * nblk];
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode) hipLaunchKernelGGL(k_probe<1>, dim3(nblk), dim3(256), 0, 0, d, 400000, 1.0f);
            else hipLaunchKernelGGL(k_probe<0>, dim3(256), dim3(64), 0, 0, d, 2000000, 1.0f);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, d, (mode ? nblk : 256) * 16, hipMemcpyDeviceToHost);
            double cyc = 0, ref = 0; const int n = mode ? nblk : 256;
            for (int i = 0; i < n; ++i) { cyc += h[2 * i]; ref += h[2 * i + 1]; }
            const double flops = mode ? (double)nblk * 4 * 400000.0 * 8 * 2048.0 : 0.0;
            printf("%s: kernel %.1f ms, shader cycles / 100 MHz reference ticks = %.3f -> %.0f MHz%s", mode ? "f32 MFMA on every SIMD" : "idle-ish (one wave per CU, s_nop)", ms,
                   cyc / ref, cyc / ref * 100.0, mode ? "" : "\n");
            if (mode) printf(", %.1f TFLOP/s\n", flops / ms / 1e9);
        }
    return 0;
}
