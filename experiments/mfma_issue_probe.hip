// Micro-probe: issue rate of v_mfma_f32_16x16x4_f32 / 32x32x2 from one wave per SIMD, operands in registers or read from LDS.
//   hipcc -O3 --offload-arch=gfx950 experiments/mfma_issue_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float L[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) L[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    f32x16 big[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) big[i][j] = 0.0f;
    float a = (float)lane, b = 1.0f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {                    // 32 independent 16x16x4, register operands
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        } else if (MODE == 1) {             // operands from ds_read_b128, 3 reads per 8 MFMAs (as k_conv_wino)
            const float* p = &L[(lane * 4 + (it & 3) * 256) & 4095];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 fa0 = *(const f32x4*)(p + g * 1024), fa1 = *(const f32x4*)(p + g * 1024 + 256), fb = *(const f32x4*)(p + g * 1024 + 512);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[8 * g + 2 * e] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[e], fb[e], acc[8 * g + 2 * e], 0, 0, 0);
                    acc[8 * g + 2 * e + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[e], fb[e], acc[8 * g + 2 * e + 1], 0, 0, 0);
                }
            }
        } else if (MODE == 2) {             // 8 independent 32x32x2 (the same FLOPs as 32 of 16x16x4 ... x2: 16 per iteration = same FLOPs)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) big[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[i], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += big[i][0] + big[i][15];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out; long long* cyc; hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode)
        for (int blocks : {256, 512}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double flop = (double)blocks * 4 * iters * 32 * 2048.0;
            printf("mode %d blocks %d: %.3f ms, %.1f TFLOP/s, wave clock ticks per iteration %.1f (s_memtime, 100 MHz ref?)\n", mode, blocks, ms, flop / ms / 1e9, (double)c / iters);
        }
    return 0;
}
