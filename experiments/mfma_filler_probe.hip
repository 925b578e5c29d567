// Micro-probe: what does a vector instruction between two independent v_mfma_f32_16x16x4_f32 cost (one wave per SIMD)?
//   hipcc -O3 --offload-arch=gfx950 experiments/mfma_filler_probe.hip -o experiments/mfma_filler_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// K fillers after every MFMA; KIND 0: v_add_f32 on 4 rotating registers, 1: v_cndmask (VOP3, SGPR-pair mask), 2: ds_read_b128 every 4th MFMA + adds, 3: s_nop 0
template <int K, int KIND>
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float L[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) L[i] = 0.5f;
    __syncthreads();
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = (float)lane, b = 1.0f;
    float x0 = 1.0f, x1 = 2.0f, x2 = 3.0f, x3 = 4.0f, y = 0.25f;
    f32x4 rd = {0, 0, 0, 0};
    const unsigned long long m = 0x5555555555555555ull;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) {
                    if ((k & 3) == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(y));
                    if ((k & 3) == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x1) : "v"(y));
                    if ((k & 3) == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x2) : "v"(y));
                    if ((k & 3) == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x3) : "v"(y));
                } else if (KIND == 1) {
                    if ((k & 3) == 0) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x0) : "v"(y), "s"(m));
                    if ((k & 3) == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x1) : "v"(y), "s"(m));
                    if ((k & 3) == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x2) : "v"(y), "s"(m));
                    if ((k & 3) == 3) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x3) : "v"(y), "s"(m));
                } else if (KIND == 2) {
                    if (k == 0 && (i & 3) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(rd) : "v"(lane * 16) : "memory");
                    else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(y));
                } else {
                    asm volatile("s_nop 0");
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = x0 + x1 + x2 + x3 + rd[0];
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int K, int KIND> void run(float* out, long long* cyc, int blocks = 256) {
    const int iters = 500;
    hipLaunchKernelGGL((probe<K, KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((probe<K, KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("kind %d, %d fillers per MFMA, %d workgroups: %.1f cycles per MFMA\n", KIND, K, blocks, (double)c / iters / 32);
}

int main() {
    float* out; long long* cyc; (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 8);
    run<0, 0>(out, cyc); run<1, 0>(out, cyc); run<2, 0>(out, cyc); run<3, 0>(out, cyc); run<4, 0>(out, cyc); run<5, 0>(out, cyc); run<6, 0>(out, cyc); run<8, 0>(out, cyc);
    run<1, 1>(out, cyc); run<2, 1>(out, cyc); run<3, 1>(out, cyc); run<4, 1>(out, cyc); run<6, 1>(out, cyc);
    run<1, 2>(out, cyc); run<3, 2>(out, cyc); run<4, 2>(out, cyc);
    run<1, 3>(out, cyc); run<3, 3>(out, cyc); run<6, 3>(out, cyc);
    printf("--- two waves per SIMD (512 workgroups)\n");
    run<0, 0>(out, cyc, 512); run<1, 0>(out, cyc, 512); run<4, 0>(out, cyc, 512); run<8, 0>(out, cyc, 512); run<4, 2>(out, cyc, 512);
    return 0;
}
