// What limits "8 x ds_read_b128 -> 32 x v_mfma_f32_32x32x2_f32" (the inner step of k_conv_igemm) below the pure-MFMA
// rate?  Variants: (0) MFMA only, operands rotate through registers; (1) + LDS fragment reads each step;
// (2) + workgroup barrier each step; (3) reads split in two halves interleaved with the MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_lds.cpp -o mfma_lds && ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define KS 20

template <int MODE>
__global__ __launch_bounds__(256, 3) void k(float* out, int iters, const float* __restrict__ src) {
    __shared__ __attribute__((aligned(16))) float As[2][128][KS];
    __shared__ __attribute__((aligned(16))) float Bs[2][136][KS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, lh = lane >> 5, wm = wv >> 1, wn = wv & 1;
    for (int i = tid; i < 2 * 128 * KS; i += 256) (&As[0][0][0])[i] = 0.001f * (i % 13);
    for (int i = tid; i < 2 * 136 * KS; i += 256) (&Bs[0][0][0])[i] = 0.002f * (i % 7);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 a0[2], a1[2], b0[2], b1[2];
    for (int h = 0; h < 2; ++h) {
        a0[h] = *(const f32x4*)&As[0][wm * 64 + l31][8 * lh + 4 * h]; a1[h] = *(const f32x4*)&As[0][wm * 64 + 32 + l31][8 * lh + 4 * h];
        b0[h] = *(const f32x4*)&Bs[0][4 + wn * 64 + l31][8 * lh + 4 * h]; b1[h] = *(const f32x4*)&Bs[0][4 + wn * 64 + 32 + l31][8 * lh + 4 * h];
    }
    const float4* gsrc = (const float4*)src + (size_t)(blockIdx.x & 1023) * 2048 + tid;
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
        float4 ga0, ga1, gb0, gb1;
        if (MODE >= 4) { ga0 = gsrc[(it & 7) * 256 * 8]; ga1 = gsrc[(it & 7) * 256 * 8 + 256]; }       // weights-like loads (L2 resident)
        if (MODE >= 5) { gb0 = gsrc[(it & 7) * 256 * 8 + 512]; gb1 = gsrc[(it & 7) * 256 * 8 + 768]; }
        if (MODE == 1 || MODE == 2 || MODE >= 4) {
            for (int h = 0; h < 2; ++h) {
                a0[h] = *(const f32x4*)&As[cur][wm * 64 + l31][8 * lh + 4 * h]; a1[h] = *(const f32x4*)&As[cur][wm * 64 + 32 + l31][8 * lh + 4 * h];
                b0[h] = *(const f32x4*)&Bs[cur][4 + wn * 64 + l31][8 * lh + 4 * h]; b1[h] = *(const f32x4*)&Bs[cur][4 + wn * 64 + 32 + l31][8 * lh + 4 * h];
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (MODE == 3) {          // read the OTHER half for the next use while this half's MFMAs run
                const int o = h ^ 1;
                a0[o] = *(const f32x4*)&As[cur][wm * 64 + l31][8 * lh + 4 * o]; a1[o] = *(const f32x4*)&As[cur][wm * 64 + 32 + l31][8 * lh + 4 * o];
                b0[o] = *(const f32x4*)&Bs[cur][4 + wn * 64 + l31][8 * lh + 4 * o]; b1[o] = *(const f32x4*)&Bs[cur][4 + wn * 64 + 32 + l31][8 * lh + 4 * o];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[h][j], b0[h][j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[h][j], b1[h][j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[h][j], b0[h][j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[h][j], b1[h][j], acc[1][1], 0, 0, 0);
            }
            if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE >= 4) {
            *(float4*)&As[cur ^ 1][tid & 127][4 * (tid >> 7)] = ga0; *(float4*)&As[cur ^ 1][tid & 127][8 + 4 * (tid >> 7)] = ga1;
        }
        if (MODE >= 5) {
            const int n = 4 + 4 * ((tid & 3) + 4 * wv), kk = (tid >> 2) & 15;
            Bs[cur ^ 1][n][kk] = gb0.x; Bs[cur ^ 1][n + 1][kk] = gb0.y; Bs[cur ^ 1][n + 2][kk] = gb0.z; Bs[cur ^ 1][n + 3][kk] = gb0.w;
            Bs[cur ^ 1][n + 64][kk] = gb1.x; Bs[cur ^ 1][n + 65][kk] = gb1.y; Bs[cur ^ 1][n + 66][kk] = gb1.z; Bs[cur ^ 1][n + 67][kk] = gb1.w;
        }
        if (MODE == 2 || MODE >= 4) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}
template <typename K> static void run(const char* name, K kern, int blocks_per_cu) {
    float* out; hipMalloc(&out, 256 * 3 * 256 * 4);
    static float* src = nullptr; if (!src) { hipMalloc(&src, (size_t)1024 * 2048 * 16 + 8 * 256 * 8 * 16 + 4096); hipMemset(src, 0, (size_t)1024 * 2048 * 16 + 8 * 256 * 8 * 16 + 4096); }
    const int iters = 4000;
    dim3 grid(256 * blocks_per_cu), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, 10, src); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters, src); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s blocks/CU=%d : %6.1f TFLOP/s\n", name, blocks_per_cu, (double)grid.x * 4 * iters * 32 * 4096.0 / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    for (int b = 1; b <= 3; ++b) {
        run("MFMA only (operands from 32 registers)", k<0>, b);
        run("+ 8 ds_read_b128 per step", k<1>, b);
        run("+ reads + barrier per step", k<2>, b);
        run("reads in two halves, interleaved", k<3>, b);
        run("reads + barrier + weight loads/stores", k<4>, b);
        run("  + input loads, transposing stores", k<5>, b);
    }
    return 0;
}
