// Probe: issue cost (cycles per instruction, one wave per SIMD) of the vector instructions the bf16x3 split could use, and whether
// v_cvt_pk_bf16_f32 + v_dot2_f32_bf16 give an exact residual (x - bf16(x)).  Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))

template <int WHICH> __global__ __launch_bounds__(256, 1) void k_rate(unsigned long long* out, float* sink) {
    float a0 = threadIdx.x * 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float c = 1.0001f; const f2 cc = {c, c};
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (WHICH == 0) { REP256(asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
    if (WHICH == 1) { REP256(asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));) }
    if (WHICH == 2) { REP256(asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %5, %6\n v_cvt_pk_bf16_f32 %2, %6, %7\n v_cvt_pk_bf16_f32 %3, %7, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
    if (WHICH == 3) { REP256(asm volatile("v_dot2_f32_bf16 %0, %4, %5, %0\n v_dot2_f32_bf16 %1, %4, %5, %1\n v_dot2_f32_bf16 %2, %4, %5, %2\n v_dot2_f32_bf16 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1));) }
    if (WHICH == 4) { REP256(asm volatile("v_perm_b32 %0, %4, %5, %6\n v_perm_b32 %1, %5, %4, %6\n v_perm_b32 %2, %4, %5, %6\n v_perm_b32 %3, %5, %4, %6" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "s"(0x07060302u));) }
    if (WHICH == 5) { REP256(asm volatile("v_and_b32 %0, %4, %0\n v_sub_f32 %1, %1, %0\n v_and_b32 %2, %4, %2\n v_sub_f32 %3, %3, %2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0xFFFF0000u));) }
    if (WHICH == 6) { REP256(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));) }
    if (WHICH == 7) { REP256(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));) }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[WHICH] = t1 - t0;
    sink[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1] + (float)(u0 + u1 + u2 + u3);
}

// beside matrix instructions: 16 bf16 MFMAs (32x32x16: 8 passes each) with N vector instructions after each
template <int NV, int WHICH> __global__ __launch_bounds__(256, 1) void k_mix(unsigned long long* out, float* sink) {
    typedef float f16v __attribute__((ext_vector_type(16)));
    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f16v acc0 = {0}, acc1 = {0};
    b8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)threadIdx.x; y[i] = (__bf16)1.0f; }
    float a0 = threadIdx.x * 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    f2 p0 = {a0, a1}, p1 = {a2, a3};
    const float c = 1.0001f; const f2 cc = {c, c};
    unsigned u0 = threadIdx.x, u1 = u0 + 1;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int it = 0; it < 32; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc0, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV / 2; ++v) {
            if (WHICH == 0) asm volatile("v_fma_f32 %0, %0, %2, %0\n v_fma_f32 %1, %1, %2, %1" : "+v"(a0), "+v"(a1) : "v"(c));
            if (WHICH == 1) asm volatile("v_pk_fma_f32 %0, %0, %2, %0\n v_pk_fma_f32 %1, %1, %2, %1" : "+v"(p0), "+v"(p1) : "v"(cc));
            if (WHICH == 3) asm volatile("v_dot2_f32_bf16 %0, %2, %3, %0\n v_dot2_f32_bf16 %1, %2, %3, %1" : "+v"(a0), "+v"(a1) : "v"(u0), "v"(u1));
        }
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc1, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV / 2; ++v) {
            if (WHICH == 0) asm volatile("v_fma_f32 %0, %0, %2, %0\n v_fma_f32 %1, %1, %2, %1" : "+v"(a2), "+v"(a3) : "v"(c));
            if (WHICH == 1) asm volatile("v_pk_fma_f32 %0, %0, %2, %0\n v_pk_fma_f32 %1, %1, %2, %1" : "+v"(p0), "+v"(p1) : "v"(cc));
            if (WHICH == 3) asm volatile("v_dot2_f32_bf16 %0, %2, %3, %0\n v_dot2_f32_bf16 %1, %2, %3, %1" : "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    float s = 0; for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    sink[threadIdx.x] = s + a0 + a1 + a2 + a3 + p0[0] + p0[1] + p1[0] + p1[1];
}

__global__ void k_exact(const float* x, float* r1, float* r2, unsigned* pk, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    unsigned h;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
    float ra, rb;
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(ra) : "v"(h), "v"(0xBF80u), "v"(a));
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(rb) : "v"(h), "v"(0xBF800000u), "v"(b));
    pk[i] = h; r1[2 * i] = ra; r1[2 * i + 1] = rb;
    unsigned m;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m) : "v"(ra), "v"(rb));
    float sa, sb;
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(sa) : "v"(m), "v"(0xBF80u), "v"(ra));
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(sb) : "v"(m), "v"(0xBF800000u), "v"(rb));
    r2[2 * i] = sa; r2[2 * i + 1] = sb;
}

static float bf16_rne(float v) { unsigned u; memcpy(&u, &v, 4); u += 0x7FFFu + ((u >> 16) & 1u); u &= 0xFFFF0000u; float r; memcpy(&r, &u, 4); return r; }

int main() {
    unsigned long long* d; float* sink; hipMalloc(&d, 64); hipMalloc(&sink, 4096);
    unsigned long long h[8];
    const char* names[8] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_pk_bf16_f32", "v_dot2_f32_bf16", "v_perm_b32", "v_and+v_sub (dependent pairs)", "v_pk_add_f32", "v_pk_mul_f32"};
    for (int rep = 0; rep < 2; ++rep) {
        k_rate<0><<<1, 256>>>(d, sink); k_rate<1><<<1, 256>>>(d, sink); k_rate<2><<<1, 256>>>(d, sink); k_rate<3><<<1, 256>>>(d, sink);
        k_rate<4><<<1, 256>>>(d, sink); k_rate<5><<<1, 256>>>(d, sink); k_rate<6><<<1, 256>>>(d, sink); k_rate<7><<<1, 256>>>(d, sink);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("%-32s %.2f cycles / instruction (1024 instructions, one wave per SIMD)\n", names[i], h[i] / 1024.0);
#define MIX(NV, W, label) do { for (int r = 0; r < 2; ++r) { k_mix<NV, W><<<1, 256>>>(d, sink); hipDeviceSynchronize(); } hipMemcpy(h, d, 8, hipMemcpyDeviceToHost); \
    printf("64 MFMA 32x32x16 bf16 + %2d x %-16s after each: %6llu cycles = %.1f per MFMA (MFMA alone: 32)\n", NV, label, h[0], h[0] / 64.0); } while (0)
    MIX(0, 0, "-"); MIX(4, 0, "v_fma_f32"); MIX(6, 0, "v_fma_f32"); MIX(8, 0, "v_fma_f32"); MIX(12, 0, "v_fma_f32"); MIX(16, 0, "v_fma_f32");
    MIX(4, 1, "v_pk_fma_f32"); MIX(8, 1, "v_pk_fma_f32"); MIX(8, 3, "v_dot2_f32_bf16");
    // exactness
    const int n = 1 << 20;
    float* x = (float*)malloc(n * 4); float *r1 = (float*)malloc(n * 4), *r2 = (float*)malloc(n * 4); unsigned* pk = (unsigned*)malloc(n * 2);
    srand(1);
    for (int i = 0; i < n; ++i) { const float m = (rand() / (float)RAND_MAX * 2 - 1); x[i] = ldexpf(m, rand() % 60 - 30); }
    x[0] = 0.0f; x[1] = -0.0f; x[2] = 1e-40f; x[3] = 3.0e38f; x[4] = 1.0f; x[5] = -1.00390625f;
    float *dx, *dr1, *dr2; unsigned* dpk; hipMalloc(&dx, n * 4); hipMalloc(&dr1, n * 4); hipMalloc(&dr2, n * 4); hipMalloc(&dpk, n * 2);
    hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice);
    k_exact<<<n / 2 / 256, 256>>>(dx, dr1, dr2, dpk, n); hipDeviceSynchronize();
    hipMemcpy(r1, dr1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r2, dr2, n * 4, hipMemcpyDeviceToHost); hipMemcpy(pk, dpk, n * 2, hipMemcpyDeviceToHost);
    long bad_h = 0, bad_r1 = 0, bad_r2 = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned hb = (i & 1) ? pk[i / 2] >> 16 : pk[i / 2] & 0xFFFFu;
        const float hv = bf16_rne(x[i]); unsigned hu; memcpy(&hu, &hv, 4);
        if ((hu >> 16) != hb) { if (bad_h < 5) printf("  hi differs: x %a  got %04x want %04x\n", x[i], hb, hu >> 16); ++bad_h; }
        const float e1 = x[i] - hv;
        if (e1 != r1[i] && !(std::isinf(hv))) { if (bad_r1 < 5) printf("  r1 differs: x %a  got %a want %a\n", x[i], r1[i], e1); ++bad_r1; }
        const float mv = bf16_rne(r1[i]), e2 = r1[i] - mv;
        if (e2 != r2[i]) { if (bad_r2 < 5) printf("  r2 differs: r1 %a  got %a want %a\n", r1[i], r2[i], e2); ++bad_r2; }
    }
    printf("exactness over %d values: hi (RNE) mismatches %ld, r1 = x - hi mismatches %ld, r2 = r1 - mid mismatches %ld\n", n, bad_h, bad_r1, bad_r2);
    return 0;
}
