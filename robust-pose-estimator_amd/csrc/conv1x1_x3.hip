// LABELLED VARIANT (never in the headline number): 1x1 stride-1 convolutions (plain GEMMs: BasicMotionEncoder.convc1 behind the correlation
// lookup, core/RAFT/core/update.py; call sites core/pose/pose_net.py:47,65,129) on the 16-bit matrix cores, every f32 product as SIX bf16
// products of a three-way split (x = hi + mid + lo, round-to-nearest parts, residuals exact; conv_wino_x3.hip explains the arithmetic).
// The variant of conv1x1.hip's rpe_conv1x1; selected by raft.CONV_BF16X3.
//
// A GEMM is the friendliest shape for the scheme: no transform, the vector work is the split of the activations alone (5.5 instructions per
// value, each value feeding 24 matrix instructions), so the kernel is bound by the matrix pipe and two workgroups fit a CU:
//   * workgroup = 4 waves, 128 output channels x 256 pixels (a run of the flattened h*w index of one image); wave w owns pixels
//     [64 w, 64 w + 64): 4 x 2 blocks of 32 x 32 = 128 accumulators;
//   * the weights (A: pre-split at pack time, 12 KB per 16-channel step and 128-channel tile) are the one operand the waves share: LDS-DMA
//     into a ring of three buffers, each wave brings a quarter; fragments are read from LDS plane by plane ([k half][32 co][8 ci]:
//     16 lanes' 16-byte reads tile the 64 banks);
//   * the activations (B) of a wave are its own: each wave DMAs ITS 64 pixels x 16 channels into its own LDS region, reads them back as
//     (pixel, 8 channels) per lane, splits and packs in registers one pixel block (half a step) ahead -- no barrier is needed for them;
//   * ONE barrier per step (the shared weights), every request is an LDS-DMA (they complete in order: the waits count them), pixels past the
//     end of the map read the map's last quad and are never stored: no masks, no patch-ups.
#include "wino_common.h"
#include <type_traits>
#pragma clang diagnostic ignored "-Wunused-lambda-capture"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define G3_CO 128
#define G3_NCB 4
#define G3K 16
#define G3_A_STEP 12288                          // bytes of A per (step, 128-channel tile): [cb 4][plane 3][k half 2][32 co][8 ci] bf16

#ifdef G3_TIMING
__device__ unsigned long long g_g3_timing[8];
extern "C" int rpe_debug_g3_timing(unsigned long long* out8) { return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_g3_timing), 64) == hipSuccess ? 0 : -1; }
#endif

struct G1X3P {
    const float* x; long long xbs;
    const unsigned short* wp; int cin, cout, coP, hw;
    const float* bias;
    float* out; long long obs; float* out2; long long o2bs;
    int mode;
};

__global__ __launch_bounds__(256, 2) void k_conv1x1_x3(G1X3P P) {
    constexpr int NCB = G3_NCB;
    // LDS: A ring 3 x 12 KB | B: 2 buffers x 4 waves x (16 channels x 64 pixels) floats = 32 KB
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * G3_A_STEP + 2 * 4 * 4096 + 512];
    constexpr int B_AT = 3 * G3_A_STEP, BIAS_AT = B_AT + 2 * 4 * 4096;               // (+ the tile's 128 bias values)
    asm volatile("" :: "s"(P.x), "s"(P.wp), "s"(P.out), "s"(P.bias), "s"(P.xbs), "s"(P.obs), "s"(P.cin), "s"(P.cout), "s"(P.coP), "s"(P.hw),
                 "s"(P.mode), "s"(P.out2), "s"(P.o2bs));
#ifdef G3_TIMING
    const unsigned long long T0 = __builtin_readcyclecounter();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = P.hw, bz = blockIdx.z, co0 = blockIdx.y * G3_CO;
    const int p0 = blockIdx.x * 256 + 64 * wv;                                     // the wave's first pixel
    const float* xb = P.x + (size_t)bz * P.xbs;
    const int nsteps = (P.cin + G3K - 1) / G3K;                                    // (the weights of channels past cin are zeros)
    const unsigned smem_lds = lds_addr_of(&smem[0]);
    if (tid < G3_CO) ((float*)&smem[BIAS_AT])[tid] = (P.bias && co0 + tid < P.cout) ? P.bias[co0 + tid] : 0.0f;   // (published by the loop's first barrier)

    // ---- requests.  A: chunks 3 w .. 3 w + 2 of the step's 12 KB (lane * 16 bytes each, the instruction offset moves both addresses).
    const unsigned a_voff = (unsigned)lane * 16u;
    const char* a_tile = (const char*)P.wp + (size_t)(co0 / G3_CO) * G3_A_STEP + (size_t)wv * 3072;
    const size_t a_stride = (size_t)(P.coP / G3_CO) * G3_A_STEP;
    auto dma_a = [&](int step, int slot) {
        const float* src = wave_uniform((const float*)(a_tile + (size_t)step * a_stride));
        const unsigned dst = smem_lds + (unsigned)slot * G3_A_STEP + (unsigned)wv * 3072u;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(a_voff), "s"(src), "s"(dst) : "memory");
    };
    // B: chunk j = channels 4 j .. 4 j + 3 of the step, lane -> (channel 4 j + (lane >> 4), pixel quad lane & 15); pixels past the map read its
    // last quad (computed, never stored)
    int pq = p0 + 4 * (lane & 15);
    pq = pq + 3 < hw ? pq : hw - 4;
    const unsigned b_voff = (unsigned)(((lane >> 4) * hw + pq) * 4);
    const unsigned b_region = smem_lds + B_AT + (unsigned)wv * 4096u;
    // cin % 16 != 0 (convc1: 324): the last step's channels past cin read channel cin - 1 again (valid memory; their weights are zeros --
    // so an Inf in that channel becomes NaN where the f32 kernel gives Inf: the one difference in special values)
    const bool ragged = (P.cin & (G3K - 1)) != 0;
    unsigned tail_voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int c = 4 * j + (lane >> 4);
        const int last = P.cin - 1 - (nsteps - 1) * G3K;
        c = c < last ? c : last;
        tail_voff[j] = (unsigned)((c * hw + pq) * 4);
    }
    auto dma_b = [&](int step, int slot) {
        const unsigned dst = b_region + (unsigned)slot * 16384u;
        const bool tail = ragged && step == nsteps - 1;                              // (wave-uniform)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* src = wave_uniform(xb + (size_t)(step * G3K + (tail ? 0 : 4 * j)) * hw);
            const unsigned vo = tail ? tail_voff[j] : b_voff;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vo), "s"(src), "s"(dst + 1024u * j) : "memory");
        }
    };

    // ---- B fragments: lane -> (pixel n = lane & 31 of the 32-pixel block, channels 8 (lane >> 5) + e): eight dwords, 256 bytes apart
    const unsigned rd_lane = b_region + (unsigned)((8 * (lane >> 5)) * 64 + (lane & 31)) * 4u;
    unsigned long long rv0 = 0, rv1 = 0, rv2 = 0, rv3 = 0;                          // (e, e + 1) pairs 0..3
    auto issue_reads = [&](int slot, int pb) {
        const unsigned a0 = rd_lane + (unsigned)slot * 16384u + (unsigned)pb * 128u, a1 = a0 + 1024u;
        asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:64" : "=v"(rv0) : "v"(a0));
        asm volatile("ds_read2_b32 %0, %1 offset0:128 offset1:192" : "=v"(rv1) : "v"(a0));
        asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:64" : "=v"(rv2) : "v"(a1));
        asm volatile("ds_read2_b32 %0, %1 offset0:128 offset1:192" : "=v"(rv3) : "v"(a1));
    };
    auto wait_reads = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(rv0), "v"(rv1), "v"(rv2), "v"(rv3) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    u32x4 B[2][3];                                    // [slot = pixel block][plane]
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    auto pack2 = [](float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_)); };
    // pair q of the lane's eight values -> element q of the three fragments of slot sl (round-to-nearest split, exact residuals)
    auto split_pair = [&](auto slc, auto qc, unsigned long long raw) {
        constexpr int sl = decltype(slc)::value, q = decltype(qc)::value;
        const float v0 = __builtin_bit_cast(float, (unsigned)raw), v1 = __builtin_bit_cast(float, (unsigned)(raw >> 32));
        unsigned ph = pack2(v0, v1);
        const float r10 = v0 - __builtin_bit_cast(float, ph << 16), r11 = v1 - __builtin_bit_cast(float, ph & 0xFFFF0000u);
        unsigned pm = pack2(r10, r11);
        const float r20 = r10 - __builtin_bit_cast(float, pm << 16), r21 = r11 - __builtin_bit_cast(float, pm & 0xFFFF0000u);
        unsigned pl = pack2(r20, r21);
        asm volatile("" : "+v"(ph), "+v"(pm), "+v"(pl));
        B[sl][0][q] = ph; B[sl][1][q] = pm; B[sl][2][q] = pl;
    };

    // ---- A fragments from LDS: (cb, plane) block of 1 KB = [k half][32 co][8 ci]; lane -> k half = lane >> 5, co = lane & 31
    const unsigned af_lane = smem_lds + (unsigned)(lane >> 5) * 512u + (unsigned)(lane & 31) * 16u;
    u32x4 Af[2][NCB];                                 // two planes in flight
    auto read_a = [&Af, af_lane](auto setc, int slot, auto planec) {
        constexpr int set = decltype(setc)::value, pl = decltype(planec)::value;
        const unsigned a = af_lane + (unsigned)slot * G3_A_STEP;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(Af[set][0]) : "v"(a), "n"((0 * 3 + pl) * 1024));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(Af[set][1]) : "v"(a), "n"((1 * 3 + pl) * 1024));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(Af[set][2]) : "v"(a), "n"((2 * 3 + pl) * 1024));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(Af[set][3]) : "v"(a), "n"((3 * 3 + pl) * 1024));
    };
    auto wait_a = [&Af](auto setc) {
        constexpr int set = decltype(setc)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(Af[set][0]), "v"(Af[set][1]), "v"(Af[set][2]), "v"(Af[set][3]) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x16 acc[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][pb][r] = 0.0f;
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1; typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 3> I3;
    // four matrix instructions: A plane (set) x B plane bp of pixel block pb, the four channel blocks
    auto mma4 = [&](auto setc, auto pbc, auto bpc) {
        constexpr int set = decltype(setc)::value, pb = decltype(pbc)::value, bp = decltype(bpc)::value;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
            acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Af[set][cb]), __builtin_bit_cast(bf16x8, B[pb][bp]), acc[cb][pb], 0, 0, 0);
    };
    // One half step = the 24 matrix instructions of pixel block pb, smallest terms first with each A plane used in one run:
    //   A lo x B hi | A mid x B mid, A mid x B hi | A hi x B lo, A hi x B mid, A hi x B hi
    // (planes: 0 hi, 1 mid, 2 lo).  On entry A's lo plane is in set 0.  Beside them the four pairs of ANOTHER pixel block's raw values
    // (already in rv0..3) are split into slot SP; `after` runs between the last two groups (the next reads / requests).
    auto half_step = [&](auto pbc, auto spc, int aslot, auto after) {
        typedef decltype(pbc) PB; typedef decltype(spc) SP;
        read_a(I1{}, aslot, I1{});                                   // mid -> set 1
        mma4(I0{}, PB{}, I0{});                                      // lo x hi
        split_pair(SP{}, I0{}, rv0);
        __builtin_amdgcn_sched_barrier(0);
        wait_a(I1{});
        read_a(I0{}, aslot, I0{});                                   // hi -> set 0 (lo is done)
        mma4(I1{}, PB{}, I1{});                                      // mid x mid
        split_pair(SP{}, I1{}, rv1);
        __builtin_amdgcn_sched_barrier(0);
        mma4(I1{}, PB{}, I0{});                                      // mid x hi
        split_pair(SP{}, I2{}, rv2);
        __builtin_amdgcn_sched_barrier(0);
        wait_a(I0{});
        mma4(I0{}, PB{}, I2{});                                      // hi x lo
        split_pair(SP{}, I3{}, rv3);
        __builtin_amdgcn_sched_barrier(0);
        mma4(I0{}, PB{}, I1{});                                      // hi x mid
        after();
        __builtin_amdgcn_sched_barrier(0);
        mma4(I0{}, PB{}, I0{});                                      // hi x hi
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: A(0), A(1), B(0), B(1) requested; the first pixel block's fragments of step 0
    dma_a(0, 0);
    dma_b(0, 0);
    dma_a(nsteps > 1 ? 1 : 0, 1);
    dma_b(nsteps > 1 ? 1 : 0, 1);
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                                 // A(0) and B(0) have landed (A(1), B(1): 3 + 4 younger)
    issue_reads(0, 0); wait_reads();
    split_pair(I0{}, I0{}, rv0); split_pair(I0{}, I1{}, rv1); split_pair(I0{}, I2{}, rv2); split_pair(I0{}, I3{}, rv3);
    issue_reads(0, 1); wait_reads();                                                 // (the second block of step 0: split beside the first half step)

    // ---- main loop.  Step s: BARRIER (A(s) is in LDS for everybody; everybody has left step s-1) -> request A(s+2) into the ring slot step
    // s-1 used; half step (s, block 0) while block 1 of step s is split; B(s+1) must have landed (3 younger requests: A(s+2));
    // half step (s, block 1) while block 0 of step s+1 is split; request B(s+2) into the buffer step s used.
    // Requests past the end re-read the last step (into buffers nobody reads again): the waits count requests, so none may be skipped.
#ifdef G3_TIMING
    const unsigned long long T1 = __builtin_readcyclecounter();
#endif
    int aslot = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int bslot = s & 1;
        const int s2 = s + 2 < nsteps ? s + 2 : nsteps - 1;
        // A(s): requested two steps ago; younger: B(s) [4, unless s < 2: the prologue's order], A(s+1) [3], B(s+1) [4]
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int a2 = aslot == 0 ? 2 : aslot - 1;                                    // ring slot of step s-1 = of step s+2
        dma_a(s2, a2);
        read_a(I0{}, aslot, I2{}); wait_a(I0{});                                     // lo plane of A(s)
        half_step(I0{}, I1{}, aslot, [&]() {
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                         // B(s+1) has landed (A(s+2) is younger)
            issue_reads(bslot ^ 1, 0); });
        wait_reads();
        read_a(I0{}, aslot, I2{}); wait_a(I0{});                                     // lo plane again for the second block
        half_step(I1{}, I0{}, aslot, [&]() {
            dma_b(s2, bslot);                                                        // (step s's raw values have all been read)
            issue_reads(bslot ^ 1, 1); });
        wait_reads();
        aslot = aslot == 2 ? 0 : aslot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" :: "v"(rv0), "v"(rv1), "v"(rv2), "v"(rv3) : "memory");
#ifdef G3_TIMING
    const unsigned long long T2 = __builtin_readcyclecounter();
#endif

    // ---- epilogue.  D layout of a 32x32 block: column (pixel) = lane & 31, row (channel) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  Stored
    // as it lies that is 256 four-byte stores per lane (measured: 73 thousand cycles, more than the loop); each block goes through the
    // wave's OWN first B buffer (32 rows x 32 floats = its 4 KB; no other wave touches it) and leaves as 16-byte stores, 128 contiguous
    // bytes per channel row.
    float* outb = P.out + (size_t)bz * P.obs;
    float* out2b = P.out2 ? P.out2 + (size_t)bz * P.o2bs : nullptr;
    const bool relu = P.mode == RPE_CONV_RELU;
    float* blk = (float*)&smem[B_AT + wv * 4096];
    const float* sbias = (const float*)&smem[BIAS_AT];
    const int e_quad = lane & 7, e_row = lane >> 3;                                   // read side: lane -> (row e_row + 8 i, pixel quad)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) blk[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[cb][pb][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // (the wave's own writes)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = e_row + 8 * i, co = co0 + 32 * cb + row, p = p0 + 32 * pb + 4 * e_quad;
                f32x4 v = *(const f32x4*)&blk[row * 32 + 4 * e_quad];
                v += sbias[32 * cb + row];
                if (relu) { v[0] = v[0] < 0.0f ? 0.0f : v[0]; v[1] = v[1] < 0.0f ? 0.0f : v[1]; v[2] = v[2] < 0.0f ? 0.0f : v[2]; v[3] = v[3] < 0.0f ? 0.0f : v[3]; }
                if (co < P.cout && p + 3 < hw) {
                    *(f32x4*)(outb + (size_t)co * hw + p) = v;
                    if (out2b) *(f32x4*)(out2b + (size_t)co * hw + p) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // (read before the next block overwrites it)
        }
#ifdef G3_TIMING
    if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == gridDim.z / 2 && tid == 0) {
        g_g3_timing[0] = T1 - T0; g_g3_timing[1] = T2 - T1; g_g3_timing[2] = __builtin_readcyclecounter() - T2; g_g3_timing[3] = nsteps;
    }
#endif
}

// weight (cout, cin) -> the three round-to-nearest bf16 parts, laid out [step = ci/16][co tile = co/128][cb = (co%128)/32][plane][k half = (ci%16)/8][co%32][ci%8]
__global__ void k_conv1x1_pack_x3(const float* __restrict__ w, unsigned short* __restrict__ wp, int cout, int cin, int coP, long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;              // over [step][co tile][cb][kh][co32][ci8]
    if (e >= total) return;
    const int ci8 = (int)(e & 7), co32 = (int)((e >> 3) & 31), kh = (int)((e >> 8) & 1), cb = (int)((e >> 9) & 3);
    const long long rest = e >> 11;
    const int nct = coP / G3_CO;
    const int co = (int)(rest % nct) * G3_CO + cb * 32 + co32, ci = (int)(rest / nct) * G3K + kh * 8 + ci8;
    const float v = (co < cout && ci < cin) ? w[(size_t)co * cin + ci] : 0.0f;
    auto bf16_rne = [](float f) { unsigned b = __builtin_bit_cast(unsigned, f); b += 0x7FFFu + ((b >> 16) & 1u); return b & 0xFFFF0000u; };
    const unsigned u = bf16_rne(v);
    const float r1 = v - __builtin_bit_cast(float, u);
    const unsigned u1 = bf16_rne(r1);
    const float r2 = r1 - __builtin_bit_cast(float, u1);
    unsigned short* d = wp + (rest * 4 + cb) * (3 * 512) + kh * 256 + co32 * 8 + ci8;
    d[0] = (unsigned short)(u >> 16); d[512] = (unsigned short)(u1 >> 16); d[1024] = (unsigned short)(bf16_rne(r2) >> 16);
}

static inline int g3_cop(int cout) { return (cout + G3_CO - 1) / G3_CO * G3_CO; }

extern "C" size_t rpe_conv1x1_x3_packed_bytes(int cout, int cin) {
    if (cout <= 0 || cin <= 0) return 0;
    return (size_t)((cin + G3K - 1) / G3K) * (g3_cop(cout) / G3_CO) * G3_A_STEP;
}

extern "C" int rpe_conv1x1_x3_pack(const float* weight, int cout, int cin, void* packed, void* stream) {
    if (!weight || !packed || cout <= 0 || cin <= 0) return RPE_E_BADARG;
    const long long total = (long long)((cin + G3K - 1) / G3K * G3K) * g3_cop(cout);
    hipLaunchKernelGGL(k_conv1x1_pack_x3, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, (unsigned short*)packed, cout, cin, g3_cop(cout), total);
    return rpe_check_launch();
}

extern "C" int rpe_conv1x1_x3(const rpe_conv_desc* d, void* stream) {
    if (!d || !d->x || !d->packed || !d->out || d->b <= 0 || d->cin <= 0 || d->cout <= 0 || d->h <= 0 || d->w <= 0) return RPE_E_BADARG;
    if (d->kh != 1 || d->kw != 1 || (d->stride != 0 && d->stride != 1)) return RPE_E_UNSUPPORTED;
    if (d->mode != RPE_CONV_LINEAR && d->mode != RPE_CONV_RELU) return RPE_E_UNSUPPORTED;
    if (d->add || d->hidden || d->zgate || d->scale || d->residual || d->stats || d->pre_norm) return RPE_E_UNSUPPORTED;
    const long long hw = (long long)d->h * d->w;
    if ((hw & 3) || hw < 4 || (((uintptr_t)d->x) & 15) || (d->x_batch_stride & 3) || (((uintptr_t)d->packed) & 15)) return RPE_E_UNSUPPORTED;
    // the epilogue stores 16-byte pieces: destination slices aligned like the input
    if ((((uintptr_t)d->out) & 15) || (d->out_batch_stride & 3) || (d->out2 && ((((uintptr_t)d->out2) & 15) || (d->out2_batch_stride & 3)))) return RPE_E_UNSUPPORTED;
    // 32-bit byte offsets: inside a group of four channel planes, or -- a ragged last step (cin % 16 != 0) addresses up to 16 planes from one base -- sixteen
    if (hw * 4 * ((d->cin & (G3K - 1)) ? 16 : 4) >= (1ll << 31)) return RPE_E_UNSUPPORTED;
    G1X3P P;
    P.x = d->x; P.xbs = d->x_batch_stride; P.wp = (const unsigned short*)d->packed; P.cin = d->cin; P.cout = d->cout; P.coP = g3_cop(d->cout);
    P.hw = (int)hw; P.bias = d->bias; P.out = d->out; P.obs = d->out_batch_stride; P.out2 = d->out2; P.o2bs = d->out2_batch_stride; P.mode = d->mode;
    hipLaunchKernelGGL(k_conv1x1_x3, dim3(ceil_div((int)hw, 256), P.coP / G3_CO, d->b), dim3(256), 0, (hipStream_t)stream, P);
    return rpe_check_launch();
}
