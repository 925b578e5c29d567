// Pose layer on gfx950: fused residual / gradient / Gauss-Newton-Hessian reduction over dense
// correspondences and the on-device L-BFGS / GN iteration loop.
//
// Replaces (reference paths): core/pose/pose_head.py:12-79 (reprojection_objective, depth_objective,
// objective, solve), core/geometry/pinhole_transforms.py:28-30,39-42,90-99 (transform, [I|-[X]x], project),
// torch.optim.LBFGS.step (line_search_fn=None) and declerative_node_lie.py:233-234 (vec7/log6 outputs).
//
// Kernel plan per objective evaluation (HBM-bound, 42 B/pixel algorithmic):
//   k_pose_reduce : grid (nblk, n).  Each thread streams 4 pixels per step with 16-byte loads from the ten
//                   f32 planes + two u8 planes, does all arithmetic in f64 (as the reference), keeps 8 (+21
//                   with the Hessian) f64 accumulators, then 64-lane butterfly -> LDS -> one partial row
//                   per block.  No atomics: the result is bit-reproducible run to run.
//   its tail      : the LAST workgroup of a row to finish (an atomic ticket per row decides who that is -- nothing else: the
//                   partials are summed in a fixed order whoever runs the tail) sums the block partials, clips the gradient, runs
//                   one L-BFGS (or GN) iteration in R^6, the left retraction T <- exp(t d) T and the stopping tests, and writes
//                   the row's outputs once it has stopped.  State lives in the caller's workspace, so the whole N-iteration
//                   solve is N + 1 launches (k_pose_init + N x k_pose_reduce) and zero host syncs.  (Rounds 1-5: a second
//                   one-workgroup-per-row kernel, k_pose_update, behind every reduction, and k_pose_finalize: 2N + 2 launches.)
#include "rpe_common.h"
#include <cstddef>
#include <cstdlib>
#include "se3_device.h"

#define NPART 32          // doubles per partial row: loss2d, loss3d, g[6], H[21], pad
#define HIST 100          // torch.optim.LBFGS history_size default
#define RED_THREADS 256
// a row's ticket and epoch words sit in their own 128-byte lines, 256 bytes from the next row's: 48 workgroups poll a row's epoch, and all
// rows' words in one line put every poll, ticket and epoch store of the launch through one memory channel
#define SYNC_STRIDE 64
#define SYNC_EPOCH 32

struct RowState {
    double T[7];
    double g[6];          // clipped gradient of the latest evaluation
    double prev_g[6];
    double d[6];
    double t, loss, prev_loss, H_diag;
    int n_iter, evals, stop, num_old;
    double old_dirs[HIST][6];
    double old_stps[HIST][6];
    double ro[HIST];
};

// Wave-uniform per-row constants, pre-converted to f64 so the reduce kernel can keep them in SGPRs
// (scalar loads) instead of spending 46 VGPRs per lane on them.
struct RowUniform {
    double R[9];
    double t[3];
    double K[9];
    double c2, c3;      // loss_weight[1]/(hw*hw), loss_weight[0]/hw
    double pad;
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static int pose_nblk(int n, int h, int w, bool hess = false) {
    int64_t hw = (int64_t)h * w;
    int64_t quads = (hw + 3) / 4;
    // >= 2 quads per thread, and ONE resident round of workgroups chip-wide: 256 CUs x 3 workgroups (2 with the Hessian's 29
    // accumulators: __launch_bounds__ below).  Measured at n = 16, 640x512 (8-iteration solve): 1024 blocks 420 / 664 us (L-BFGS / GN),
    // 768 413 / 659, 512 419 / 620, 2048 451 / 769 -- a second, partly filled round costs more than the shorter threads gain.
    int64_t per_row = (quads + RED_THREADS * 2 - 1) / (RED_THREADS * 2);
    int64_t want = ((hess ? 512 : 768) + n - 1) / n;
    int64_t nblk = per_row < want ? per_row : want;
    if (nblk < 1) nblk = 1;
    int64_t maxblk = (quads + RED_THREADS - 1) / RED_THREADS;
    if (nblk > maxblk) nblk = maxblk;
    if (nblk > 2048) nblk = 2048;
    return (int)nblk;
}

extern "C" size_t rpe_pose_workspace_bytes(int n, int h, int w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    size_t st = align_up(sizeof(RowState) * (size_t)n, 256) + align_up(sizeof(RowUniform) * (size_t)n, 256);
    size_t pa = align_up(sizeof(double) * NPART * (size_t)pose_nblk(1, h, w) * n, 256);   // room for any partition_rows (n = 1 has the most blocks per
                                                                                          // row; the Hessian launch never has more)
    return st + pa + sizeof(int) * SYNC_STRIDE * (size_t)n + 256;                         // + one ticket counter and one epoch word per row
}

struct PoseArgs {
    const float* flow; const float* pcl1; const float* pcl2; const float* w1; const float* w2;
    const uint8_t* m1; const uint8_t* m2; const float* K; const float* lw;
    int n, h, w;
};

__device__ __forceinline__ int tri(int i, int j) {   // upper-triangle index, i <= j
    return i * 6 - (i * (i - 1)) / 2 + (j - i);
}

template <bool HESS>
__device__ __forceinline__ void pixel_terms(double* acc, double px, double py, double fl_x, double fl_y,
                                            const double p[3], const double q[3], double w1, double w2, bool m1, bool m2,
                                            const double R[9], const double t[3], const double K[9],
                                            double c2, double c3, double Wd, double Hd) {
    // X = R p + t   (pinhole_transforms.py:28-30)
    double X = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + t[0];
    double Y = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + t[1];
    double Z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + t[2];
    // ipts = K X ; depth = clamp(iz, 1e-12)   (pinhole_transforms.py:93-98)
    double ix = K[0] * X + K[1] * Y + K[2] * Z;
    double iy = K[3] * X + K[4] * Y + K[5] * Z;
    double iz = K[6] * X + K[7] * Y + K[8] * Z;
    double dep = iz < 1e-12 ? 1e-12 : iz;                 // NaN stays NaN, like torch.clamp
    double passz = iz >= 1e-12 ? 1.0 : 0.0;
    // Gauss-Newton (HESS; not on the reference's L-BFGS path, which keeps torch's five IEEE divisions bit for bit): ONE reciprocal --
    // the hardware's estimate + one Newton step, < 1 ulp -- serves the projection, its gradient and the Jacobian rows below
    double invd = 0.0;
    if (HESS) { const double r0 = __builtin_amdgcn_rcp(dep); invd = r0 == 0.0 ? r0 : fma(fma(-dep, r0, 1.0), r0, r0); }   // (1 / inf = 0, as the division gives)
    double u = HESS ? ix * invd : ix / dep, v = HESS ? iy * invd : iy / dep;
    double fx = px + fl_x, fy = py + fl_y;                // pose_head.py:19
    double ex = fx - u, ey = fy - v;
    double r2 = (ex * ex + ey * ey) * w1;                 // :21-22
    bool inimg = (fx > 0.0) && (fy > 0.0) && (fx < Wd) && (fy < Hd);   // :24
    bool bad = isinf(r2) || isnan(r2) || !inimg || !m1;   // :25
    double gate2 = bad ? 0.0 : 1.0;
    acc[0] += bad ? 0.0 : r2;                             // :28-29
    double ex3 = X - q[0], ey3 = Y - q[1], ez3 = Z - q[2];   // :41-43
    double r3 = (ex3 * ex3 + ey3 * ey3 + ez3 * ez3) * w2;
    bool ok3 = m1 && m2;                                  // :47
    double gate3 = ok3 ? 1.0 : 0.0;
    acc[1] += ok3 ? r3 : 0.0;
    // gradient, multiplied out the way autograd does (0 * nan = nan reaches the pose, as in the reference)
    double a2 = -2.0 * w1 * gate2 * c2;
    double gu = a2 * ex, gv = a2 * ey;
    double a3 = 2.0 * w2 * gate3 * c3;
    double gX, gY, gZ;
    double Ju[6], Jv[6];                                  // HESS: rows of d(u, v)/d(xi) = d(u, v)/dX [I | -[X]x]
    if (HESS) {
        // the same chain rule with the projection's Jacobian rows written out (they are needed for J^T J anyway)
        const double up = u * passz, vp = v * passz;
        Ju[0] = (K[0] - up * K[6]) * invd; Ju[1] = (K[1] - up * K[7]) * invd; Ju[2] = (K[2] - up * K[8]) * invd;
        Jv[0] = (K[3] - vp * K[6]) * invd; Jv[1] = (K[4] - vp * K[7]) * invd; Jv[2] = (K[5] - vp * K[8]) * invd;
        gX = gu * Ju[0] + gv * Jv[0] + a3 * ex3;
        gY = gu * Ju[1] + gv * Jv[1] + a3 * ey3;
        gZ = gu * Ju[2] + gv * Jv[2] + a3 * ez3;
    } else {
        double g_ix = gu / dep, g_iy = gv / dep;
        double g_iz = -(gu * ix + gv * iy) / (dep * dep) * passz;
        gX = K[0] * g_ix + K[3] * g_iy + K[6] * g_iz + a3 * ex3;
        gY = K[1] * g_ix + K[4] * g_iy + K[7] * g_iz + a3 * ey3;
        gZ = K[2] * g_ix + K[5] * g_iy + K[8] * g_iz + a3 * ez3;
    }
    acc[2] += gX; acc[3] += gY; acc[4] += gZ;            // [I | -[X]x]^T gX
    acc[5] += Y * gZ - Z * gY;
    acc[6] += Z * gX - X * gZ;
    acc[7] += X * gY - Y * gX;
    if (HESS) {
        double* Hh = acc + 8;
        double s2 = bad ? 0.0 : 2.0 * w1 * c2;
        double s3 = ok3 ? 2.0 * w2 * c3 : 0.0;
        // (Measured and NOT adopted: the 21 products as packed f32 FMAs with per-thread f32 partial sums -- 9 % faster, but the
        // iterates leave the 1e-9 band around the f64 oracle; a two-pixel vector width with three workgroups per CU (161 VGPRs) and
        // scheduling barriers every other pixel or none -- no change: PMC puts the kernel's vector pipe at 60 % busy with ~280 f64
        // instructions per pixel, full rate on CDNA4, i.e. the floor of this formulation is ~38 us per evaluation of 16 frames.)
        // J = A P with A = d(u, v)/dX (2 x 3; for the 3-D term the identity) and P = [I | -[X]x] (3 x 6), so
        //   H += P^T M P,  M = s2 A^T A + s3 I   (3 x 3 symmetric: 6 entries)
        // = [[M, B], [B^T, C]] with B = -M [X]x (rows m_i x X) and C = [X]x B: 66 fused operations per pixel instead of the ~95 of the
        // 21 independent 6-vector products + the 3-D term's own block
        // a masked-out term contributes exactly nothing, whatever its Jacobian holds (selects, not products with zero: 0 * inf = nan)
        if (s2 == 0.0 && s3 == 0.0) return;
        const bool on2 = s2 != 0.0;
        const double au0 = on2 ? Ju[0] : 0.0, au1 = on2 ? Ju[1] : 0.0, au2 = on2 ? Ju[2] : 0.0;
        const double av0 = on2 ? Jv[0] : 0.0, av1 = on2 ? Jv[1] : 0.0, av2 = on2 ? Jv[2] : 0.0;
        const double su0 = s2 * au0, su1 = s2 * au1, su2 = s2 * au2, sv0 = s2 * av0, sv1 = s2 * av1, sv2 = s2 * av2;
        const double m00 = fma(su0, au0, fma(sv0, av0, s3)), m01 = fma(su0, au1, sv0 * av1), m02 = fma(su0, au2, sv0 * av2);
        const double m11 = fma(su1, au1, fma(sv1, av1, s3)), m12 = fma(su1, au2, sv1 * av2);
        const double m22 = fma(su2, au2, fma(sv2, av2, s3));
        Hh[tri(0, 0)] += m00; Hh[tri(0, 1)] += m01; Hh[tri(0, 2)] += m02; Hh[tri(1, 1)] += m11; Hh[tri(1, 2)] += m12; Hh[tri(2, 2)] += m22;
        // B[i][.] = (m_i2 Y - m_i1 Z, m_i0 Z - m_i2 X, m_i1 X - m_i0 Y), m_i = row i of M
        const double b00 = fma(m02, Y, -(m01 * Z)), b01 = fma(m00, Z, -(m02 * X)), b02 = fma(m01, X, -(m00 * Y));
        const double b10 = fma(m12, Y, -(m11 * Z)), b11 = fma(m01, Z, -(m12 * X)), b12 = fma(m11, X, -(m01 * Y));
        const double b20 = fma(m22, Y, -(m12 * Z)), b21 = fma(m02, Z, -(m22 * X)), b22 = fma(m12, X, -(m02 * Y));
        Hh[tri(0, 3)] += b00; Hh[tri(0, 4)] += b01; Hh[tri(0, 5)] += b02;
        Hh[tri(1, 3)] += b10; Hh[tri(1, 4)] += b11; Hh[tri(1, 5)] += b12;
        Hh[tri(2, 3)] += b20; Hh[tri(2, 4)] += b21; Hh[tri(2, 5)] += b22;
        // C = [X]x B, [X]x = [[0, -Z, Y], [Z, 0, -X], [-Y, X, 0]] (upper triangle)
        Hh[tri(3, 3)] += fma(Y, b20, -(Z * b10)); Hh[tri(3, 4)] += fma(Y, b21, -(Z * b11)); Hh[tri(3, 5)] += fma(Y, b22, -(Z * b12));
        Hh[tri(4, 4)] += fma(Z, b01, -(X * b21)); Hh[tri(4, 5)] += fma(Z, b02, -(X * b22));
        Hh[tri(5, 5)] += fma(X, b12, -(Y * b02));
    }
}

#ifdef RPE_POSE_PROBE
// diagnostic build only (tools/probe_pose_tail.sh): 100 MHz wall-clock stamps of the last tail of row 0
// per evaluation e < 32 of row 0: [8e+0] workgroup 0 leaves the poll, [8e+1..4] the tail's phases, [8e+5] the LAST workgroup leaves the pixel
// loop, [8e+6] epoch words stored, [8e+7] workgroup 0 leaves the pixel loop
__device__ long long g_pose_probe[256];
__device__ int g_pose_probe_eval;
extern "C" int rpe_pose_probe_read(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pose_probe), sizeof(long long) * 256) == hipSuccess ? 0 : -2; }
#define PROBE(i) do { if (row == 0 && threadIdx.x == 0 && g_pose_probe_eval < 32) g_pose_probe[8 * g_pose_probe_eval + (i)] = wall_clock64(); } while (0)
#else
#define PROBE(i) do { } while (0)
#endif

struct SolveOpts { double tol_grad, tol_change; int history; };
// what the tail of a solve's reduction needs (rpe_pose_reduce's stand-alone launches have no tail)
struct TailArgs {
    int* tickets;                 // SYNC_STRIDE ints per row: [0] the ticket counter (zero between evaluations), [SYNC_EPOCH] the epoch word
    int evals;                    // PERSIST: evaluations this launch runs
    int mode, max_iter;
    SolveOpts opt;
    double* T_out; float* vec7; float* log6; int32_t* info;
};
template <bool COH>
__device__ bool reduce_tail(const PoseArgs& A, RowUniform* uni, RowState* states, const double* partials, int row, int nblk, const TailArgs& Z, double* line);

#define EPOCH_STOPPED (1 << 30)
// a workgroup-uniform double from LDS, moved to scalar registers
__device__ __forceinline__ double lds_uniform(const double* p) {
    const double v = *p;
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// Waiting never deadlocks: a workgroup only ever waits for workgroups of ITS OWN ROW, a row's workgroups are contiguous in dispatch order,
// and a row whose workgroups are all resident runs to its end and leaves -- so whatever else shares the device (another solve on another
// stream), some row is always complete and progressing, and the slots it frees go to the rows that are not.  (The host still only asks
// for one launch when the whole grid fits: the partition is sized so that nothing waits for a slot.)
// TAIL: the launch belongs to a solve: the row's last workgroup runs the update, and the launch runs Z.evals evaluations -- between
// evaluations a row's workgroups wait for the row's tail (an epoch word per row), so there is no launch boundary, no dispatch of 768
// workgroups and no drained chip between them.  Z.evals > 1 needs every workgroup of the grid resident at once (the host checks; else it
// launches this SAME kernel once per evaluation with Z.evals = 1: one instantiation, hence one set of floating-point contractions and
// bit-identical sums, whichever way a solve is launched -- chunked tracking relies on that).
template <bool HESS, int VEC, bool TAIL>
__global__ __launch_bounds__(RED_THREADS, HESS ? 2 : 3) void k_pose_reduce(PoseArgs A, RowUniform* uni, RowState* states,
                                                                             double* __restrict__ partials, TailArgs Z) {
    constexpr bool PERSIST = TAIL;
    constexpr int NACC = HESS ? 29 : 8;
    const int row = blockIdx.y;
    const int nblk = gridDim.x;
    double* prow = partials + ((size_t)row * nblk + blockIdx.x) * NPART;
    if (states && states[row].stop != 0) return;          // finished rows cost nothing (every workgroup of the row sees the same flag: no ticket is drawn)
    const RowUniform& U = uni[row];
    double R[9], t[3], K[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { R[i] = U.R[i]; K[i] = U.K[i]; }
    t[0] = U.t[0]; t[1] = U.t[1]; t[2] = U.t[2];
    const double c2 = U.c2, c3 = U.c3;
    const double Wd = (double)A.w, Hd = (double)A.h;
    __shared__ double red[RED_THREADS / RPE_WAVE][NPART];
    __shared__ int is_last, epoch_seen;
    int* const ticket = Z.tickets + (size_t)row * SYNC_STRIDE;
    double* const line = (double*)(ticket + SYNC_EPOCH);      // the row's pose line: [epoch | R0..R6 || epoch | R7 R8 t0 t1 t2 | - -], 128-byte aligned

#pragma nounroll
    for (int it = 0; it < (PERSIST ? Z.evals : 1); ++it) {
        // PERSIST: the plane pointers and the per-thread offsets are recomputed per evaluation from two values the compiler cannot see
        // through -- hoisted out of the evaluation loop (every address of every pass of the pixel loop) they cost 150 spilled registers
        int64_t hw = (int64_t)A.h * A.w;
        unsigned tid = threadIdx.x;
        if constexpr (PERSIST) { asm volatile("" : "+s"(hw)); asm volatile("" : "+v"(tid)); }
        const float* flx = A.flow + (size_t)row * 2 * hw; const float* fly = flx + hw;
        const float* p1 = A.pcl1 + (size_t)row * 3 * hw;
        const float* p2 = A.pcl2 + (size_t)row * 3 * hw;
        const float* w1 = A.w1 + (size_t)row * hw; const float* w2 = A.w2 + (size_t)row * hw;
        const uint8_t* m1 = A.m1 + (size_t)row * hw; const uint8_t* m2 = A.m2 + (size_t)row * hw;
        struct Quad { float f0[VEC], f1[VEC], a0[VEC], a1[VEC], a2[VEC], b0[VEC], b1[VEC], b2[VEC], ww1[VEC], ww2[VEC]; uint8_t mm1[VEC], mm2[VEC]; };
        auto load_quad = [&](int64_t vq, Quad& Q) {
            const int64_t base = vq * VEC;
            if constexpr (VEC == 4) {
                *(float4*)Q.f0 = *(const float4*)(flx + base); *(float4*)Q.f1 = *(const float4*)(fly + base);
                *(float4*)Q.a0 = *(const float4*)(p1 + base); *(float4*)Q.a1 = *(const float4*)(p1 + hw + base);
                *(float4*)Q.a2 = *(const float4*)(p1 + 2 * hw + base);
                *(float4*)Q.b0 = *(const float4*)(p2 + base); *(float4*)Q.b1 = *(const float4*)(p2 + hw + base);
                *(float4*)Q.b2 = *(const float4*)(p2 + 2 * hw + base);
                *(float4*)Q.ww1 = *(const float4*)(w1 + base); *(float4*)Q.ww2 = *(const float4*)(w2 + base);
                *(uint32_t*)Q.mm1 = *(const uint32_t*)(m1 + base); *(uint32_t*)Q.mm2 = *(const uint32_t*)(m2 + base);
            } else {
                Q.f0[0] = flx[base]; Q.f1[0] = fly[base];
                Q.a0[0] = p1[base]; Q.a1[0] = p1[hw + base]; Q.a2[0] = p1[2 * hw + base];
                Q.b0[0] = p2[base]; Q.b1[0] = p2[hw + base]; Q.b2[0] = p2[2 * hw + base];
                Q.ww1[0] = w1[base]; Q.ww2[0] = w2[base]; Q.mm1[0] = m1[base]; Q.mm2[0] = m2[base];
            }
        };
        const int64_t nvec = (hw + VEC - 1) / VEC;
        int64_t vq = (int64_t)blockIdx.x * RED_THREADS + tid;
        // (Measured and NOT adopted: the first pass's loads issued before the wait for the row's tail -- the 42 registers held across the
        // poll spill, and the solve went from 406 to 459 us.)
        if (PERSIST && it > 0) {
            // The row's tail of evaluation it - 1 publishes the new pose in the row's 128-byte pose line: each 64-byte half carries its own
            // copy of the epoch word in front of its part of (R, t), written after the data has been acknowledged -- a half that shows
            // epoch >= it shows the new data (one memory transaction per half).  Wave 0 polls the line with one 16-lane load: the pose
            // arrives WITH the epoch, not a round trip later.
            if (threadIdx.x < 64) {
                double v = 0.0;
                // (bounded: ~4 M polls = seconds.  It never gets there -- see "Waiting never deadlocks" above --, but a kernel that can spin
                // for ever takes the whole device with it if an assumption about the dispatcher ever fails; past the bound the workgroup goes on
                // with what it has read, and the solve ends with a wrong pose instead of not at all)
                for (int spin = 0; spin < (1 << 22); ++spin) {
                    if (threadIdx.x < 16) v = __hip_atomic_load(line + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int ea = __builtin_amdgcn_readlane(__double2loint(v), 0), eb = __builtin_amdgcn_readlane(__double2loint(v), 8);
                    if (threadIdx.x == 0) epoch_seen = ea;
                    if ((ea & ~EPOCH_STOPPED) >= it && (eb & ~EPOCH_STOPPED) >= it) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                // lanes 1..7: R[0..6], lanes 9..13: R[7], R[8], t[0..2]
                if ((threadIdx.x >= 1 && threadIdx.x < 8) || (threadIdx.x >= 9 && threadIdx.x < 14)) red[0][threadIdx.x < 8 ? threadIdx.x - 1 : threadIdx.x - 2] = v;
            }
            __syncthreads();
            if (epoch_seen & EPOCH_STOPPED) return;
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = lds_uniform(&red[0][i]);
#pragma unroll
            for (int i = 0; i < 3; ++i) t[i] = lds_uniform(&red[0][9 + i]);
            __syncthreads();                                  // (red is the reduction's scratch again below)
        }
#ifdef RPE_POSE_PROBE
        if (TAIL && row == 0 && threadIdx.x == 0) { if (blockIdx.x == 0) { g_pose_probe_eval = it; g_pose_probe[8 * it + 0] = wall_clock64(); } }
#endif
        double acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
        auto compute_quad = [&](int64_t vq_, const Quad& Q) {
            const int64_t base = vq_ * VEC;
            const int y = (int)(base / A.w);
            const int x = (int)(base - (int64_t)y * A.w);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                double p[3] = {(double)Q.a0[k], (double)Q.a1[k], (double)Q.a2[k]};
                double q[3] = {(double)Q.b0[k], (double)Q.b1[k], (double)Q.b2[k]};
                pixel_terms<HESS>(acc, (double)(x + k) + 0.5, (double)y + 0.5, (double)Q.f0[k], (double)Q.f1[k], p, q,
                                  (double)Q.ww1[k], (double)Q.ww2[k], Q.mm1[k] != 0, Q.mm2[k] != 0, R, t, K, c2, c3, Wd, Hd);
                if (VEC > 1) __builtin_amdgcn_sched_barrier(0);   // one pixel at a time: bounds f64 live ranges
            }
        };
        for (; vq < nvec; vq += (int64_t)nblk * RED_THREADS) {
            Quad Q;
            load_quad(vq, Q);
            compute_quad(vq, Q);
        }

#ifdef RPE_POSE_PROBE
        if (TAIL && row == 0 && threadIdx.x == 0 && it < 32) { atomicMax((unsigned long long*)&g_pose_probe[8 * it + 5], (unsigned long long)wall_clock64());
                                                                 if (blockIdx.x == 0) g_pose_probe[8 * it + 7] = wall_clock64(); }
#endif
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            double s = wave_sum(acc[i]);
            if (lane == 0) red[wv][i] = s;
        }
        __syncthreads();
        if (threadIdx.x < NPART) {
            double s = 0.0;
            if (threadIdx.x < NACC) s = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
            // TAIL: the partial row is read by ANOTHER workgroup of this launch (possibly on another XCD, behind another L2): a device-scope
            // store (written through) here and device-scope loads there.  A device-scope FENCE instead would write back and invalidate the
            // whole L2 of this XCD once per workgroup: measured 141 us per evaluation instead of 51.
            if constexpr (TAIL) __hip_atomic_store(prow + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else prow[threadIdx.x] = s;
        }
        if constexpr (TAIL) {
            // the row's last workgroup to get here runs the update.  The barrier waits for this workgroup's stores above (they have reached
            // the device's point of coherence when they are acknowledged); then one ticket per workgroup.
            __syncthreads();
            if (threadIdx.x == 0) is_last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1;
            __syncthreads();
            if (!is_last) {
                if constexpr (PERSIST) continue; else return;
            }
            if (threadIdx.x == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next evaluation
            const bool stopped = reduce_tail<PERSIST>(A, uni, states, partials, row, nblk, Z, line);
            if constexpr (PERSIST) {
                // the tail's device-scope stores (pose line data, state) are acknowledged when this barrier lets the workgroup through; then
                // the two epoch words of the pose line
                __syncthreads();
                if (threadIdx.x < 2) __hip_atomic_store((long long*)line + 8 * threadIdx.x, (long long)((it + 1) | (stopped ? EPOCH_STOPPED : 0)), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
#ifdef RPE_POSE_PROBE
                if (row == 0 && threadIdx.x == 0 && it < 32) g_pose_probe[8 * it + 6] = wall_clock64();
#endif
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- update
__device__ __forceinline__ void write_rt(RowUniform& U, const double* Tp) {
    double qx = Tp[3], qy = Tp[4], qz = Tp[5], qw = Tp[6];
    U.R[0] = 1.0 - 2.0 * (qy * qy + qz * qz); U.R[1] = 2.0 * (qx * qy - qz * qw); U.R[2] = 2.0 * (qx * qz + qy * qw);
    U.R[3] = 2.0 * (qx * qy + qz * qw); U.R[4] = 1.0 - 2.0 * (qx * qx + qz * qz); U.R[5] = 2.0 * (qy * qz - qx * qw);
    U.R[6] = 2.0 * (qx * qz - qy * qw); U.R[7] = 2.0 * (qy * qz + qx * qw); U.R[8] = 1.0 - 2.0 * (qx * qx + qy * qy);
    U.t[0] = Tp[0]; U.t[1] = Tp[1]; U.t[2] = Tp[2];
}
__device__ __forceinline__ void write_consts(RowUniform& U, const float* K, const float* lw, int row, int h, int w) {
    for (int i = 0; i < 9; ++i) U.K[i] = (double)K[(size_t)row * 9 + i];
    const double hwd = (double)h * (double)w;
    U.c2 = (double)lw[row * 2 + 1] / hwd / hwd;     // mean then /(h*w)  (pose_head.py:29)
    U.c3 = (double)lw[row * 2 + 0] / hwd;           // mean              (pose_head.py:51)
    U.pad = 0.0;
}
// uniforms for rpe_pose_reduce's explicit poses
__global__ void k_pose_prep(RowUniform* uni, const double* T, const float* K, const float* lw, int n, int h, int w) {
    int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    write_rt(uni[row], T + (size_t)row * 7);
    write_consts(uni[row], K, lw, row, h, w);
}

__device__ __forceinline__ double absmax6(const double* v) {   // torch .abs().max(): NaN propagates
    double m = 0.0;
    for (int i = 0; i < 6; ++i) { double a = fabs(v[i]); if (a > m || isnan(a)) m = a; if (isnan(m)) return m; }
    return m;
}
__device__ __forceinline__ double dot6(const double* a, const double* b) {
    double s = 0.0;
    for (int i = 0; i < 6; ++i) s += a[i] * b[i];
    return s;
}

__device__ void apply_step(RowState& S, RowUniform& U, const double* dir, double tstep) {
    // LieGroupParameter.add_: group <- exp(alpha * update) * group
    V3<double> tau = v3<double>(tstep * dir[0], tstep * dir[1], tstep * dir[2]);
    V3<double> phi = v3<double>(tstep * dir[3], tstep * dir[4], tstep * dir[5]);
    Pose<double> E = se3_exp(tau, phi);
    Pose<double> Tn = se3_mul(E, pose_load(S.T));
    pose_store(S.T, Tn);
    write_rt(U, S.T);
}

// Sums the block partials of one row in a fixed order into vals[0..NPART) (LDS): 256 threads = 8 parts x 32 values; part p takes
// blocks p, p+8, ... with four loads in flight (one lane walking the 160 rows of a 640x512 frame serially cost 24 us of
// dependent L2 round trips per launch), then the parts are added in the order 0..7.  Ends with a barrier.
#define UPD_THREADS 256
// COHERENT: the partial rows were written by other workgroups of the SAME launch (the reduction's tail): device-scope loads, which no
// cache of this CU can answer with a stale line.
template <bool COHERENT>
__device__ __forceinline__ double ld_partial(const double* p) {
    if constexpr (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
struct Nothing { __device__ __forceinline__ void operator()() const {} };
// ``between`` runs once, after the first eight loads have been requested and before any of them is used (the tail stores the row state it
// requested earlier into LDS there: all of the tail's loads are then one round trip).
template <bool COHERENT, typename F = Nothing>
__device__ __forceinline__ void sum_partials(const double* partials, int row, int nblk, int tid, double* vals, double (*red)[NPART], F between = F()) {
    const int j = tid & 31, part = tid >> 5;
    const double* p = partials + (size_t)row * nblk * NPART + j;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    // part p takes blocks p, p + 8, ...: rounds of four go to s0..s3, the rest to s0 -- the ORDER is the contract (it fixes the bits); the
    // loads of up to eight blocks (two rounds, or a round and the rest) are issued together: a device-scope load is a ~1.2 us round trip
    // to the memory side, and three dependent trips were most of the tail's 5 us "loads + sum"
    int b = part;
    do {                                                            // (at least once: ``between`` must run in every thread)
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = b + 8 * i < nblk ? ld_partial<COHERENT>(p + (size_t)(b + 8 * i) * NPART) : 0.0;
        if (b == part) { __builtin_amdgcn_sched_barrier(0); between(); }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int bb = b + 32 * r;
            if (bb + 24 < nblk) { s0 += v[4 * r]; s1 += v[4 * r + 1]; s2 += v[4 * r + 2]; s3 += v[4 * r + 3]; }
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (bb + 8 * i < nblk) s0 += v[4 * r + i];
            }
        }
        b += 64;
    } while (b < nblk);
    red[part][j] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (tid < NPART) {
        double t = 0.0;
        for (int q = 0; q < UPD_THREADS / 32; ++q) t += red[q][tid];
        vals[tid] = t;
    }
    __syncthreads();
}

// EDITING THE TAIL: it is inlined into k_pose_reduce, whose pixel loop sits at 168 of the 170 registers three workgroups per CU allow -- the
// register allocation of the LOOP moves with the tail's code.  Measured: the GN branch's 27 divisions written as six reciprocals (a tail-only
// change) took the L-BFGS solve from 392 to 462 us; `noinline` on the tail does not compile ("illegal VGPR to SGPR copy").  After any edit
// below: tools/bench_pose_solve.py, both solver modes.
// One lane runs the iteration logic; every access it makes to the row's state used to be a dependent L2 round trip (28 us per
// launch at batch 1).  The workgroup stages the state's head (poses, gradients, counters: 31 doubles), the (y, s, rho) history
// and the row's rotation in LDS, lane 0 works there, and the workgroup writes back what changed.
__device__ void pose_update_row(RowState& S, RowUniform& U, double* h_al, double (*Lc)[6], int* wb, const double* vals, const float* lw, int row,
                                int h, int w, int mode, int max_iter, SolveOpts opt);

__device__ __forceinline__ void finalize_row(const RowState& S, int row, double* T_out, float* vec7, float* log6, int32_t* info) {
    for (int i = 0; i < 7; ++i) {
        T_out[row * 7 + i] = S.T[i];
        if (vec7) vec7[row * 7 + i] = (float)S.T[i];       // out.group.vec().float()
    }
    if (log6) {                                           // out.log().float()
        V3<double> tau, phi;
        se3_log(pose_load(S.T), tau, phi);
        float* o = log6 + row * 6;
        o[0] = (float)tau.x; o[1] = (float)tau.y; o[2] = (float)tau.z;
        o[3] = (float)phi.x; o[4] = (float)phi.y; o[5] = (float)phi.z;
    }
    if (info) { info[row * 4 + 0] = S.n_iter; info[row * 4 + 1] = S.evals; info[row * 4 + 2] = S.stop; info[row * 4 + 3] = 0; }
}

// COH (the persistent launch): the row's state was last written by the tail of the previous evaluation, which may have run on another XCD
// behind another L2 -- every access to it is a device-scope load / store (written through, never answered by a stale line).  A fence pair
// per tail instead (buffer_inv + buffer_wbl2 of the whole L2) was measured at +55 us per evaluation.
template <bool COH>
__device__ __forceinline__ double ldd(const double* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH>
__device__ __forceinline__ void std_(double* p, double v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// returns (to every thread) whether the row has stopped
template <bool COH>
__device__ bool reduce_tail(const PoseArgs& A, RowUniform* uni, RowState* states, const double* partials, int row, int nblk, const TailArgs& Z, double* line) {
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));       // the tail's addresses are computed HERE: hoisted in front of the evaluation loop they are spilled, and every
    //                                       reload of one is a wait for all outstanding loads (the tail's loads then go out one round trip at a time)
    PROBE(1);
    RowState& G = states[row];
    constexpr int HEAD = 31;                                  // doubles in front of old_dirs (T, g, prev_g, d, t, loss, prev_loss, H_diag, 4 ints)
    static_assert(offsetof(RowState, old_dirs) == HEAD * sizeof(double), "RowState head");
    __shared__ double vals[NPART], red[UPD_THREADS / 32][NPART];
    __shared__ RowState S;
    __shared__ RowUniform U;
    __shared__ double h_al[HIST], Lc[6][6];
    __shared__ int wb[2];                                     // history entries [wb[0], wb[1]) changed
    // Every load of the tail is requested before the first one is used: state head, row constants, the (y, s, rho) history and -- inside
    // sum_partials -- the first eight partial rows share ONE round trip to the memory side (device-scope loads are not answered by a cache:
    // ~1.2 us each; written as load -> LDS store pairs the compiler waited for each before requesting the next: six round trips)
    constexpr int NU = (int)(sizeof(RowUniform) / sizeof(double));
    // (unconditional loads at clamped indices: a load under a lane condition is its own basic block, and the compiler waits for every
    // outstanding load at each of those)
    const double v_head = ldd<COH>((const double*)&G + (lane < HEAD ? lane : HEAD - 1));
    const double v_uni = ldd<COH>((const double*)&uni[row] + (lane < NU ? lane : NU - 1));
    // the history is fetched without waiting for its length (a dependent round trip): it never holds more pairs than iterations
    const int nold = Z.mode != RPE_SOLVER_GN ? (Z.max_iter < Z.opt.history ? Z.max_iter : Z.opt.history) : 0;
    // (one element of each array per thread covers 42 pairs = every solve of up to 42 iterations; longer histories take the plain loop below)
    const int e0 = lane < HIST * 6 ? lane : HIST * 6 - 1;
    const double v_dir = ldd<COH>(&G.old_dirs[0][0] + e0), v_stp = ldd<COH>(&G.old_stps[0][0] + e0);
    const double v_ro = ldd<COH>(&G.ro[lane < HIST ? lane : HIST - 1]);
    sum_partials<true>(partials, row, nblk, lane, vals, red, [&]() {      // (ends with a barrier)
        if (lane < HEAD) ((double*)&S)[lane] = v_head;
        if (lane < NU) ((double*)&U)[lane] = v_uni;
        if (lane < nold * 6) { (&S.old_dirs[0][0])[lane] = v_dir; (&S.old_stps[0][0])[lane] = v_stp; }
        if (lane < nold) S.ro[lane] = v_ro;
        if (lane == 0) { wb[0] = 0; wb[1] = 0; }
    });
    if (nold * 6 > UPD_THREADS) {                                    // (histories of more than 42 pairs: the rest, after the fact)
        for (int e = lane + UPD_THREADS; e < nold * 6; e += UPD_THREADS) { (&S.old_dirs[0][0])[e] = ldd<COH>(&G.old_dirs[0][0] + e); (&S.old_stps[0][0])[e] = ldd<COH>(&G.old_stps[0][0] + e); }
        __syncthreads();
    }
    PROBE(2);
    if (lane == 0) pose_update_row(S, U, h_al, Lc, wb, vals, A.lw, row, A.h, A.w, Z.mode, Z.max_iter, Z.opt);
    PROBE(3);
    __syncthreads();
    if (lane < HEAD) std_<COH>((double*)&G + lane, ((const double*)&S)[lane]);
    if (lane < 12) std_<COH>((double*)&uni[row] + lane, ((const double*)&U)[lane]);          // R, t (write_rt)
    if (COH && lane < 12) std_<COH>(line + (lane < 7 ? 1 + lane : 2 + lane), ((const double*)&U)[lane]);   // ... and into the row's pose line (see the poll)
    for (int e = wb[0] * 6 + lane; e < wb[1] * 6; e += UPD_THREADS) { std_<COH>(&G.old_dirs[0][0] + e, (&S.old_dirs[0][0])[e]); std_<COH>(&G.old_stps[0][0] + e, (&S.old_stps[0][0])[e]); }
    for (int e = wb[0] + lane; e < wb[1]; e += UPD_THREADS) std_<COH>(&G.ro[e], S.ro[e]);
    // a row that has stopped is never touched again: its outputs are written here, once (DeclarativeFunctionLie.forward's vec7 / log6)
    const bool stopped = S.stop != 0;
    if (lane == 64 && stopped) finalize_row(S, row, Z.T_out, Z.vec7, Z.log6, Z.info);
    PROBE(4);
    return stopped;
}

__device__ void pose_update_row(RowState& S, RowUniform& U, double* h_al, double (*L)[6], int* wb, const double* vals, const float* lw, int row,
                                int h, int w, int mode, int max_iter, SolveOpts opt) {
    const double hwd = (double)h * (double)w;
    const double loss2d = vals[0] / hwd / hwd, loss3d = vals[1] / hwd;
    const double loss = (double)lw[row * 2 + 1] * loss2d + (double)lw[row * 2 + 0] * loss3d;
    double g[6];
    for (int i = 0; i < 6; ++i) g[i] = vals[2 + i];
    const double tol_grad = opt.tol_grad, tol_change = opt.tol_change, lr = 1.0;
    const int hist = opt.history;

    if (mode == RPE_SOLVER_GN) {
        S.n_iter += 1; S.evals += 1; S.loss = loss;
        for (int i = 0; i < 6; ++i) S.g[i] = g[i];
        // Cholesky H = L L^T on the 6x6 upper triangle (L in LDS: indexed by loop variables, it would otherwise live in scratch memory)
        bool ok = true;
        for (int i = 0; i < 6; ++i) ok = ok && isfinite(g[i]);
        for (int i = 0; i < 6 && ok; ++i) {
            for (int j = 0; j <= i; ++j) {
                double s = vals[8 + tri(j, i)];
                for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
                if (i == j) { if (!(s > 0.0) || !isfinite(s)) { ok = false; break; } L[i][i] = sqrt(s); }
                else L[i][j] = s / L[j][j];
            }
        }
        if (!ok) { S.stop = RPE_STOP_NOT_PD; return; }
        double yv[6], dl[6];
        for (int i = 0; i < 6; ++i) { double s = -g[i]; for (int k = 0; k < i; ++k) s -= L[i][k] * yv[k]; yv[i] = s / L[i][i]; }
        for (int i = 5; i >= 0; --i) { double s = yv[i]; for (int k = i + 1; k < 6; ++k) s -= L[k][i] * dl[k]; dl[i] = s / L[i][i]; }
        for (int i = 0; i < 6; ++i) S.d[i] = dl[i];
        S.t = 1.0;
        apply_step(S, U, dl, 1.0);
        if (absmax6(dl) <= tol_change) S.stop = RPE_STOP_STEP;
        else if (S.n_iter == max_iter) S.stop = RPE_STOP_MAX_ITER;
        return;
    }

    // ---- closure tail: clip_grad_norm_(y, 10)   (pose_head.py:76)
    double nrm = sqrt(dot6(g, g));
    double coef = 10.0 / (nrm + 1e-6);
    if (coef > 1.0) coef = 1.0;
    for (int i = 0; i < 6; ++i) g[i] *= coef;
    for (int i = 0; i < 6; ++i) S.g[i] = g[i];

    const int max_eval = max_iter * 5 / 4;
    if (S.evals == 0) {                       // initial evaluation of LBFGS.step
        S.evals = 1; S.loss = loss;
        if (absmax6(g) <= tol_grad) { S.stop = RPE_STOP_OPT_AT_START; return; }
        if (max_iter <= 0) { S.stop = RPE_STOP_MAX_ITER; return; }
    } else {                                  // evaluation that closes iteration S.n_iter
        S.evals += 1; S.loss = loss;
        bool opt = absmax6(g) <= tol_grad;
        double td[6];
        for (int i = 0; i < 6; ++i) td[i] = S.d[i] * S.t;
        if (S.evals >= max_eval) { S.stop = RPE_STOP_MAX_EVAL; return; }
        if (opt) { S.stop = RPE_STOP_OPT; return; }
        if (absmax6(td) <= tol_change) { S.stop = RPE_STOP_STEP; return; }
        if (fabs(loss - S.prev_loss) < tol_change) { S.stop = RPE_STOP_LOSS; return; }
    }
    // ---- next iteration: direction
    S.n_iter += 1;
    double d[6];
    if (S.n_iter == 1) {
        for (int i = 0; i < 6; ++i) d[i] = -g[i];
        S.H_diag = 1.0; S.num_old = 0;
    } else {
        double yk[6], sk[6];
        for (int i = 0; i < 6; ++i) { yk[i] = g[i] - S.prev_g[i]; sk[i] = S.d[i] * S.t; }
        double ys = dot6(yk, sk);
        if (ys > 1e-10) {
            bool shifted = false;
            if (S.num_old == hist) {                  // history full: drop the oldest pair
                for (int k = 1; k < hist; ++k) {
                    for (int i = 0; i < 6; ++i) { S.old_dirs[k - 1][i] = S.old_dirs[k][i]; S.old_stps[k - 1][i] = S.old_stps[k][i]; }
                    S.ro[k - 1] = S.ro[k];
                }
                S.num_old = hist - 1;
                shifted = true;                       // (every entry moved)
            }
            const int no = S.num_old;
            for (int i = 0; i < 6; ++i) { S.old_dirs[no][i] = yk[i]; S.old_stps[no][i] = sk[i]; }
            S.ro[no] = 1.0 / ys;
            S.num_old = no + 1;
            wb[0] = shifted ? 0 : no;
            wb[1] = no + 1;
            S.H_diag = ys / dot6(yk, yk);
        }
        const int nold = S.num_old;
        double qv[6];
        for (int i = 0; i < 6; ++i) qv[i] = -g[i];
        for (int k = nold - 1; k >= 0; --k) {
            const double a = dot6(S.old_stps[k], qv) * S.ro[k];
            h_al[k] = a;
            for (int i = 0; i < 6; ++i) qv[i] += S.old_dirs[k][i] * (-a);
        }
        for (int i = 0; i < 6; ++i) d[i] = qv[i] * S.H_diag;
        for (int k = 0; k < nold; ++k) {
            double be = dot6(S.old_dirs[k], d) * S.ro[k];
            for (int i = 0; i < 6; ++i) d[i] += S.old_stps[k][i] * (h_al[k] - be);
        }
    }
    for (int i = 0; i < 6; ++i) { S.prev_g[i] = g[i]; S.d[i] = d[i]; }
    S.prev_loss = loss;
    double tstep;
    if (S.n_iter == 1) {
        double l1 = 0.0;
        for (int i = 0; i < 6; ++i) l1 += fabs(g[i]);
        double inv = 1.0 / l1;
        tstep = (inv < 1.0 ? inv : 1.0) * lr;         // python min(1., x): x if x < 1. else 1.
    } else tstep = lr;
    S.t = tstep;
    double gtd = dot6(g, d);
    if (gtd > -tol_change) { S.stop = RPE_STOP_GTD; return; }
    apply_step(S, U, d, tstep);
    if (S.n_iter == max_iter) S.stop = RPE_STOP_MAX_ITER;
}

__global__ void k_pose_init(RowState* states, RowUniform* uni, int* tickets, const float* K, const float* lw, int n, int h, int w) {
    int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    tickets[(size_t)row * SYNC_STRIDE] = 0;
    ((long long*)(tickets + (size_t)row * SYNC_STRIDE + SYNC_EPOCH))[0] = 0;       // the two epoch words of the row's pose line
    ((long long*)(tickets + (size_t)row * SYNC_STRIDE + SYNC_EPOCH))[8] = 0;
    RowState& S = states[row];
    for (int i = 0; i < 6; ++i) { S.T[i] = 0.0; S.g[i] = 0.0; S.prev_g[i] = 0.0; S.d[i] = 0.0; }
    S.T[6] = 1.0;
    S.t = 0.0; S.loss = 0.0; S.prev_loss = 0.0; S.H_diag = 1.0;
    S.n_iter = 0; S.evals = 0; S.stop = 0; S.num_old = 0;
    write_rt(uni[row], S.T);
    write_consts(uni[row], K, lw, row, h, w);
}

// only for a solve without a single evaluation (Gauss-Newton with iters = 0): rows otherwise write their outputs in the tail that stops them
__global__ void k_pose_finalize(RowState* states, int n, double* T_out, float* vec7, float* log6, int32_t* info) {
    int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    RowState& S = states[row];
    if (S.stop == 0) S.stop = RPE_STOP_MAX_ITER;
    finalize_row(S, row, T_out, vec7, log6, info);
}

// Packs the reduced sums of one row into the rpe_pose_reduce output layout.
__global__ __launch_bounds__(UPD_THREADS) void k_pose_pack(const double* partials, int nblk, const float* lw, int h, int w, double* out) {
    const int row = blockIdx.x, lane = threadIdx.x;
    __shared__ double vals[NPART], red[UPD_THREADS / 32][NPART];
    sum_partials<false>(partials, row, nblk, lane, vals, red);
    if (lane >= 32) return;
    const double hwd = (double)h * (double)w;
    double* o = out + (size_t)row * 32;
    double l2 = vals[0] / hwd / hwd, l3 = vals[1] / hwd;
    if (lane == 0) o[0] = l2;
    else if (lane == 1) o[1] = l3;
    else if (lane == 2) o[2] = (double)lw[row * 2 + 1] * l2 + (double)lw[row * 2 + 0] * l3;
    else if (lane < 9) o[lane] = vals[lane - 1];
    else if (lane < 30) o[lane] = vals[lane - 1];
    else o[lane] = 0.0;
}

static bool pose_vec_ok(const PoseArgs& A) {
    bool vec = ((int64_t)A.h * A.w) % 4 == 0 && A.w % 4 == 0;
    const void* ptrs[] = {A.flow, A.pcl1, A.pcl2, A.w1, A.w2};
    for (const void* p : ptrs) vec = vec && ((uintptr_t)p % 16 == 0);
    return vec && ((uintptr_t)A.m1 % 4 == 0) && ((uintptr_t)A.m2 % 4 == 0);
}

template <bool TAIL>
static void launch_reduce(const PoseArgs& A, RowUniform* T, RowState* st, double* partials, int nblk, bool hess, hipStream_t s, const TailArgs& Z) {
    dim3 grid(nblk, A.n), block(RED_THREADS);
    if (hess) {
        if (pose_vec_ok(A)) hipLaunchKernelGGL((k_pose_reduce<true, 4, TAIL>), grid, block, 0, s, A, T, st, partials, Z);
        else hipLaunchKernelGGL((k_pose_reduce<true, 1, TAIL>), grid, block, 0, s, A, T, st, partials, Z);
    } else {
        if (pose_vec_ok(A)) hipLaunchKernelGGL((k_pose_reduce<false, 4, TAIL>), grid, block, 0, s, A, T, st, partials, Z);
        else hipLaunchKernelGGL((k_pose_reduce<false, 1, TAIL>), grid, block, 0, s, A, T, st, partials, Z);
    }
}

// Workgroups of the solve's kernel the current device holds at once (occupancy x compute units), 0 on any error: all evaluations go into
// one launch only when the whole grid fits (a workgroup waiting for its row's tail never yields its slot).
static int persistent_capacity(bool hess, bool vec) {
    const void* fn = hess ? (vec ? (const void*)k_pose_reduce<true, 4, true> : (const void*)k_pose_reduce<true, 1, true>)
                          : (vec ? (const void*)k_pose_reduce<false, 4, true> : (const void*)k_pose_reduce<false, 1, true>);
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    // a fact about (device, kernel): asked once (the occupancy query costs tens of microseconds of host time, a solve is enqueued in ~10);
    // a benign race between host threads writes the same value twice
    static int known[16][2][2];
    const bool cacheable = dev >= 0 && dev < 16;
    if (cacheable && known[dev][hess][vec] > 0) return known[dev][hess][vec];
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, RED_THREADS, 0) != hipSuccess) return 0;
    if (cacheable) known[dev][hess][vec] = cus * per_cu;
    return cus * per_cu;
}

static bool carve(void* ws, int n, int h, int w, RowState** st, RowUniform** uni, double** partials, int** tickets = nullptr) {
    if (!ws) return false;
    uintptr_t p = ((uintptr_t)ws + 255) / 256 * 256;
    *st = (RowState*)p;
    p += align_up(sizeof(RowState) * (size_t)n, 256);
    *uni = (RowUniform*)p;
    p += align_up(sizeof(RowUniform) * (size_t)n, 256);
    *partials = (double*)p;
    p += align_up(sizeof(double) * NPART * (size_t)pose_nblk(1, h, w) * n, 256);
    if (tickets) *tickets = (int*)p;
    return true;
}

extern "C" int rpe_pose_reduce(const float* flow, const float* pcl1, const float* pcl2, const float* w1, const float* w2,
                               const uint8_t* mask1, const uint8_t* mask2, const float* K, const float* loss_weight,
                               const double* T, int n, int h, int w, int need_hessian, double* out, void* workspace,
                               void* stream) {
    if (!flow || !pcl1 || !pcl2 || !w1 || !w2 || !mask1 || !mask2 || !K || !loss_weight || !T || !out || n <= 0 || h <= 0 || w <= 0)
        return RPE_E_BADARG;
    RowState* st; RowUniform* uni; double* partials;
    if (!carve(workspace, n, h, w, &st, &uni, &partials)) return RPE_E_BADARG;
    PoseArgs A{flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight, n, h, w};
    hipStream_t s = (hipStream_t)stream;
    int nblk = pose_nblk(n, h, w, need_hessian != 0);
    hipLaunchKernelGGL(k_pose_prep, dim3(ceil_div(n, 64)), dim3(64), 0, s, uni, T, K, loss_weight, n, h, w);
    launch_reduce<false>(A, uni, nullptr, partials, nblk, need_hessian != 0, s, TailArgs{});
    hipLaunchKernelGGL(k_pose_pack, dim3(n), dim3(UPD_THREADS), 0, s, (const double*)partials, nblk, loss_weight, h, w, out);
    return rpe_check_launch();
}

extern "C" int rpe_pose_solve_opts(const float* flow, const float* pcl1, const float* pcl2, const float* w1, const float* w2,
                                   const uint8_t* mask1, const uint8_t* mask2, const float* K, const float* loss_weight,
                                   int n, int h, int w, int mode, int iters, double tolerance_grad, double tolerance_change,
                                   int history_size, double* T_out, float* vec7, float* log6, int32_t* info,
                                   void* workspace, void* stream);

extern "C" int rpe_pose_solve_ex(const float* flow, const float* pcl1, const float* pcl2, const float* w1, const float* w2,
                                 const uint8_t* mask1, const uint8_t* mask2, const float* K, const float* loss_weight,
                                 int n, int h, int w, int mode, int iters, const rpe_solve_opts* opts, double* T_out, float* vec7, float* log6,
                                 int32_t* info, void* workspace, void* stream);

extern "C" int rpe_pose_solve(const float* flow, const float* pcl1, const float* pcl2, const float* w1, const float* w2,
                              const uint8_t* mask1, const uint8_t* mask2, const float* K, const float* loss_weight,
                              int n, int h, int w, int mode, int iters, double* T_out, float* vec7, float* log6,
                              int32_t* info, void* workspace, void* stream) {
    // torch.optim.LBFGS defaults, as DPoseSE3Head.solve constructs it (pose_head.py:70)
    return rpe_pose_solve_opts(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight, n, h, w, mode, iters, 1e-7, 1e-9, HIST,
                               T_out, vec7, log6, info, workspace, stream);
}

extern "C" int rpe_pose_solve_opts(const float* flow, const float* pcl1, const float* pcl2, const float* w1, const float* w2,
                                   const uint8_t* mask1, const uint8_t* mask2, const float* K, const float* loss_weight,
                                   int n, int h, int w, int mode, int iters, double tolerance_grad, double tolerance_change,
                                   int history_size, double* T_out, float* vec7, float* log6, int32_t* info,
                                   void* workspace, void* stream) {
    rpe_solve_opts o;
    o.struct_size = (int)sizeof(rpe_solve_opts); o.history_size = history_size; o.tolerance_grad = tolerance_grad; o.tolerance_change = tolerance_change;
    o.partition_rows = 0; o.reserved = 0;
    return rpe_pose_solve_ex(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight, n, h, w, mode, iters, &o, T_out, vec7, log6, info, workspace, stream);
}

extern "C" int rpe_pose_solve_ex(const float* flow, const float* pcl1, const float* pcl2, const float* w1, const float* w2,
                                 const uint8_t* mask1, const uint8_t* mask2, const float* K, const float* loss_weight,
                                 int n, int h, int w, int mode, int iters, const rpe_solve_opts* opts, double* T_out, float* vec7, float* log6,
                                 int32_t* info, void* workspace, void* stream) {
    // a caller built against a later header passes a LONGER struct: the fields this library knows are a prefix of it.  Shorter = an older
    // layout this library cannot complete (no field has been appended yet: the first one will read its default when struct_size stops before it)
    if (!opts || opts->struct_size < (int)sizeof(rpe_solve_opts)) return RPE_E_BADARG;
    const int history_size = opts->history_size;
    const double tolerance_grad = opts->tolerance_grad, tolerance_change = opts->tolerance_change;
    if (history_size < 1 || history_size > HIST || !(tolerance_grad >= 0.0) || !(tolerance_change >= 0.0) || opts->partition_rows < 0) return RPE_E_BADARG;
    const SolveOpts opt{tolerance_grad, tolerance_change, history_size};
    if (!flow || !pcl1 || !pcl2 || !w1 || !w2 || !mask1 || !mask2 || !K || !loss_weight || !T_out || n <= 0 || h <= 0 || w <= 0 || iters < 0)
        return RPE_E_BADARG;
    if (mode != RPE_SOLVER_LBFGS && mode != RPE_SOLVER_GN) return RPE_E_BADARG;
    RowState* st; RowUniform* uni; double* partials; int* tickets;
    if (!carve(workspace, n, h, w, &st, &uni, &partials, &tickets)) return RPE_E_BADARG;
    PoseArgs A{flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight, n, h, w};
    hipStream_t s = (hipStream_t)stream;
    // The reduction's block partition fixes the order of the f64 sums.  By default it follows the batch (one resident round of
    // workgroups chip-wide); with partition_rows = p it is the one a p-row batch would get, so with p = 1 a row's iterates are
    // bit-identical to solving it alone (rpe_pose_workspace_bytes covers every partition: n = 1 has the most blocks per row)
    int nblk = pose_nblk(opts->partition_rows > 0 && opts->partition_rows < n ? opts->partition_rows : n, h, w, mode == RPE_SOLVER_GN);
    hipLaunchKernelGGL(k_pose_init, dim3(ceil_div(n, 64)), dim3(64), 0, s, st, uni, tickets, K, loss_weight, n, h, w);
    // LBFGS with max_iter = N costs N evaluations (the last iteration moves without re-evaluating);
    // torch evaluates the closure once even for max_iter = 0.
    int evals = mode == RPE_SOLVER_LBFGS && iters == 0 ? 1 : iters;
    // every evaluation is ONE launch: the reduction, and in its tail (the row's last workgroup) the update, the stopping tests and -- for a
    // row that stops -- its outputs.  The last evaluation always stops a row (n_iter == max_iter at the latest).
    // ... and all evaluations are ONE launch when every workgroup of the grid is resident at once (the default partition is sized for
    // that: one round chip-wide): a row's workgroups then wait for their row's tail instead of for the next launch.  Otherwise the same
    // kernel is launched once per evaluation.
    const bool persist = evals > 1 && !(opts->reserved & RPE_SOLVE_LAUNCH_PER_EVALUATION) &&
                         (long long)nblk * n <= persistent_capacity(mode == RPE_SOLVER_GN, pose_vec_ok(A));
    const TailArgs Z{tickets, persist ? evals : 1, mode, iters, opt, T_out, vec7, log6, info};
    for (int it = 0; it < (persist ? 1 : evals); ++it) launch_reduce<true>(A, uni, st, partials, nblk, mode == RPE_SOLVER_GN, s, Z);
    if (evals == 0) hipLaunchKernelGGL(k_pose_finalize, dim3(ceil_div(n, 64)), dim3(64), 0, s, st, n, T_out, vec7, log6, info);
    return rpe_check_launch();
}
