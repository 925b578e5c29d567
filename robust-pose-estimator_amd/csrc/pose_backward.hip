// Backward of the declarative pose layer (implicit differentiation of the argmin) for gfx950.
//
// Replaces the reference's DeclarativeNodeLie.gradient / _get_objective_derivatives
// (core/optimization/declerative_node_lie.py:13-170,249-267), which builds fYY and fXY row by row with autograd
// (6 + 6 backward passes over the whole image per input), by their closed forms in two streaming passes.
//
// What the reference differentiates.  With double_backward=True the objective goes through Transform.apply
// (core/geometry/pinhole_transforms.py:58-76), whose backward is the explicit grad_T = grad_X [I | -[X]x] built from its
// SAVED OUTPUT X -- which autograd re-attaches to the graph, so the second differentiation also sees X inside the
// Jacobian (this is why the reference symmetrises fYY afterwards).  With l_p(X) the per-pixel loss of pose_head.py:12-58
// as a function of X = R p1 + t, J_p = [I | -[X_p]x] with columns J_i, g_p = dl_p/dX and M_p = d2l_p/dX2 (the EXACT 3x3
// Hessian: the reprojection term keeps the curvature of the pinhole projection times the residual, not only the
// Gauss-Newton part):
//     fY       = sum_p J_p^T g_p                                                 tangent gradient at the solution
//     fYY[i,j] = sum_p J_i . M_p J_j + [i >= 3] sum_p (J_j x g_p)_(i-3) ,        H = (fYY + fYY^T) / 2
//     fXY^T u for an input x:  sum over the pixel's entries of (d fY / d x)^T u
//         w1:   (J u) . g2_p / w1          w2:   (J u) . g3_p / w2          loss_weight: u . (fY of one term / its weight)
//         pcl2: -2 c3 w2 m3 (J u)          flow_c: -2 c2 w1 m2 (d pi_c/da) . K (J u)        pcl1: R^T (M_p (J u) + g_p x u_phi)
// The layer's gradient is then fXY^T u with u = -H^-1 v (v = dL/d log-pose), solved on the host side (6x6).
// Checked against gradients produced by the reference's own autograd code (tests/golden/backward_*.npz).
//
// Kernels (f64 arithmetic on the f32 inputs, like the forward):
//   k_bwd_moments : per block partial sums of the unit-loss-weight tangent gradients of both terms (6 + 6) and the 21
//                   upper-triangle entries of fYY; k_bwd_finish adds the partials in a fixed order (no atomics)
//   k_bwd_grads   : one thread per pixel writes the five per-pixel input gradients (any of them optional)
#include "rpe_common.h"

#define BW_THREADS 256
#define BW_NACC 42            // g2u[6], g3u[6], sum J^T M J [21], S = sum X g^T [9]
#define BW_NPART 48

struct BwArgs {
    const float* flow; const float* pcl1; const float* pcl2; const float* w1; const float* w2;
    const uint8_t* m1; const uint8_t* m2; const float* K; const float* lw; const double* T;
    int n, h, w;
};

struct BwRow { double R[9], t[3], K[9], c2u, c3u, lw0, lw1; };

__device__ __forceinline__ void load_row(const BwArgs& A, int row, BwRow& U) {
    const double* Tp = A.T + (size_t)row * 7;
    const double qx = Tp[3], qy = Tp[4], qz = Tp[5], qw = Tp[6];
    U.R[0] = 1.0 - 2.0 * (qy * qy + qz * qz); U.R[1] = 2.0 * (qx * qy - qz * qw); U.R[2] = 2.0 * (qx * qz + qy * qw);
    U.R[3] = 2.0 * (qx * qy + qz * qw); U.R[4] = 1.0 - 2.0 * (qx * qx + qz * qz); U.R[5] = 2.0 * (qy * qz - qx * qw);
    U.R[6] = 2.0 * (qx * qz - qy * qw); U.R[7] = 2.0 * (qy * qz + qx * qw); U.R[8] = 1.0 - 2.0 * (qx * qx + qy * qy);
    U.t[0] = Tp[0]; U.t[1] = Tp[1]; U.t[2] = Tp[2];
    for (int i = 0; i < 9; ++i) U.K[i] = (double)A.K[(size_t)row * 9 + i];
    const double hwd = (double)A.h * (double)A.w;
    U.c2u = 1.0 / hwd / hwd;            // mean, then / (h*w)   (pose_head.py:29)
    U.c3u = 1.0 / hwd;                  // mean                 (pose_head.py:51)
    U.lw0 = (double)A.lw[row * 2 + 0]; U.lw1 = (double)A.lw[row * 2 + 1];
}

// Everything both kernels need at one pixel.
struct PixelTerms {
    double X[3];
    double g2[3], g3[3];       // dl/dX of the two terms with unit loss weight and unit per-pixel weight (gates applied)
    double M[6];               // d2l/dX2 (xx, xy, xz, yy, yz, zz) with all weights applied
    double dpi[2][3];          // d pi_c / d a  (a = K X)
    double s2;                 // 2 * c2u * gate2  (factor of the reprojection term without lw1, w1)
    double s3;                 // 2 * c3u * gate3
};

__device__ __forceinline__ void pixel_terms(const BwRow& U, double px, double py, double fl_x, double fl_y, const double p[3], const double q[3],
                                            double w1, double w2, bool m1, bool m2, double Wd, double Hd, PixelTerms& P) {
    const double* R = U.R; const double* K = U.K;
    const double X = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + U.t[0];
    const double Y = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + U.t[1];
    const double Z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + U.t[2];
    P.X[0] = X; P.X[1] = Y; P.X[2] = Z;
    const double a0 = K[0] * X + K[1] * Y + K[2] * Z, a1 = K[3] * X + K[4] * Y + K[5] * Z, a2 = K[6] * X + K[7] * Y + K[8] * Z;
    const double d = a2 < 1e-12 ? 1e-12 : a2;                  // clamp(z, 1e-12)        (pinhole_transforms.py:96)
    const double s = a2 >= 1e-12 ? 1.0 : 0.0;                  // its derivative
    const double u = a0 / d, v = a1 / d;
    const double fx = px + fl_x, fy = py + fl_y;
    const double r0 = fx - u, r1 = fy - v;
    const double res = (r0 * r0 + r1 * r1) * w1;
    const bool inimg = (fx > 0.0) && (fy > 0.0) && (fx < Wd) && (fy < Hd);
    const bool bad = isinf(res) || isnan(res) || !inimg || !m1; // pose_head.py:24-28: zeroed by assignment -> no gradient at all
    P.s2 = bad ? 0.0 : 2.0 * U.c2u;
    P.s3 = (m1 && m2) ? 2.0 * U.c3u : 0.0;
    const double id = 1.0 / d, id2 = id * id;
    P.dpi[0][0] = id;  P.dpi[0][1] = 0.0; P.dpi[0][2] = -s * a0 * id2;
    P.dpi[1][0] = 0.0; P.dpi[1][1] = id;  P.dpi[1][2] = -s * a1 * id2;
    // dl2/da = -s2 (r0 dpi0 + r1 dpi1)   [unit w1, lw1]
    double ga[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) ga[i] = bad ? 0.0 : -P.s2 * (r0 * P.dpi[0][i] + r1 * P.dpi[1][i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) P.g2[i] = K[i] * ga[0] + K[3 + i] * ga[1] + K[6 + i] * ga[2];     // K^T ga
    const double e0 = X - q[0], e1 = Y - q[1], e2 = Z - q[2];
    P.g3[0] = P.s3 * e0; P.g3[1] = P.s3 * e1; P.g3[2] = P.s3 * e2;
    // d2l2/da2 = s2 [dpi0 dpi0^T + dpi1 dpi1^T - r0 d2pi0 - r1 d2pi1];   d2pi_c: (c,2) = (2,c) = -s/d^2, (2,2) = 2 s a_c / d^3
    double Ha[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Ha[i][j] = P.dpi[0][i] * P.dpi[0][j] + P.dpi[1][i] * P.dpi[1][j];
    const double cr = s * id2;
    Ha[0][2] += r0 * cr; Ha[2][0] += r0 * cr;
    Ha[1][2] += r1 * cr; Ha[2][1] += r1 * cr;
    Ha[2][2] -= 2.0 * s * (r0 * a0 + r1 * a1) * id2 * id;
    const double k2 = bad ? 0.0 : P.s2 * w1 * U.lw1;
    // M = K^T Ha K * k2 + s3 w2 lw0 I
    double HK[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) HK[i][j] = Ha[i][0] * K[j] + Ha[i][1] * K[3 + j] + Ha[i][2] * K[6 + j];
    double Mx[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Mx[i][j] = bad ? 0.0 : k2 * (K[i] * HK[0][j] + K[3 + i] * HK[1][j] + K[6 + i] * HK[2][j]);
    const double k3 = P.s3 * w2 * U.lw0;
    P.M[0] = Mx[0][0] + k3; P.M[1] = Mx[0][1]; P.M[2] = Mx[0][2]; P.M[3] = Mx[1][1] + k3; P.M[4] = Mx[1][2]; P.M[5] = Mx[2][2] + k3;
}

__device__ __forceinline__ int tri6(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

// J^T v for J = [I | -[X]x]: (v, X x v)
__device__ __forceinline__ void jt_apply(const double X[3], const double v[3], double out[6]) {
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
    out[3] = X[1] * v[2] - X[2] * v[1];
    out[4] = X[2] * v[0] - X[0] * v[2];
    out[5] = X[0] * v[1] - X[1] * v[0];
}

__device__ __forceinline__ void load_pixel(const BwArgs& A, int row, int64_t hw, int64_t p, double& fl_x, double& fl_y, double P1[3], double P2[3],
                                           double& w1, double& w2, bool& m1, bool& m2) {
    const float* fl = A.flow + (size_t)row * 2 * hw;
    fl_x = (double)fl[p]; fl_y = (double)fl[hw + p];
    const float* a = A.pcl1 + (size_t)row * 3 * hw; const float* b = A.pcl2 + (size_t)row * 3 * hw;
    P1[0] = (double)a[p]; P1[1] = (double)a[hw + p]; P1[2] = (double)a[2 * hw + p];
    P2[0] = (double)b[p]; P2[1] = (double)b[hw + p]; P2[2] = (double)b[2 * hw + p];
    w1 = (double)A.w1[(size_t)row * hw + p]; w2 = (double)A.w2[(size_t)row * hw + p];
    m1 = A.m1[(size_t)row * hw + p] != 0; m2 = A.m2[(size_t)row * hw + p] != 0;
}

__global__ __launch_bounds__(BW_THREADS) void k_bwd_moments(BwArgs A, double* __restrict__ partials) {
    const int row = blockIdx.y, nblk = gridDim.x;
    const int64_t hw = (int64_t)A.h * A.w;
    BwRow U;
    load_row(A, row, U);
    double acc[BW_NACC];
#pragma unroll
    for (int i = 0; i < BW_NACC; ++i) acc[i] = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * BW_THREADS + threadIdx.x; p < hw; p += (int64_t)nblk * BW_THREADS) {
        double fx, fy, P1[3], P2[3], w1, w2; bool m1, m2;
        load_pixel(A, row, hw, p, fx, fy, P1, P2, w1, w2, m1, m2);
        const int y = (int)(p / A.w), x = (int)(p - (int64_t)y * A.w);
        PixelTerms T;
        pixel_terms(U, (double)x + 0.5, (double)y + 0.5, fx, fy, P1, P2, w1, w2, m1, m2, (double)A.w, (double)A.h, T);
        double g2w[3] = {T.g2[0] * w1, T.g2[1] * w1, T.g2[2] * w1}, g3w[3] = {T.g3[0] * w2, T.g3[1] * w2, T.g3[2] * w2};
        double t6[6];
        jt_apply(T.X, g2w, t6);
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i] += t6[i];
        jt_apply(T.X, g3w, t6);
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[6 + i] += t6[i];
        // J^T M J: columns of J are e_0..e_2 and e_k x X
        const double X = T.X[0], Y = T.X[1], Z = T.X[2];
        const double J[6][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, -Z, Y}, {Z, 0, -X}, {-Y, X, 0}};
        double MJ[6][3];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            MJ[j][0] = T.M[0] * J[j][0] + T.M[1] * J[j][1] + T.M[2] * J[j][2];
            MJ[j][1] = T.M[1] * J[j][0] + T.M[3] * J[j][1] + T.M[4] * J[j][2];
            MJ[j][2] = T.M[2] * J[j][0] + T.M[4] * J[j][1] + T.M[5] * J[j][2];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) acc[12 + tri6(i, j)] += J[i][0] * MJ[j][0] + J[i][1] * MJ[j][1] + J[i][2] * MJ[j][2];
        // S = sum X g^T with the full per-pixel gradient g: the part of fYY that comes from differentiating J itself
        const double g[3] = {U.lw1 * g2w[0] + U.lw0 * g3w[0], U.lw1 * g2w[1] + U.lw0 * g3w[1], U.lw1 * g2w[2] + U.lw0 * g3w[2]};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[33 + 3 * a + b] += T.X[a] * g[b];
    }
    __shared__ double red[BW_THREADS / RPE_WAVE][BW_NPART];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < BW_NACC; ++i) {
        const double sacc = wave_sum(acc[i]);
        if (lane == 0) red[wv][i] = sacc;
    }
    __syncthreads();
    if (threadIdx.x < BW_NPART) {
        double sacc = 0.0;
        if (threadIdx.x < BW_NACC) sacc = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        partials[((size_t)row * nblk + blockIdx.x) * BW_NPART + threadIdx.x] = sacc;
    }
}

// out (n, 48): [0,6) g2u, [6,12) g3u (tangent gradients of the two terms with unit loss weight), [12,48) H row-major 6x6 =
// (fYY + fYY^T)/2.  The J-derivative part of fYY, E[3+k][j] = sum_p (J_j x g_p)_k, needs only G = sum g (= fY_tau) and S:
//   E[3+k][j<3] = (e_j x G)_k ,   E[3+a][3+b] = S[a][b] - delta_ab tr S
__global__ __launch_bounds__(64) void k_bwd_finish(const double* __restrict__ partials, int nblk, const float* __restrict__ lw,
                                                   double* __restrict__ out) {
    const int row = blockIdx.x, lane = threadIdx.x;
    __shared__ double vals[BW_NPART];
    if (lane < BW_NPART) {
        double sacc = 0.0;
        for (int b = 0; b < nblk; ++b) sacc += partials[((size_t)row * nblk + b) * BW_NPART + lane];
        vals[lane] = sacc;
    }
    __syncthreads();
    double* o = out + (size_t)row * 48;
    if (lane < 12) o[lane] = vals[lane];
    if (lane < 36) {
        const int i = lane / 6, j = lane % 6;
        double hij = vals[12 + (i <= j ? tri6(i, j) : tri6(j, i))];
        const double lw0 = (double)lw[row * 2 + 0], lw1 = (double)lw[row * 2 + 1];
        const double G[3] = {lw1 * vals[0] + lw0 * vals[6], lw1 * vals[1] + lw0 * vals[7], lw1 * vals[2] + lw0 * vals[8]};
        const double* S = vals + 33;
        auto E = [&](int r, int c) -> double {                  // the non-symmetric extra term of fYY
            if (r < 3) return 0.0;
            const int k = r - 3;
            if (c < 3) {                                         // (e_c x G)_k
                const int k1 = (c + 1) % 3, k2 = (c + 2) % 3;    // e_c x G = G[k2] e_k1 - G[k1] e_k2  (cyclic)
                return k == k1 ? -G[k2] : (k == k2 ? G[k1] : 0.0);
            }
            const int b = c - 3;
            return S[3 * k + b] - (k == b ? S[0] + S[4] + S[8] : 0.0);
        };
        hij += 0.5 * (E(i, j) + E(j, i));
        o[12 + lane] = hij;
    }
}

struct BwOut { float* gflow; float* gp1; float* gp2; float* gw1; float* gw2; };

__device__ __forceinline__ float nan0(double v) { return isnan(v) ? 0.0f : (float)v; }     // gradient[isnan] = 0 (declerative_node_lie.py:76)

__global__ __launch_bounds__(BW_THREADS) void k_bwd_grads(BwArgs A, const double* __restrict__ uvec, BwOut O) {
    const int row = blockIdx.y;
    const int64_t hw = (int64_t)A.h * A.w;
    const int64_t p = (int64_t)blockIdx.x * BW_THREADS + threadIdx.x;
    if (p >= hw) return;
    BwRow U;
    load_row(A, row, U);
    double u[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) u[i] = uvec[(size_t)row * 6 + i];
    double fx, fy, P1[3], P2[3], w1, w2; bool m1, m2;
    load_pixel(A, row, hw, p, fx, fy, P1, P2, w1, w2, m1, m2);
    const int y = (int)(p / A.w), x = (int)(p - (int64_t)y * A.w);
    PixelTerms T;
    pixel_terms(U, (double)x + 0.5, (double)y + 0.5, fx, fy, P1, P2, w1, w2, m1, m2, (double)A.w, (double)A.h, T);
    // J u = u_tau + u_phi x X
    const double X = T.X[0], Y = T.X[1], Z = T.X[2];
    const double Ju[3] = {u[0] + u[4] * Z - u[5] * Y, u[1] + u[5] * X - u[3] * Z, u[2] + u[3] * Y - u[4] * X};
    if (O.gw1) O.gw1[(size_t)row * hw + p] = nan0(U.lw1 * (Ju[0] * T.g2[0] + Ju[1] * T.g2[1] + Ju[2] * T.g2[2]));
    if (O.gw2) O.gw2[(size_t)row * hw + p] = nan0(U.lw0 * (Ju[0] * T.g3[0] + Ju[1] * T.g3[1] + Ju[2] * T.g3[2]));
    if (O.gp2) {
        const double c = -T.s3 * w2 * U.lw0;
        float* g = O.gp2 + (size_t)row * 3 * hw;
        g[p] = nan0(c * Ju[0]); g[hw + p] = nan0(c * Ju[1]); g[2 * hw + p] = nan0(c * Ju[2]);
    }
    if (O.gflow) {
        const double* K = U.K;
        const double KJ[3] = {K[0] * Ju[0] + K[1] * Ju[1] + K[2] * Ju[2], K[3] * Ju[0] + K[4] * Ju[1] + K[5] * Ju[2],
                              K[6] * Ju[0] + K[7] * Ju[1] + K[8] * Ju[2]};
        const double c = -T.s2 * w1 * U.lw1;
        float* g = O.gflow + (size_t)row * 2 * hw;
        g[p] = nan0(c * (T.dpi[0][0] * KJ[0] + T.dpi[0][1] * KJ[1] + T.dpi[0][2] * KJ[2]));
        g[hw + p] = nan0(c * (T.dpi[1][0] * KJ[0] + T.dpi[1][1] * KJ[1] + T.dpi[1][2] * KJ[2]));
    }
    if (O.gp1) {
        // full per-pixel gradient g and the J-derivative term g x u_phi
        const double gt[3] = {U.lw1 * w1 * T.g2[0] + U.lw0 * w2 * T.g3[0], U.lw1 * w1 * T.g2[1] + U.lw0 * w2 * T.g3[1],
                              U.lw1 * w1 * T.g2[2] + U.lw0 * w2 * T.g3[2]};
        const double MJ[3] = {T.M[0] * Ju[0] + T.M[1] * Ju[1] + T.M[2] * Ju[2] + (gt[1] * u[5] - gt[2] * u[4]),
                              T.M[1] * Ju[0] + T.M[3] * Ju[1] + T.M[4] * Ju[2] + (gt[2] * u[3] - gt[0] * u[5]),
                              T.M[2] * Ju[0] + T.M[4] * Ju[1] + T.M[5] * Ju[2] + (gt[0] * u[4] - gt[1] * u[3])};
        const double* R = U.R;
        float* g = O.gp1 + (size_t)row * 3 * hw;
        g[p] = nan0(R[0] * MJ[0] + R[3] * MJ[1] + R[6] * MJ[2]);
        g[hw + p] = nan0(R[1] * MJ[0] + R[4] * MJ[1] + R[7] * MJ[2]);
        g[2 * hw + p] = nan0(R[2] * MJ[0] + R[5] * MJ[1] + R[8] * MJ[2]);
    }
}

static int bw_nblk(int n, int h, int w) {
    int64_t hw = (int64_t)h * w;
    int64_t per = (hw + BW_THREADS * 4 - 1) / (BW_THREADS * 4);
    int64_t want = (1024 + n - 1) / n;
    int64_t nb = per < want ? per : want;
    return (int)(nb < 1 ? 1 : nb);
}

extern "C" size_t rpe_pose_backward_workspace_bytes(int n, int h, int w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    return sizeof(double) * BW_NPART * (size_t)bw_nblk(n, h, w) * n + 256;
}

extern "C" int rpe_pose_backward_moments(const float* flow, const float* pcl1, const float* pcl2, const float* w1, const float* w2,
                                         const uint8_t* mask1, const uint8_t* mask2, const float* K, const float* loss_weight,
                                         const double* T, int n, int h, int w, double* out, void* workspace, void* stream) {
    if (!flow || !pcl1 || !pcl2 || !w1 || !w2 || !mask1 || !mask2 || !K || !loss_weight || !T || !out || !workspace || n <= 0 || h <= 0 || w <= 0)
        return RPE_E_BADARG;
    BwArgs A{flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight, T, n, h, w};
    double* partials = (double*)(((uintptr_t)workspace + 255) / 256 * 256);
    const int nblk = bw_nblk(n, h, w);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bwd_moments, dim3(nblk, n), dim3(BW_THREADS), 0, s, A, partials);
    hipLaunchKernelGGL(k_bwd_finish, dim3(n), dim3(64), 0, s, (const double*)partials, nblk, loss_weight, out);
    return rpe_check_launch();
}

extern "C" int rpe_pose_backward_grads(const float* flow, const float* pcl1, const float* pcl2, const float* w1, const float* w2,
                                       const uint8_t* mask1, const uint8_t* mask2, const float* K, const float* loss_weight,
                                       const double* T, const double* u, int n, int h, int w, float* grad_flow, float* grad_pcl1,
                                       float* grad_pcl2, float* grad_w1, float* grad_w2, void* stream) {
    if (!flow || !pcl1 || !pcl2 || !w1 || !w2 || !mask1 || !mask2 || !K || !loss_weight || !T || !u || n <= 0 || h <= 0 || w <= 0)
        return RPE_E_BADARG;
    BwArgs A{flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight, T, n, h, w};
    BwOut O{grad_flow, grad_pcl1, grad_pcl2, grad_w1, grad_w2};
    const int64_t hw = (int64_t)h * w;
    hipLaunchKernelGGL(k_bwd_grads, dim3(ceil_div(hw, BW_THREADS), n), dim3(BW_THREADS), 0, (hipStream_t)stream, A, u, O);
    return rpe_check_launch();
}
