// Stand-alone consumer of the C ABI (no Python, no torch): build with
//   hipcc --offload-arch=gfx950 -I include tests/abi/abi_smoke.cpp -L robust-pose-estimator_amd -lrpe_hip -o abi_smoke
// Runs rpe_se3_exp/log round trips and a tiny pose solve with a known answer, prints "ABI_SMOKE_OK".
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rpe.h"

#define CK(x) do { if ((x) != hipSuccess) { std::printf("hip error line %d\n", __LINE__); return 2; } } while (0)

int main() {
    std::printf("%s\n", rpe_version());
    // --- SE(3): log(exp(xi)) == xi
    const int n = 64;
    std::vector<double> xi(n * 6), back(n * 6);
    for (int i = 0; i < n * 6; ++i) xi[i] = 0.3 * std::sin(0.37 * i + 1.0);
    double *d_xi, *d_T, *d_back;
    CK(hipMalloc(&d_xi, n * 6 * 8)); CK(hipMalloc(&d_T, n * 7 * 8)); CK(hipMalloc(&d_back, n * 6 * 8));
    CK(hipMemcpy(d_xi, xi.data(), n * 6 * 8, hipMemcpyHostToDevice));
    if (rpe_se3_exp(d_xi, d_T, n, RPE_F64, nullptr) != RPE_OK || rpe_se3_log(d_T, d_back, n, RPE_F64, nullptr) != RPE_OK) return 3;
    CK(hipMemcpy(back.data(), d_back, n * 6 * 8, hipMemcpyDeviceToHost));
    double err = 0;
    for (int i = 0; i < n * 6; ++i) err = std::fmax(err, std::fabs(back[i] - xi[i]));
    std::printf("se3 round trip max err %.3e\n", err);
    if (!(err < 1e-12)) return 4;
    // --- pose solve: a fronto-parallel plane translated by (0.01, -0.02, 0.03); flow and target cloud are consistent,
    //     so the minimiser is that translation
    const int h = 32, w = 48, hw = h * w;
    const float fx = 50.f, cx = 24.f, cy = 16.f, tx = 0.01f, ty = -0.02f, tz = 0.03f;
    std::vector<float> flow(2 * hw), p1(3 * hw), p2(3 * hw), ones(hw, 1.0f), K = {fx, 0, cx, 0, fx, cy, 0, 0, 1}, lw = {1, 1};
    std::vector<unsigned char> m(hw, 1);
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        int p = y * w + x;
        float Z = 1.0f + 0.2f * std::sin(0.3f * x) * std::cos(0.2f * y);
        float X = (x + 0.5f - cx) / fx * Z, Y = (y + 0.5f - cy) / fx * Z;
        p1[p] = X; p1[hw + p] = Y; p1[2 * hw + p] = Z;
        float X2 = X + tx, Y2 = Y + ty, Z2 = Z + tz;
        p2[p] = X2; p2[hw + p] = Y2; p2[2 * hw + p] = Z2;
        flow[p] = fx * X2 / Z2 + cx - (x + 0.5f); flow[hw + p] = fx * Y2 / Z2 + cy - (y + 0.5f);
    }
    float *d_flow, *d_p1, *d_p2, *d_w, *d_K, *d_lw, *d_v7, *d_l6; unsigned char* d_m; double* d_Tout; int* d_info; void* d_ws;
    size_t wsb = rpe_pose_workspace_bytes(1, h, w);
    CK(hipMalloc(&d_flow, 2 * hw * 4)); CK(hipMalloc(&d_p1, 3 * hw * 4)); CK(hipMalloc(&d_p2, 3 * hw * 4)); CK(hipMalloc(&d_w, hw * 4));
    CK(hipMalloc(&d_K, 36)); CK(hipMalloc(&d_lw, 8)); CK(hipMalloc(&d_m, hw)); CK(hipMalloc(&d_Tout, 56)); CK(hipMalloc(&d_v7, 28));
    CK(hipMalloc(&d_l6, 24)); CK(hipMalloc(&d_info, 16)); CK(hipMalloc(&d_ws, wsb));
    CK(hipMemcpy(d_flow, flow.data(), 2 * hw * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_p1, p1.data(), 3 * hw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_p2, p2.data(), 3 * hw * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, ones.data(), hw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_K, K.data(), 36, hipMemcpyHostToDevice)); CK(hipMemcpy(d_lw, lw.data(), 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_m, m.data(), hw, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    int rc = rpe_pose_solve(d_flow, d_p1, d_p2, d_w, d_w, d_m, d_m, d_K, d_lw, 1, h, w, RPE_SOLVER_GN, 10, d_Tout, d_v7, d_l6, d_info, d_ws, st);
    if (rc != RPE_OK) { std::printf("rpe_pose_solve rc %d\n", rc); return 5; }
    CK(hipStreamSynchronize(st));
    double T[7]; int info[4];
    CK(hipMemcpy(T, d_Tout, 56, hipMemcpyDeviceToHost)); CK(hipMemcpy(info, d_info, 16, hipMemcpyDeviceToHost));
    std::printf("pose %.6f %.6f %.6f | q %.2e %.2e %.2e %.6f | iters %d stop %d\n", T[0], T[1], T[2], T[3], T[4], T[5], T[6], info[0], info[2]);
    if (std::fabs(T[0] - tx) > 1e-5 || std::fabs(T[1] - ty) > 1e-5 || std::fabs(T[2] - tz) > 1e-5 || std::fabs(T[6] - 1.0) > 1e-8) return 6;
    // --- bad arguments are rejected with a status, not a crash
    if (rpe_pose_solve(nullptr, d_p1, d_p2, d_w, d_w, d_m, d_m, d_K, d_lw, 1, h, w, 0, 8, d_Tout, nullptr, nullptr, nullptr, d_ws, st) != RPE_E_BADARG) return 7;
    if (rpe_corr_lookup(d_ws, d_flow, 1, 8, 8, 4, 3, d_p1, st) != RPE_E_BADARG) return 8;       // radius 3 unsupported
    std::printf("ABI_SMOKE_OK\n");
    return 0;
}
