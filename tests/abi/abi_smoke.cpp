// Stand-alone consumer of the C ABI (no Python, no torch): build with
//   hipcc --offload-arch=gfx950 -I include tests/abi/abi_smoke.cpp -L robust-pose-estimator_amd -lrpe_hip -o abi_smoke
// Runs rpe_se3_exp/log round trips, a tiny pose solve with a known answer, a correlation pyramid build + lookup, one
// fused convolution checked against host loops and a two-stream prepared launch list (rpe_run_ops), prints "ABI_SMOKE_OK".
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
    // --- correlation pyramid: build + lookup at integer coordinates.  Level-0 window tap (i, j) of query q is then
    //     <fmap1[:, q], fmap2[:, q + (i-4, j-4)]> / sqrt(C) (zero outside the map); channel index = i*9 + j with i the x offset
    {
        const int b = 1, C = 32, h8 = 16, w8 = 24, nq = h8 * w8, L = 4;
        std::vector<float> f1((size_t)C * nq), f2((size_t)C * nq), co(2 * nq), out((size_t)L * 81 * nq);
        for (size_t i = 0; i < f1.size(); ++i) { f1[i] = std::sin(0.013f * i + 0.5f); f2[i] = std::cos(0.007f * i); }
        for (int q = 0; q < nq; ++q) { co[q] = (float)(q % w8); co[nq + q] = (float)(q / w8); }
        float *d_f1, *d_f2, *d_co, *d_out; void* d_pyr;
        const size_t pb = rpe_corr_pyramid_bytes(b, h8, w8, L);
        if (pb == 0) return 9;
        CK(hipMalloc(&d_f1, f1.size() * 4)); CK(hipMalloc(&d_f2, f2.size() * 4)); CK(hipMalloc(&d_co, co.size() * 4));
        CK(hipMalloc(&d_out, out.size() * 4)); CK(hipMalloc(&d_pyr, pb));
        CK(hipMemcpy(d_f1, f1.data(), f1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_f2, f2.data(), f2.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_co, co.data(), co.size() * 4, hipMemcpyHostToDevice));
        if (rpe_corr_build(d_f1, d_f2, b, C, h8, w8, L, d_pyr, st) != RPE_OK) return 10;
        if (rpe_corr_lookup(d_pyr, d_co, b, h8, w8, L, 4, d_out, st) != RPE_OK) return 11;
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
        double cerr = 0;
        for (int q = 0; q < nq; q += 7) for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) {
            const int x = q % w8 + i - 4, y = q / w8 + j - 4;
            double ref = 0;
            if (x >= 0 && x < w8 && y >= 0 && y < h8) { for (int c = 0; c < C; ++c) ref += (double)f1[(size_t)c * nq + q] * f2[(size_t)c * nq + y * w8 + x]; ref /= std::sqrt((double)C); }
            cerr = std::fmax(cerr, std::fabs(out[(size_t)(i * 9 + j) * nq + q] - ref));
        }
        std::printf("corr build+lookup max err %.3e\n", cerr);
        if (!(cerr < 2e-5)) return 12;
    }
    // --- one fused convolution: 3x3, 16 -> 16 channels, bias + ReLU, against host loops
    {
        const int cin = 16, cout = 16, hh = 8, ww = 12, hw2 = hh * ww;
        std::vector<float> x((size_t)cin * hw2), wt((size_t)cout * cin * 9), bias(cout), y((size_t)cout * hw2);
        for (size_t i = 0; i < x.size(); ++i) x[i] = std::sin(0.11f * i);
        for (size_t i = 0; i < wt.size(); ++i) wt[i] = 0.1f * std::cos(0.23f * i + 0.1f);
        for (int c = 0; c < cout; ++c) bias[c] = 0.05f * c - 0.3f;
        float *d_x, *d_wt, *d_b, *d_y, *d_pk;
        const size_t pf = rpe_conv_packed_floats(cout, cin, 3, 3);
        CK(hipMalloc(&d_x, x.size() * 4)); CK(hipMalloc(&d_wt, wt.size() * 4)); CK(hipMalloc(&d_b, cout * 4)); CK(hipMalloc(&d_y, y.size() * 4));
        CK(hipMalloc(&d_pk, pf * 4));
        CK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_wt, wt.data(), wt.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_b, bias.data(), cout * 4, hipMemcpyHostToDevice));
        if (rpe_conv_pack(d_wt, cout, cin, 3, 3, d_pk, st) != RPE_OK) return 13;
        rpe_conv_desc d = {};
        d.x = d_x; d.x_batch_stride = (long long)cin * hw2; d.packed = d_pk; d.bias = d_b; d.out = d_y; d.out_batch_stride = (long long)cout * hw2;
        d.b = 1; d.cin = cin; d.cout = cout; d.h = hh; d.w = ww; d.kh = 3; d.kw = 3; d.mode = RPE_CONV_RELU; d.stride = 1;
        if (rpe_conv_fused(&d, st) != RPE_OK) return 14;
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(y.data(), d_y, y.size() * 4, hipMemcpyDeviceToHost));
        double verr = 0;
        for (int co2 = 0; co2 < cout; ++co2) for (int yy = 0; yy < hh; ++yy) for (int xx = 0; xx < ww; ++xx) {
            double a = bias[co2];
            for (int ci = 0; ci < cin; ++ci) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
                const int y2 = yy + dy, x2 = xx + dx;
                if (y2 < 0 || y2 >= hh || x2 < 0 || x2 >= ww) continue;
                a += (double)wt[((size_t)(co2 * cin + ci) * 3 + dy + 1) * 3 + dx + 1] * x[(size_t)ci * hw2 + y2 * ww + x2];
            }
            verr = std::fmax(verr, std::fabs(y[(size_t)co2 * hw2 + yy * ww + xx] - (a > 0 ? a : 0)));
        }
        std::printf("conv_fused max err %.3e\n", verr);
        if (!(verr < 1e-5)) return 15;
    }
    // --- a prepared launch list (rpe_run_ops): the same convolution twice into two buffers on two streams, forked and joined by an event
    //     pair, then a plane copy on the first stream that reads the second stream's result; a failing op is reported by index
    {
        const int cin = 16, cout = 16, hh = 8, ww = 12, hw2 = hh * ww;
        std::vector<float> x((size_t)cin * hw2), wt((size_t)cout * cin * 9), ya((size_t)cout * hw2), yc((size_t)cout * hw2);
        for (size_t i = 0; i < x.size(); ++i) x[i] = std::cos(0.07f * i);
        for (size_t i = 0; i < wt.size(); ++i) wt[i] = 0.05f * std::sin(0.31f * i);
        float *d_x, *d_wt, *d_pk, *d_ya, *d_yb, *d_yc;
        const size_t pf = rpe_conv_packed_floats(cout, cin, 3, 3);
        CK(hipMalloc(&d_x, x.size() * 4)); CK(hipMalloc(&d_wt, wt.size() * 4)); CK(hipMalloc(&d_pk, pf * 4));
        CK(hipMalloc(&d_ya, ya.size() * 4)); CK(hipMalloc(&d_yb, ya.size() * 4)); CK(hipMalloc(&d_yc, ya.size() * 4));
        CK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_wt, wt.data(), wt.size() * 4, hipMemcpyHostToDevice));
        if (rpe_conv_pack(d_wt, cout, cin, 3, 3, d_pk, st) != RPE_OK) return 16;
        rpe_conv_desc da = {};
        da.x = d_x; da.x_batch_stride = (long long)cin * hw2; da.packed = d_pk; da.out = d_ya; da.out_batch_stride = (long long)cout * hw2;
        da.b = 1; da.cin = cin; da.cout = cout; da.h = hh; da.w = ww; da.kh = 3; da.kw = 3; da.mode = RPE_CONV_LINEAR; da.stride = 1;
        rpe_conv_desc db = da; db.out = d_yb;
        rpe_copy_planes_args cp = {d_yb, (long long)cout * hw2, d_yc, (long long)cout * hw2, 1, cout, hw2};
        hipStream_t st2; CK(hipStreamCreate(&st2));
        hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
        void* cells[3] = {fork, join, nullptr};                           // (a NULL cell: the op is skipped)
        const rpe_op list[] = {{RPE_OP_EVENT_RECORD, 0, &cells[0]}, {RPE_OP_STREAM_WAIT, 1, &cells[0]}, {RPE_OP_CONV_FUSED, 1, &db},
                               {RPE_OP_EVENT_RECORD, 1, &cells[1]}, {RPE_OP_CONV_FUSED, 0, &da}, {RPE_OP_EVENT_RECORD, 0, &cells[2]},
                               {RPE_OP_STREAM_WAIT, 0, &cells[1]}, {RPE_OP_COPY_PLANES, 0, &cp}};
        void* streams[2] = {st, st2};
        int failed = -5;
        if (rpe_run_ops(list, 8, streams, 2, &failed) != RPE_OK || failed != -1) { std::printf("rpe_run_ops failed at op %d\n", failed); return 17; }
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(ya.data(), d_ya, ya.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(yc.data(), d_yc, yc.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ya.size(); ++i) if (ya[i] != yc[i]) { std::printf("launch list: the two streams disagree at %zu\n", i); return 18; }
        rpe_copy_planes_args bad = cp; bad.src = nullptr;
        const rpe_op list2[] = {{RPE_OP_COPY_PLANES, 0, &cp}, {RPE_OP_COPY_PLANES, 0, &bad}};
        if (rpe_run_ops(list2, 2, streams, 2, &failed) != RPE_E_BADARG || failed != 1) return 19;
        CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st2));
        std::printf("launch list ok (ABI %d.%d)\n", rpe_abi_version(), rpe_abi_minor());
    }
    std::printf("ABI_SMOKE_OK\n");
    return 0;
}
