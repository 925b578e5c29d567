// Stereo depth, back-projection, flow warps and the 1/8 stacks of the weight heads, fused.
//
// Replaces (reference paths): core/pose/pose_net.py:73-79 (disparity -> depth, validity, proj :121-125),
// :104-108 (three bilinear remaps + one nearest remap, core/interpol/flow_utils.py:4-26) and :110-113
// (bilinear x0.125 of two 8-channel stacks).  The reference materialises pcl2, warped image and warped
// stereo flow at full resolution only to down-sample two of them again; here
//   k_geom_full  : one thread per pixel -> depth2, mask2&valid, pcl1, warped pcl2 (taps rebuilt from the
//                  stereo disparity on the fly), warped mask.             HBM-bound, ~60 B/pixel.
//   k_geom_down8 : one thread per 1/8 pixel -> both 8-channel stacks from the 2x2 centre pixels of each 8x8
//                  cell (bilinear x0.125 with align_corners=False samples exactly 8i+3.5).
// Sampling positions follow torch's grid_sample arithmetic operation by operation (no FMA contraction), so
// the integer tap indices are identical to the reference's; rpe_warp_taps exposes them for the tests.
#include "rpe_common.h"
#include "sampling.h"

struct Kinv3 { float m[9]; };

// 3x3 inverse by adjugate, evaluated in f64 and rounded once (torch.linalg.inv is LU in f32: ~1 ulp apart).
__device__ __forceinline__ Kinv3 invert_k(const float* K) {
    double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
    double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    double det = a * A + b * B + c * C;
    double r = 1.0 / det;
    Kinv3 o;
    o.m[0] = (float)(A * r); o.m[1] = (float)(-(b * i - c * h) * r); o.m[2] = (float)((b * f - c * e) * r);
    o.m[3] = (float)(B * r); o.m[4] = (float)((a * i - c * g) * r);  o.m[5] = (float)(-(a * f - c * d) * r);
    o.m[6] = (float)(C * r); o.m[7] = (float)(-(a * h - b * g) * r); o.m[8] = (float)((a * e - b * d) * r);
    return o;
}

// un-normalised grid_sample position of flow_utils.py:9-11 for pixel index idx displaced by flow
__device__ __forceinline__ float sample_pos(float flow, int idx, int size) {
    return rt_pos(rn_add(flow, (float)idx), size);
}

__device__ __forceinline__ float depth_from_disp(float b, float sfx, bool& valid) {
    float d = rn_div(b, -sfx);                 // pose_net.py:73
    valid = (d > 0.0f) && (d <= 1.0f);            // :74
    return valid ? d : 1.0f;                      // :75
}

// back-projected ray K^-1 [x+.5, y+.5, 1]
__device__ __forceinline__ void ray(const Kinv3& Ki, int x, int y, float& rx, float& ry, float& rz) {
    float px = (float)x + 0.5f, py = (float)y + 0.5f;
    rx = Ki.m[0] * px + Ki.m[1] * py + Ki.m[2];
    ry = Ki.m[3] * px + Ki.m[4] * py + Ki.m[5];
    rz = Ki.m[6] * px + Ki.m[7] * py + Ki.m[8];
}

struct Taps { int x0, y0; float nw, ne, sw, se; };

__device__ __forceinline__ Taps bilinear_taps(float ix, float iy) {
    Taps t;
    float fx0, fy0;
    t.x0 = safe_floor(ix, fx0); t.y0 = safe_floor(iy, fy0);
    float fx1 = fx0 + 1.0f, fy1 = fy0 + 1.0f;
    t.nw = (fx1 - ix) * (fy1 - iy);
    t.ne = (ix - fx0) * (fy1 - iy);
    t.sw = (fx1 - ix) * (iy - fy0);
    t.se = (ix - fx0) * (iy - fy0);
    return t;
}

__device__ __forceinline__ bool inb(int x, int y, int w, int h) { return x >= 0 && x < w && y >= 0 && y < h; }

// bilinear sample of one plane (zero padding)
__device__ __forceinline__ float sample_plane(const float* p, const Taps& t, int w, int h) {
    float acc = 0.0f;
    if (inb(t.x0, t.y0, w, h)) acc += p[(size_t)t.y0 * w + t.x0] * t.nw;
    if (inb(t.x0 + 1, t.y0, w, h)) acc += p[(size_t)t.y0 * w + t.x0 + 1] * t.ne;
    if (inb(t.x0, t.y0 + 1, w, h)) acc += p[(size_t)(t.y0 + 1) * w + t.x0] * t.sw;
    if (inb(t.x0 + 1, t.y0 + 1, w, h)) acc += p[(size_t)(t.y0 + 1) * w + t.x0 + 1] * t.se;
    return acc;
}

// bilinear sample of the (never materialised) cloud pcl2 = depth2 * ray
__device__ __forceinline__ void sample_cloud(const float* sf2x, float b, const Kinv3& Ki, const Taps& t, int w, int h,
                                             float& ox, float& oy, float& oz) {
    ox = oy = oz = 0.0f;
    const int dx[4] = {0, 1, 0, 1}, dy[4] = {0, 0, 1, 1};
    const float wt[4] = {t.nw, t.ne, t.sw, t.se};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int xx = t.x0 + dx[k], yy = t.y0 + dy[k];
        if (inb(xx, yy, w, h)) {
            bool v;
            float d = depth_from_disp(b, sf2x[(size_t)yy * w + xx], v);
            float rx, ry, rz;
            ray(Ki, xx, yy, rx, ry, rz);
            ox += (d * rx) * wt[k]; oy += (d * ry) * wt[k]; oz += (d * rz) * wt[k];
        }
    }
}

__global__ __launch_bounds__(256) void k_geom_full(const float* __restrict__ sflow2, const float* __restrict__ tflow,
                                                   const float* __restrict__ baseline, const float* __restrict__ K,
                                                   const float* __restrict__ depth1, const uint8_t* __restrict__ mask2,
                                                   int h, int w, float* __restrict__ depth2, uint8_t* __restrict__ mask2v,
                                                   float* __restrict__ pcl1, float* __restrict__ pcl2w,
                                                   uint8_t* __restrict__ mask2w, float* __restrict__ pcl2) {
    const int row = blockIdx.y;
    const size_t hw = (size_t)h * w;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const int y = (int)(p / w), x = (int)(p - (size_t)y * w);
    const Kinv3 Ki = invert_k(K + (size_t)row * 9);
    const float b = baseline[row];
    const float* sf2x = sflow2 + (size_t)row * 2 * hw;
    const uint8_t* m2 = mask2 + (size_t)row * hw;

    bool valid;
    float d2 = depth_from_disp(b, sf2x[p], valid);
    depth2[(size_t)row * hw + p] = d2;
    mask2v[(size_t)row * hw + p] = (uint8_t)((m2[p] != 0) && valid);     // pose_net.py:77
    float rx, ry, rz;
    ray(Ki, x, y, rx, ry, rz);
    float d1 = depth1[(size_t)row * hw + p];
    float* o1 = pcl1 + (size_t)row * 3 * hw + p;
    o1[0] = d1 * rx; o1[hw] = d1 * ry; o1[2 * hw] = d1 * rz;             // proj :121-125
    if (pcl2) {
        float* o2 = pcl2 + (size_t)row * 3 * hw + p;
        o2[0] = d2 * rx; o2[hw] = d2 * ry; o2[2 * hw] = d2 * rz;
    }
    // warps by the temporal flow (flow_utils.py:4-26)
    const float* tf = tflow + (size_t)row * 2 * hw;
    float ix = sample_pos(tf[p], x, w), iy = sample_pos(tf[hw + p], y, h);
    Taps t = bilinear_taps(ix, iy);
    float cx, cy, cz;
    sample_cloud(sf2x, b, Ki, t, w, h, cx, cy, cz);
    float* ow = pcl2w + (size_t)row * 3 * hw + p;
    ow[0] = cx; ow[hw] = cy; ow[2 * hw] = cz;
    float tmpf;
    int xn = safe_floor(rintf(ix), tmpf), yn = safe_floor(rintf(iy), tmpf);   // nearest: round half to even
    uint8_t mw = 0;
    if (inb(xn, yn, w, h)) {
        bool vn;
        depth_from_disp(b, sf2x[(size_t)yn * w + xn], vn);
        mw = (uint8_t)((m2[(size_t)yn * w + xn] != 0) && vn);            // valid_mapping & mask (pose_net.py:107-108)
    }
    mask2w[(size_t)row * hw + p] = mw;
}

// F.interpolate(scale_factor=0.125, bilinear, align_corners=False): output (i,j) = mean of the 2x2 pixels
// (8i+3..4, 8j+3..4) with weights .5/.5, evaluated as torch does: .5*(.5*a+.5*b) + .5*(.5*c+.5*d).
__device__ __forceinline__ float down4(float a, float b, float c, float d) {
    return 0.5f * (0.5f * a + 0.5f * b) + 0.5f * (0.5f * c + 0.5f * d);
}

__global__ __launch_bounds__(256) void k_geom_down8(const float* __restrict__ sflow2, const float* __restrict__ tflow,
                                                    const float* __restrict__ image1l, const float* __restrict__ image2l,
                                                    const float* __restrict__ sflow1, const float* __restrict__ pcl1,
                                                    const float* __restrict__ pcl2w, int h, int w,
                                                    float* __restrict__ inp1, float* __restrict__ inp2) {
    const int row = blockIdx.y;
    const int h8 = h / 8, w8 = w / 8;
    const size_t hw = (size_t)h * w, hw8 = (size_t)h8 * w8;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= hw8) return;
    const int i = (int)(q / w8), j = (int)(q - (size_t)i * w8);
    const float* tf = tflow + (size_t)row * 2 * hw;
    float v1[8][4], v2[8][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = 8 * i + 3 + (k >> 1), x = 8 * j + 3 + (k & 1);
        const size_t p = (size_t)y * w + x;
        v1[0][k] = sflow1[(size_t)row * 2 * hw + p]; v1[1][k] = sflow1[(size_t)row * 2 * hw + hw + p];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v1[2 + c][k] = image1l[((size_t)row * 3 + c) * hw + p];
            v1[5 + c][k] = pcl1[((size_t)row * 3 + c) * hw + p];
            v2[5 + c][k] = pcl2w[((size_t)row * 3 + c) * hw + p];
        }
        float ix = sample_pos(tf[p], x, w), iy = sample_pos(tf[hw + p], y, h);
        Taps t = bilinear_taps(ix, iy);
        v2[0][k] = sample_plane(sflow2 + (size_t)row * 2 * hw, t, w, h);
        v2[1][k] = sample_plane(sflow2 + (size_t)row * 2 * hw + hw, t, w, h);
#pragma unroll
        for (int c = 0; c < 3; ++c) v2[2 + c][k] = sample_plane(image2l + ((size_t)row * 3 + c) * hw, t, w, h);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        inp1[((size_t)row * 8 + c) * hw8 + q] = down4(v1[c][0], v1[c][1], v1[c][2], v1[c][3]);
        inp2[((size_t)row * 8 + c) * hw8 + q] = down4(v2[c][0], v2[c][1], v2[c][2], v2[c][3]);
    }
}

__global__ void k_flow2depth(const float* __restrict__ sflow, const float* __restrict__ baseline, int h, int w,
                             float* __restrict__ depth, uint8_t* __restrict__ valid) {
    const int row = blockIdx.y;
    const size_t hw = (size_t)h * w;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    bool v;
    float d = depth_from_disp(baseline[row], sflow[(size_t)row * 2 * hw + p], v);
    depth[(size_t)row * hw + p] = d;
    valid[(size_t)row * hw + p] = (uint8_t)v;
}

__global__ void k_warp_taps(const float* __restrict__ flow, int h, int w, int32_t* x0, int32_t* y0, int32_t* xn, int32_t* yn) {
    const int row = blockIdx.y;
    const size_t hw = (size_t)h * w;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const int y = (int)(p / w), x = (int)(p - (size_t)y * w);
    float ix = sample_pos(flow[(size_t)row * 2 * hw + p], x, w), iy = sample_pos(flow[(size_t)row * 2 * hw + hw + p], y, h);
    size_t o = (size_t)row * hw + p;
    float tmp;
    x0[o] = safe_floor(ix, tmp); y0[o] = safe_floor(iy, tmp);
    xn[o] = safe_floor(rintf(ix), tmp); yn[o] = safe_floor(rintf(iy), tmp);
}

extern "C" int rpe_depth_backproject_warp(const float* stereo_flow2, const float* time_flow, const float* baseline,
                                          const float* K, const float* depth1, const float* image1l, const float* image2l,
                                          const float* stereo_flow1, const uint8_t* mask2, int n, int h, int w,
                                          float* depth2, uint8_t* mask2_valid, float* pcl1, float* pcl2w, uint8_t* mask2w,
                                          float* inp1, float* inp2, float* pcl2, void* stream) {
    if (!stereo_flow2 || !time_flow || !baseline || !K || !depth1 || !image1l || !image2l || !stereo_flow1 || !mask2 ||
        !depth2 || !mask2_valid || !pcl1 || !pcl2w || !mask2w || !inp1 || !inp2 || n <= 0 || h <= 0 || w <= 0)
        return RPE_E_BADARG;
    if (h % 8 || w % 8) return RPE_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    size_t hw = (size_t)h * w;
    hipLaunchKernelGGL(k_geom_full, dim3(ceil_div(hw, 256), n), dim3(256), 0, s, stereo_flow2, time_flow, baseline, K, depth1,
                       mask2, h, w, depth2, mask2_valid, pcl1, pcl2w, mask2w, pcl2);
    hipLaunchKernelGGL(k_geom_down8, dim3(ceil_div(hw / 64, 256), n), dim3(256), 0, s, stereo_flow2, time_flow, image1l,
                       image2l, stereo_flow1, (const float*)pcl1, (const float*)pcl2w, h, w, inp1, inp2);
    return rpe_check_launch();
}

extern "C" int rpe_flow2depth(const float* stereo_flow, const float* baseline, int n, int h, int w, float* depth,
                              uint8_t* valid, void* stream) {
    if (!stereo_flow || !baseline || !depth || !valid || n <= 0 || h <= 0 || w <= 0) return RPE_E_BADARG;
    hipLaunchKernelGGL(k_flow2depth, dim3(ceil_div((size_t)h * w, 256), n), dim3(256), 0, (hipStream_t)stream, stereo_flow,
                       baseline, h, w, depth, valid);
    return rpe_check_launch();
}

extern "C" int rpe_warp_taps(const float* flow, int n, int h, int w, int32_t* x0, int32_t* y0, int32_t* xn, int32_t* yn,
                             void* stream) {
    if (!flow || !x0 || !y0 || !xn || !yn || n <= 0 || h <= 0 || w <= 0) return RPE_E_BADARG;
    hipLaunchKernelGGL(k_warp_taps, dim3(ceil_div((size_t)h * w, 256), n), dim3(256), 0, (hipStream_t)stream, flow, h, w,
                       x0, y0, xn, yn);
    return rpe_check_launch();
}
