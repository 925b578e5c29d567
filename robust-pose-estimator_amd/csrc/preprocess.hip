// Input side of the path: the per-frame preprocessing the reference runs with numpy / cv2 /
// torchvision on the CPU before PoseEstimator sees a frame.
//
// Replaces:
//   dataset/stereo_dataset.py:12-16  mask_specularities  (brightness threshold, AND with the tool mask, 11x11 erosion)
//   dataset/transforms.py:20-39      ResizeStereo        (bilinear / nearest resize that conserves the aspect ratio,
//                                                         then centre crop; torchvision 0.14: antialias off,
//                                                         align_corners=False)
//   dataset/video_dataset.py:59-61, stereo_dataset.py:35-37   uint8 HWC -> float32 CHW
// All of it is HBM-bound byte work: one read of the decoded frame, one write of the network input.
#include "rpe_common.h"

// torch rounds every operation of the bilinear formula separately; HIP's __fmul_rn / __fadd_rn are plain * and + (still
// contractable into FMAs) unless OCML_BASIC_ROUNDED_OPERATIONS is defined, so contraction is switched off for this file.
#pragma clang fp contract(off)

#define ER 5                                   // 11x11 structuring element
#define TW 64
#define TH 16

// out = erode_11x11((r + g + b < thr) & mask).  cv2.erode's default border is +inf: pixels outside the image never erode.
__global__ __launch_bounds__(256) void k_mask_specularities(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask, int h, int w,
                                                            int thr, uint8_t* __restrict__ out) {
    __shared__ uint8_t tile[TH + 2 * ER][TW + 2 * ER + 2];
    __shared__ uint8_t hmin[TH + 2 * ER][TW];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    for (int i = threadIdx.x; i < (TH + 2 * ER) * (TW + 2 * ER); i += blockDim.x) {
        const int ty = i / (TW + 2 * ER), tx = i % (TW + 2 * ER);
        const int y = y0 + ty - ER, x = x0 + tx - ER;
        uint8_t v = 1;
        if (y >= 0 && y < h && x >= 0 && x < w) {
            const uint8_t* p = img + ((size_t)y * w + x) * 3;
            const int s = (int)p[0] + (int)p[1] + (int)p[2];
            v = (s < thr) && (!mask || mask[(size_t)y * w + x] != 0);
        }
        tile[ty][tx] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (TH + 2 * ER) * TW; i += blockDim.x) {
        const int ty = i / TW, tx = i % TW;
        uint8_t m = 1;
#pragma unroll
        for (int d = 0; d <= 2 * ER; ++d) m &= tile[ty][tx + d];
        hmin[ty][tx] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TH * TW; i += blockDim.x) {
        const int ty = i / TW, tx = i % TW;
        const int y = y0 + ty, x = x0 + tx;
        if (y >= h || x >= w) continue;
        uint8_t m = 1;
#pragma unroll
        for (int d = 0; d <= 2 * ER; ++d) m &= hmin[ty + d][tx];
        out[(size_t)y * w + x] = m;
    }
}

// Source position of torch's upsample_bilinear2d (align_corners=False): scale * (dst + 0.5) - 0.5, clamped at 0.
__device__ __forceinline__ void bil(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float src = rn_sub(rn_mul(scale, rn_add((float)dst, 0.5f)), 0.5f);     // as torch rounds it: no FMA
    src = src < 0.0f ? 0.0f : src;
    i0 = (int)src;
    i0 = i0 > in_size - 1 ? in_size - 1 : i0;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = rn_sub(src, (float)i0);
    l0 = rn_sub(1.0f, l1);
}

// out (c, oh, ow) f32 = centre crop (top, left) of bilinear resize of the input to (rh, rw).
// U8HWC: input uint8 (h, w, c) as decoded; else float32 (c, h, w).
template <bool U8HWC>
__global__ __launch_bounds__(256) void k_resize_crop(const void* __restrict__ in, int c, int h, int w, int rh, int rw, int top, int left,
                                                     int oh, int ow, float* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= ow) return;
    const float sh = rn_div((float)h, (float)rh), sw = rn_div((float)w, (float)rw);   // area_pixel_compute_scale<float>
    int y0, y1, x0, x1; float hy0, hy1, wx0, wx1;
    bil(y + top, sh, h, y0, y1, hy0, hy1);
    bil(x + left, sw, w, x0, x1, wx0, wx1);
    for (int ch = 0; ch < c; ++ch) {
        float p00, p01, p10, p11;
        if (U8HWC) {
            const uint8_t* p = (const uint8_t*)in;
            p00 = p[((size_t)y0 * w + x0) * c + ch]; p01 = p[((size_t)y0 * w + x1) * c + ch];
            p10 = p[((size_t)y1 * w + x0) * c + ch]; p11 = p[((size_t)y1 * w + x1) * c + ch];
        } else {
            const float* p = (const float*)in + (size_t)ch * h * w;
            p00 = p[(size_t)y0 * w + x0]; p01 = p[(size_t)y0 * w + x1]; p10 = p[(size_t)y1 * w + x0]; p11 = p[(size_t)y1 * w + x1];
        }
        // torch: h0lambda * (w0lambda * p00 + w1lambda * p01) + h1lambda * (w0lambda * p10 + w1lambda * p11)
        // (one correctly-rounded operation at a time: no FMA contraction, so both input formats give the same bits)
        const float r0 = rn_add(rn_mul(wx0, p00), rn_mul(wx1, p01)), r1 = rn_add(rn_mul(wx0, p10), rn_mul(wx1, p11));
        out[((size_t)ch * oh + y) * ow + x] = rn_add(rn_mul(hy0, r0), rn_mul(hy1, r1));
    }
}

// Nearest resize + centre crop of a one-channel byte mask: torch 'nearest' = floor(dst * scale), scale = in / out in f32.
__global__ __launch_bounds__(256) void k_resize_crop_nearest(const uint8_t* __restrict__ in, int h, int w, int rh, int rw, int top, int left,
                                                             int oh, int ow, uint8_t* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= ow) return;
    const float sh = rn_div((float)h, (float)rh), sw = rn_div((float)w, (float)rw);
    int ys = (int)floorf(rn_mul((float)(y + top), sh)), xs = (int)floorf(rn_mul((float)(x + left), sw));
    ys = ys > h - 1 ? h - 1 : ys; xs = xs > w - 1 ? w - 1 : xs;
    out[(size_t)y * ow + x] = in[(size_t)ys * w + xs];
}

extern "C" int rpe_mask_specularities(const uint8_t* img_hwc, const uint8_t* mask, int h, int w, int sum_threshold, uint8_t* out,
                                      void* stream) {
    if (!img_hwc || !out || h <= 0 || w <= 0) return RPE_E_BADARG;
    hipLaunchKernelGGL(k_mask_specularities, dim3(ceil_div(w, TW), ceil_div(h, TH)), dim3(256), 0, (hipStream_t)stream, img_hwc, mask, h, w,
                       sum_threshold, out);
    return rpe_check_launch();
}

static bool crop_ok(int rh, int rw, int top, int left, int oh, int ow) {
    return rh > 0 && rw > 0 && top >= 0 && left >= 0 && oh > 0 && ow > 0 && top + oh <= rh && left + ow <= rw;
}

extern "C" int rpe_resize_crop(const void* in, int in_is_u8_hwc, int c, int h, int w, int resized_h, int resized_w, int top, int left,
                               int out_h, int out_w, float* out, void* stream) {
    if (!in || !out || c <= 0 || h <= 0 || w <= 0 || !crop_ok(resized_h, resized_w, top, left, out_h, out_w)) return RPE_E_BADARG;
    dim3 grid(ceil_div(out_w, 256), out_h), block(256);
    if (in_is_u8_hwc) hipLaunchKernelGGL(k_resize_crop<true>, grid, block, 0, (hipStream_t)stream, in, c, h, w, resized_h, resized_w, top, left, out_h, out_w, out);
    else hipLaunchKernelGGL(k_resize_crop<false>, grid, block, 0, (hipStream_t)stream, in, c, h, w, resized_h, resized_w, top, left, out_h, out_w, out);
    return rpe_check_launch();
}

extern "C" int rpe_resize_crop_mask(const uint8_t* in, int h, int w, int resized_h, int resized_w, int top, int left, int out_h, int out_w,
                                    uint8_t* out, void* stream) {
    if (!in || !out || h <= 0 || w <= 0 || !crop_ok(resized_h, resized_w, top, left, out_h, out_w)) return RPE_E_BADARG;
    hipLaunchKernelGGL(k_resize_crop_nearest, dim3(ceil_div(out_w, 256), out_h), dim3(256), 0, (hipStream_t)stream, in, h, w, resized_h,
                       resized_w, top, left, out_h, out_w, out);
    return rpe_check_launch();
}

// ---- rectification (dataset/preprocess/stereo_rectify.py:44-51: cv2.remap(img, map1, map2, INTER_NEAREST), default border =
// constant 0).  cv2 rounds the float maps with cvRound (round half to even) and saturates to int16; a source pixel outside the
// image gives 0.  Planar (c,h,w) input of T = u8 / f32, maps (out_h,out_w) f32; HBM-bound gather, one thread per output pixel,
// all channels (the index is shared).
template <typename T>
__global__ __launch_bounds__(256) void k_remap_nearest(const T* __restrict__ src, int c, int h, int w, const float* __restrict__ mapx,
                                                       const float* __restrict__ mapy, int oh, int ow, T* __restrict__ dst) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= ow) return;
    const size_t o = (size_t)y * ow + x;
    const float fx = __builtin_nontemporal_load(mapx + o), fy = __builtin_nontemporal_load(mapy + o);
    // cvRound + saturate_cast<short>: NaN / out-of-range land outside every image
    const float cx = fminf(fmaxf(fx, -32768.0f), 32767.0f), cy = fminf(fmaxf(fy, -32768.0f), 32767.0f);
    const int sx = (fx == fx) ? __float2int_rn(cx) : -32768, sy = (fy == fy) ? __float2int_rn(cy) : -32768;
    const bool in = sx >= 0 && sx < w && sy >= 0 && sy < h;
    const size_t si = in ? (size_t)sy * w + sx : 0;
    for (int ch = 0; ch < c; ++ch) {
        const T v = in ? src[(size_t)ch * h * w + si] : (T)0;
        __builtin_nontemporal_store(v, dst + (size_t)ch * oh * ow + o);
    }
}

extern "C" int rpe_remap_nearest(const void* src, int src_is_u8, int c, int h, int w, const float* mapx, const float* mapy, int out_h, int out_w,
                                 void* dst, void* stream) {
    if (!src || !dst || !mapx || !mapy || c <= 0 || h <= 0 || w <= 0 || out_h <= 0 || out_w <= 0) return RPE_E_BADARG;
    dim3 grid(ceil_div(out_w, 256), out_h), block(256);
    if (src_is_u8) hipLaunchKernelGGL(k_remap_nearest<uint8_t>, grid, block, 0, (hipStream_t)stream, (const uint8_t*)src, c, h, w, mapx, mapy, out_h, out_w, (uint8_t*)dst);
    else hipLaunchKernelGGL(k_remap_nearest<float>, grid, block, 0, (hipStream_t)stream, (const float*)src, c, h, w, mapx, mapy, out_h, out_w, (float*)dst);
    return rpe_check_launch();
}

// Pseudo-rectification (dataset/preprocess/stereo_rectify.py:52-59 pseudo_rectify_2d, used by dataset/rectification.py:55-58 for
// mode='pseudo'): cv2.warpAffine(img, [[1, 0, tx], [0, 1, ty]], (w, h)) with its defaults INTER_LINEAR / BORDER_CONSTANT(0).
// OpenCV 4.x imgwarp.cpp, restated: the matrix is inverted (source = dst - t), source coordinates are fixed point with
// AB_BITS = 10 and rounded to 1/32 pixel -- X = (cvRound(-tx * 1024) + 16 + 1024 x) >> 5, integer part X >> 5, fraction X & 31 --
// and the four taps are blended with the 5-bit bilinear table: integer weights (32-a)(32-b), a(32-b), (32-a)b, ab times 32
// (sum 2^15), result (sum + 2^14) >> 15 for uint8; float images use the table's float weights and a float sum.  Taps outside
// the image contribute the border value 0.  host side passes X0 = cvRound(-tx*1024) + 16, Y0c = the per-row constants' -ty term:
// Y0(y) = cvRound((y - ty) * 1024) + 16 is computed per row in double, as OpenCV does.
template <typename T>
__global__ __launch_bounds__(256) void k_shift_bilinear(const T* __restrict__ src, int c, int h, int w, int X0, double mty, T* __restrict__ dst) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int Y0 = (int)__double2ll_rn(((double)y + mty) * 1024.0) + 16;       // saturate_cast<int>(double) = cvRound: round half to even
    const int X = (X0 + x * 1024) >> 5, Y = Y0 >> 5;                           // arithmetic shifts: floor
    int sx = X >> 5, sy = Y >> 5;
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx); sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);   // saturate_cast<short>
    const int ax = X & 31, ay = Y & 31;
    const bool x0ok = sx >= 0 && sx < w, x1ok = sx + 1 >= 0 && sx + 1 < w, y0ok = sy >= 0 && sy < h, y1ok = sy + 1 >= 0 && sy + 1 < h;
    const size_t o = (size_t)y * w + x;
    for (int ch = 0; ch < c; ++ch) {
        const T* p = src + (size_t)ch * h * w;
        const T p00 = (x0ok && y0ok) ? p[(size_t)sy * w + sx] : (T)0, p01 = (x1ok && y0ok) ? p[(size_t)sy * w + sx + 1] : (T)0;
        const T p10 = (x0ok && y1ok) ? p[(size_t)(sy + 1) * w + sx] : (T)0, p11 = (x1ok && y1ok) ? p[(size_t)(sy + 1) * w + sx + 1] : (T)0;
        if constexpr (sizeof(T) == 1) {
            const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
            const int v = (w00 * (int)p00 + w01 * (int)p01 + w10 * (int)p10 + w11 * (int)p11 + (1 << 14)) >> 15;
            dst[(size_t)ch * h * w + o] = (T)(v < 0 ? 0 : (v > 255 ? 255 : v));
        } else {
            const float fx = (float)ax * (1.0f / 32.0f), fy = (float)ay * (1.0f / 32.0f);
            const float w00 = (1.0f - fy) * (1.0f - fx), w01 = (1.0f - fy) * fx, w10 = fy * (1.0f - fx), w11 = fy * fx;
            dst[(size_t)ch * h * w + o] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(p00, w00), __fmul_rn(p01, w01)), __fmul_rn(p10, w10)), __fmul_rn(p11, w11));
        }
    }
}

extern "C" int rpe_shift_bilinear(const void* src, int src_is_u8, int c, int h, int w, float tx, float ty, void* dst, void* stream) {
    if (!src || !dst || c <= 0 || h <= 0 || w <= 0 || !(tx == tx) || !(ty == ty)) return RPE_E_BADARG;
    // the inverse of [[1,0,tx],[0,1,ty]] in double, as warpAffine computes it: b1 = -tx, b2 = -ty (exact)
    const double mtx = -(double)tx, mty = -(double)ty;
    const double sx = mtx * 1024.0;
    if (!(fabs(sx) < 2.0e9) || !(fabs(mty) < 1.0e6)) return RPE_E_UNSUPPORTED;
    const int X0 = (int)llrint(sx) + 16;                                      // cvRound: round half to even (default FP environment)
    dim3 grid(ceil_div(w, 256), h), block(256);
    if (src_is_u8) hipLaunchKernelGGL(k_shift_bilinear<uint8_t>, grid, block, 0, (hipStream_t)stream, (const uint8_t*)src, c, h, w, X0, mty, (uint8_t*)dst);
    else hipLaunchKernelGGL(k_shift_bilinear<float>, grid, block, 0, (hipStream_t)stream, (const float*)src, c, h, w, X0, mty, (float*)dst);
    return rpe_check_launch();
}
