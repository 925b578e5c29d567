// The two TinyUNet weight heads of PoseNet as one chain of hand-written kernels (both heads and all frames per launch).
//
// Replaces (reference): core/pose/pose_net.py:109-115 -- torch.cat of the 1/8 stacks with the GRU hidden state and the
// context, TinyUNet(264) / TinyUNet(272) (core/unet/unet.py:7-82: three encoder stages conv3x3-BN-ReLU-conv3x3 with 2x2
// max pooling, two decoder stages up-conv2x2 / centre-cropped skip / conv3x3-ReLU-BN-conv3x3, 1x1 head), bilinear resize
// to the image size and nn.Sigmoid -- about 55 library launches per frame pair, with inference-mode batch norm folded.
// The maps are tiny (64x80 down to 9x13 at 640x512), so this is launch-bound work: 15 launches here, no concatenation
// (the first layer and the decoder read their inputs from several sources), batch norm / bias / ReLU in the epilogues.
//
// Parameter blob of one head (rpe_unet_params_floats floats, built by the host from the module's tensors), widths 16/32/64:
//   encoder stage i (cin_i -> c_i):  w1 [cin_i][9][c_i] | scale [c_i] | shift [c_i] (BN folded with conv1's bias) | w2 [c_i][9][c_i] | b2 [c_i]
//   decoder stage j (c -> c/2):      up [c][4][c/2] | upb [c/2] | w1 [c][9][c/2] | b1 | scale | shift (BN after the ReLU) | w2 [c/2][9][c/2] | b2
//   head:                            w [16] | b [1]
// (3x3 weights tap-major per input channel with the output channels contiguous: uniform across a workgroup -> scalar loads.)
#include "rpe_common.h"

#define UT 16                 // output channels per thread (up-convolution; the 3x3 kernel is instantiated for 16 and 4)

struct USrc { const float* p; long long bs; int c, h, w, oy, ox; };        // (b, c, h, w) read at (y + oy, x + ox)
struct UConvHead {
    USrc src[4]; int nsrc;
    const float* w; const float* bias; const float* scale; const float* shift;   // w [cin][9][cout]
    float* out; int cout;
};
struct UConvP { UConvHead hd[2]; int b, ho, wo, relu_first, relu_last; };

// valid 3x3 convolution, thread = output pixel x CT output channels:  v = acc + bias; [ReLU]; v = v * scale + shift; [ReLU]
// (CT = 4 when 16 channels per thread would leave most of the chip idle: one frame pair of sequential tracking).
// The weights of UCK input channels at a time ([ci][9][CT], this workgroup's CT output channels) are staged in LDS by the whole
// workgroup (coalesced, double-buffered) and read back as 16-byte broadcasts: as wave-uniform scalar loads -- 9 x CT per input
// channel, more than the scalar register file holds -- they were a chain of load latencies (the first layer, 264 -> 16 channels
// on 62 x 78 pixels: 214 us at batch 1).
#define UCK 8
template <int CT>
__global__ __launch_bounds__(256) void k_u_conv3(UConvP P) {
    __shared__ __attribute__((aligned(16))) float wsm[2][UCK * 9 * CT];
    const int head = blockIdx.z / P.b, bz = blockIdx.z % P.b;
    const UConvHead& H = P.hd[head];
    const int co0 = blockIdx.y * CT;
    if (co0 >= H.cout) return;                                   // (workgroup-uniform)
    const int tid = threadIdx.x;
    const int p = blockIdx.x * blockDim.x + tid;
    const int npix = P.ho * P.wo;
    const bool ok = p < npix;
    const int y = ok ? p / P.wo : 0, x = ok ? p - (p / P.wo) * P.wo : 0;
    float acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[j] = 0.0f;
    const int cout = H.cout;
    // chunk q of the flattened input-channel axis (all sources back to back) -> its weights: rows (ci * 9 + t) of cout floats, of
    // which this workgroup takes columns [co0, co0 + CT)
    int cin_total = 0;
    for (int s = 0; s < H.nsrc; ++s) cin_total += H.src[s].c;
    const int nchunk = (cin_total + UCK - 1) / UCK;
    constexpr int WN = UCK * 9 * CT, NLD = (WN + 255) / 256;
    float wreg[NLD];
    auto wload = [&](int q) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + 256 * i;                          // e = (ci_local * 9 + t) * CT + j
            const int row = e / CT, j = e - row * CT;
            const int ci = q * UCK + row / 9;
            wreg[i] = (e < WN && ci < cin_total) ? H.w[(size_t)(q * UCK * 9 + row) * cout + co0 + j] : 0.0f;
        }
    };
    auto wstash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + 256 * i;
            if (e < WN) wsm[buf][e] = wreg[i];
        }
    };
    wload(0); wstash(0);
    __syncthreads();
    int q = 0, cbase = 0;                                        // cbase: first flattened channel of the current source
    bool aligned = true;                                         // every source a multiple of UCK channels (the architecture's widths are)
    for (int s = 0; s < H.nsrc; ++s) aligned = aligned && (H.src[s].c % UCK == 0);
    for (int s = 0; s < H.nsrc; ++s) {
        const USrc S = H.src[s];
        const float* base = S.p + (size_t)bz * S.bs + (size_t)(y + S.oy) * S.w + (x + S.ox);
        const size_t plane = (size_t)S.h * S.w;
        if (aligned) {
            // a chunk at a time: its 8 x 9 input values are requested together (72 loads in flight per thread instead of 9: the layer
            // is a latency chain at batch 1), the next chunk's weights travel meanwhile
            for (int ci0 = 0; ci0 < S.c; ci0 += UCK) {
                if (cbase + ci0 > 0) { ++q; __syncthreads(); }
                if (q + 1 < nchunk) wload(q + 1);
                float v[UCK][9];
#pragma unroll
                for (int c = 0; c < UCK; ++c) {
                    const float* ip = base + (size_t)(ci0 + c) * plane;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) v[c][dy * 3 + dx] = ip[dy * S.w + dx];
                }
                if (q + 1 < nchunk) wstash((q + 1) & 1);
                const float* wl = &wsm[q & 1][0];
#pragma unroll
                for (int c = 0; c < UCK; ++c)
#pragma unroll
                    for (int t = 0; t < 9; ++t)
#pragma unroll
                        for (int j = 0; j < CT; j += 4) {
                            const float4 w4 = *(const float4*)(wl + (c * 9 + t) * CT + j);
                            // (explicit fused multiply-adds: left to the compiler's contraction, the CT = 4 and CT = 16 instantiations came out
                            //  with different roundings -- a frame's weights then depended on the batch it was launched in)
                            acc[j] = fmaf(v[c][t], w4.x, acc[j]); acc[j + 1] = fmaf(v[c][t], w4.y, acc[j + 1]);
                            acc[j + 2] = fmaf(v[c][t], w4.z, acc[j + 2]); acc[j + 3] = fmaf(v[c][t], w4.w, acc[j + 3]);
                        }
            }
        } else {
            for (int ci = 0; ci < S.c; ++ci) {
                const int cf = cbase + ci;                        // flattened channel
                if (cf % UCK == 0 && cf > 0) {                    // next chunk (workgroup-uniform: every thread walks the same channels)
                    ++q;
                    __syncthreads();
                }
                if (cf % UCK == 0) {                              // prefetch the chunk after this one while this one is consumed
                    if (q + 1 < nchunk) { wload(q + 1); wstash((q + 1) & 1); }
                }
                const float* ip = base + (size_t)ci * plane;
                float v[9];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) v[dy * 3 + dx] = ip[dy * S.w + dx];
                const float* wl = &wsm[q & 1][(cf % UCK) * 9 * CT];
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int j = 0; j < CT; j += 4) {
                        const float4 w4 = *(const float4*)(wl + t * CT + j);
                        acc[j] = fmaf(v[t], w4.x, acc[j]); acc[j + 1] = fmaf(v[t], w4.y, acc[j + 1]);
                        acc[j + 2] = fmaf(v[t], w4.z, acc[j + 2]); acc[j + 3] = fmaf(v[t], w4.w, acc[j + 3]);
                    }
            }
        }
        cbase += S.c;
    }
    if (!ok) return;
    float* o = H.out + ((size_t)bz * H.cout + co0) * npix + p;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        float v = acc[j] + (H.bias ? H.bias[co0 + j] : 0.0f);
        if (P.relu_first) v = v < 0.0f ? 0.0f : v;
        if (H.scale) v = fmaf(v, H.scale[co0 + j], H.shift[co0 + j]);
        if (P.relu_last) v = v < 0.0f ? 0.0f : v;
        o[(size_t)j * npix] = v;
    }
}

// 2x2 max pooling (floor), planes = heads * b * c
__global__ void k_u_pool(const float* __restrict__ in, float* __restrict__ out, int h, int w, long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int ho = h / 2, wo = w / 2;
    const int x = (int)(e % wo), y = (int)((e / wo) % ho);
    const long long pl = e / ((long long)wo * ho);
    const float* s = in + pl * h * w + (size_t)(2 * y) * w + 2 * x;
    out[e] = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[w], s[w + 1]));
}

struct UUpHead { const float* in; const float* w; const float* bias; float* out; };   // w [cin][4][cout]
struct UUpP { UUpHead hd[2]; int b, cin, cout, h, w; };
// ConvTranspose2d(k = 2, stride 2): thread = input pixel -> its 2x2 output pixels x UT channels
__global__ __launch_bounds__(256) void k_u_upconv(UUpP P) {
    const int head = blockIdx.z / P.b, bz = blockIdx.z % P.b;
    const UUpHead& H = P.hd[head];
    const int co0 = blockIdx.y * UT;
    const int p = blockIdx.x * blockDim.x + threadIdx.x, npix = P.h * P.w;
    if (p >= npix) return;
    float acc[4][UT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < UT; ++j) acc[t][j] = 0.0f;
    const float* ip = H.in + (size_t)bz * P.cin * npix + p;
    const float* wp = H.w + co0;
    for (int ci = 0; ci < P.cin; ++ci) {
        const float v = ip[(size_t)ci * npix];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < UT; ++j) acc[t][j] += v * wp[(size_t)t * P.cout + j];
        wp += (size_t)4 * P.cout;
    }
    const int y = p / P.w, x = p - y * P.w, wo = 2 * P.w;
    const size_t opl = (size_t)4 * npix;
    float* o = H.out + ((size_t)bz * P.cout + co0) * opl + (size_t)(2 * y) * wo + 2 * x;
#pragma unroll
    for (int j = 0; j < UT; ++j) {
        const float bi = H.bias[co0 + j];
        o[j * opl] = acc[0][j] + bi; o[j * opl + 1] = acc[1][j] + bi;
        o[j * opl + wo] = acc[2][j] + bi; o[j * opl + wo + 1] = acc[3][j] + bi;
    }
}

// 1x1 head (16 -> 1) + bias
struct UHeadP { const float* in[2]; const float* w[2]; float* out[2]; int b, npix; };
__global__ void k_u_head(UHeadP P) {
    const int head = blockIdx.z / P.b, bz = blockIdx.z % P.b;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.npix) return;
    const float* ip = P.in[head] + (size_t)bz * 16 * P.npix + p;
    const float* w = P.w[head];
    float a = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) a += ip[(size_t)c * P.npix] * w[c];
    P.out[head][(size_t)bz * P.npix + p] = a + w[16];
}

// F.interpolate(mode='bilinear', align_corners=False) of the (h, w) map to (H, W), then sigmoid
struct UResP { const float* in[2]; float* out[2]; int b, h, w, H, W; float sy, sx; };
__global__ void k_u_resize_sigmoid(UResP P) {
    const int head = blockIdx.z / P.b, bz = blockIdx.z % P.b;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)P.H * P.W) return;
    const int oy = (int)(e / P.W), ox = (int)(e - (long long)oy * P.W);
    // area_pixel_compute_source_index: src = scale * (dst + 0.5) - 0.5, clamped below at 0 (float arithmetic, as torch's opmath)
    float fy = rn_sub(rn_mul(P.sy, rn_add((float)oy, 0.5f)), 0.5f), fx = rn_sub(rn_mul(P.sx, rn_add((float)ox, 0.5f)), 0.5f);
    fy = fy < 0.0f ? 0.0f : fy; fx = fx < 0.0f ? 0.0f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < P.h - 1 ? 1 : 0), x1 = x0 + (x0 < P.w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
    const float* s = P.in[head] + (size_t)bz * P.h * P.w;
    const float v = hy * (hx * s[y0 * P.w + x0] + lx * s[y0 * P.w + x1]) + ly * (hx * s[y1 * P.w + x0] + lx * s[y1 * P.w + x1]);
    P.out[head][(size_t)bz * P.H * P.W + e] = 1.0f / (1.0f + expf(-v));
}

// ------------------------------------------------------------------------------------------------ host side
static const int WIDTHS[3] = {16, 32, 64};

struct Geo { int h[3], w[3], hs[3], ws[3]; int uh[2], uw[2], dh[2], dw[2]; };   // encoder conv1/conv2 (skip) sizes; decoder sizes
static bool unet_geo(int h8, int w8, Geo& g) {
    int h = h8, w = w8;
    for (int i = 0; i < 3; ++i) {
        g.h[i] = h - 2; g.w[i] = w - 2; g.hs[i] = h - 4; g.ws[i] = w - 4;
        if (g.hs[i] < 2 || g.ws[i] < 2) return false;          // the reference pools after the last stage too: >= 44 at 1/8 scale
        h = g.hs[i] / 2; w = g.ws[i] / 2;
    }
    int ch = g.hs[2], cw = g.ws[2];
    for (int j = 0; j < 2; ++j) {
        g.uh[j] = 2 * ch; g.uw[j] = 2 * cw;                    // up-conv output; the skip (stage 1 - j) is centre-cropped to it
        if (g.uh[j] > g.hs[1 - j] || g.uw[j] > g.ws[1 - j]) return false;
        g.dh[j] = g.uh[j] - 4; g.dw[j] = g.uw[j] - 4;
        if (g.dh[j] < 1 || g.dw[j] < 1) return false;
        ch = g.dh[j]; cw = g.dw[j];
    }
    return true;
}

static void launch_conv3(const UConvP& P, int cout, int b, hipStream_t s) {
    const int tiles = ceil_div(P.ho * P.wo, 256);
#ifndef U_CONV3_MIN_WG
#define U_CONV3_MIN_WG 512
#endif
    if ((long long)tiles * (cout / 16) * 2 * b < U_CONV3_MIN_WG)
        hipLaunchKernelGGL(k_u_conv3<4>, dim3(tiles, cout / 4, 2 * b), dim3(256), 0, s, P);
    else
        hipLaunchKernelGGL(k_u_conv3<16>, dim3(tiles, cout / 16, 2 * b), dim3(256), 0, s, P);
}

extern "C" size_t rpe_unet_params_floats(int in_channels) {
    if (in_channels <= 0) return 0;
    size_t n = 0;
    int cin = in_channels;
    for (int i = 0; i < 3; ++i) { const int c = WIDTHS[i]; n += (size_t)cin * 9 * c + 2 * c + (size_t)c * 9 * c + c; cin = c; }
    for (int j = 0; j < 2; ++j) { const int c = WIDTHS[2 - j], c2 = c / 2; n += (size_t)c * 4 * c2 + c2 + (size_t)c * 9 * c2 + 3 * c2 + (size_t)c2 * 9 * c2 + c2; }
    return n + 17;
}

// floats of scratch for nheads heads and b frames
static size_t unet_ws_floats(const Geo& g, int b, int nheads) {
    size_t n = 0;
    for (int i = 0; i < 3; ++i) n += (size_t)WIDTHS[i] * (g.h[i] * g.w[i] + g.hs[i] * g.ws[i] + (i < 2 ? (g.hs[i] / 2) * (g.ws[i] / 2) : 0));
    for (int j = 0; j < 2; ++j) { const int c2 = WIDTHS[2 - j] / 2; n += (size_t)c2 * (g.uh[j] * g.uw[j] + (g.dh[j] + 2) * (g.dw[j] + 2) + g.dh[j] * g.dw[j]); }
    n += (size_t)g.dh[1] * g.dw[1];
    return n * b * nheads;
}

extern "C" size_t rpe_unet_workspace_bytes(int b, int h8, int w8) {
    Geo g;
    if (b <= 0 || !unet_geo(h8, w8, g)) return 0;
    return unet_ws_floats(g, b, 2) * sizeof(float) + 256;
}

extern "C" int rpe_unet_heads(const float* inp1, const float* inp2, const float* hidden, const float* context, long long hidden_bs,
                              long long context_bs, const float* params2d, const float* params3d, int b, int h8, int w8, int H, int W,
                              float* out2d, float* out3d, void* workspace, void* stream) {
    Geo g;
    if (!inp1 || !inp2 || !hidden || !context || !params2d || !params3d || !out2d || !out3d || !workspace || b <= 0 || H <= 0 || W <= 0) return RPE_E_BADARG;
    if (!unet_geo(h8, w8, g)) return RPE_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)(((uintptr_t)workspace + 255) / 256 * 256);
    const float* prm[2] = {params2d, params3d};
    const int hw8 = h8 * w8;
    auto take = [&](size_t per_frame) { float* p = ws; ws += per_frame * b; return p; };     // one buffer per head
    const float* cur[2] = {nullptr, nullptr};
    float* skip[2][3];
    int cin_first[2] = {8 + 128 + 128, 8 + 8 + 128 + 128};
    int ch = h8, cw = w8;
    for (int i = 0; i < 3; ++i) {
        const int c = WIDTHS[i];
        UConvP A{}; A.b = b; A.ho = g.h[i]; A.wo = g.w[i]; A.relu_first = 0; A.relu_last = 1;
        UConvP B{}; B.b = b; B.ho = g.hs[i]; B.wo = g.ws[i]; B.relu_first = 0; B.relu_last = 0;
        float* mid[2]; float* pooled[2] = {nullptr, nullptr};
        for (int hd = 0; hd < 2; ++hd) {
            const int cin = i == 0 ? cin_first[hd] : WIDTHS[i - 1];
            const float* w1 = prm[hd]; const float* sc = w1 + (size_t)cin * 9 * c; const float* sh = sc + c;
            const float* w2 = sh + c; const float* b2 = w2 + (size_t)c * 9 * c;
            prm[hd] = b2 + c;
            mid[hd] = take((size_t)c * g.h[i] * g.w[i]); skip[hd][i] = take((size_t)c * g.hs[i] * g.ws[i]);
            UConvHead& a = A.hd[hd];
            if (i == 0) {                                    // cat((inp1[, inp2], hidden, context)) without the cat
                int k = 0;
                a.src[k++] = USrc{inp1, (long long)8 * hw8, 8, h8, w8, 0, 0};
                if (hd == 1) a.src[k++] = USrc{inp2, (long long)8 * hw8, 8, h8, w8, 0, 0};
                a.src[k++] = USrc{hidden, hidden_bs, 128, h8, w8, 0, 0};
                a.src[k++] = USrc{context, context_bs, 128, h8, w8, 0, 0};
                a.nsrc = k;
            } else { a.src[0] = USrc{cur[hd], (long long)cin * ch * cw, cin, ch, cw, 0, 0}; a.nsrc = 1; }
            a.w = w1; a.bias = nullptr; a.scale = sc; a.shift = sh; a.out = mid[hd]; a.cout = c;
            UConvHead& bb = B.hd[hd];
            bb.src[0] = USrc{mid[hd], (long long)c * g.h[i] * g.w[i], c, g.h[i], g.w[i], 0, 0}; bb.nsrc = 1;
            bb.w = w2; bb.bias = b2; bb.scale = nullptr; bb.shift = nullptr; bb.out = skip[hd][i]; bb.cout = c;
        }
        launch_conv3(A, c, b, s);
        launch_conv3(B, c, b, s);
        if (i < 2) {
            ch = g.hs[i] / 2; cw = g.ws[i] / 2;
            for (int hd = 0; hd < 2; ++hd) {
                pooled[hd] = take((size_t)c * ch * cw);
                const long long total = (long long)b * c * ch * cw;
                hipLaunchKernelGGL(k_u_pool, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const float*)skip[hd][i], pooled[hd], g.hs[i], g.ws[i], total);
                cur[hd] = pooled[hd];
            }
        } else { cur[0] = skip[0][2]; cur[1] = skip[1][2]; ch = g.hs[2]; cw = g.ws[2]; }
    }
    for (int j = 0; j < 2; ++j) {
        const int c = WIDTHS[2 - j], c2 = c / 2, si = 1 - j;
        UUpP U{}; U.b = b; U.cin = c; U.cout = c2; U.h = ch; U.w = cw;
        UConvP A{}; A.b = b; A.ho = g.dh[j] + 2; A.wo = g.dw[j] + 2; A.relu_first = 1; A.relu_last = 0;
        UConvP B{}; B.b = b; B.ho = g.dh[j]; B.wo = g.dw[j];
        for (int hd = 0; hd < 2; ++hd) {
            const float* up = prm[hd]; const float* upb = up + (size_t)c * 4 * c2; const float* w1 = upb + c2;
            const float* b1 = w1 + (size_t)c * 9 * c2; const float* sc = b1 + c2; const float* sh = sc + c2;
            const float* w2 = sh + c2; const float* b2 = w2 + (size_t)c2 * 9 * c2;
            prm[hd] = b2 + c2;
            float* upo = take((size_t)c2 * g.uh[j] * g.uw[j]);
            float* mid = take((size_t)c2 * A.ho * A.wo);
            float* dout = take((size_t)c2 * g.dh[j] * g.dw[j]);
            U.hd[hd] = UUpHead{cur[hd], up, upb, upo};
            UConvHead& a = A.hd[hd];
            a.src[0] = USrc{upo, (long long)c2 * g.uh[j] * g.uw[j], c2, g.uh[j], g.uw[j], 0, 0};
            a.src[1] = USrc{skip[hd][si], (long long)c2 * g.hs[si] * g.ws[si], c2, g.hs[si], g.ws[si], (g.hs[si] - g.uh[j]) / 2, (g.ws[si] - g.uw[j]) / 2};
            a.nsrc = 2; a.w = w1; a.bias = b1; a.scale = sc; a.shift = sh; a.out = mid; a.cout = c2;
            UConvHead& bb = B.hd[hd];
            bb.src[0] = USrc{mid, (long long)c2 * A.ho * A.wo, c2, A.ho, A.wo, 0, 0}; bb.nsrc = 1;
            bb.w = w2; bb.bias = b2; bb.scale = nullptr; bb.shift = nullptr; bb.out = dout; bb.cout = c2;
            cur[hd] = dout;
        }
        hipLaunchKernelGGL(k_u_upconv, dim3(ceil_div(ch * cw, 256), c2 / UT, 2 * b), dim3(256), 0, s, U);
        launch_conv3(A, c2, b, s);
        launch_conv3(B, c2, b, s);
        ch = g.dh[j]; cw = g.dw[j];
    }
    UHeadP Hp{}; Hp.b = b; Hp.npix = ch * cw;
    UResP R{}; R.b = b; R.h = ch; R.w = cw; R.H = H; R.W = W; R.sy = (float)ch / (float)H; R.sx = (float)cw / (float)W;
    float* hm[2];
    for (int hd = 0; hd < 2; ++hd) {
        hm[hd] = take((size_t)ch * cw);
        Hp.in[hd] = cur[hd]; Hp.w[hd] = prm[hd]; Hp.out[hd] = hm[hd];
        R.in[hd] = hm[hd];
    }
    R.out[0] = out2d; R.out[1] = out3d;
    hipLaunchKernelGGL(k_u_head, dim3(ceil_div(ch * cw, 256), 1, 2 * b), dim3(256), 0, s, Hp);
    hipLaunchKernelGGL(k_u_resize_sigmoid, dim3(ceil_div((long long)H * W, 256), 1, 2 * b), dim3(256), 0, s, R);
    return rpe_check_launch();
}
