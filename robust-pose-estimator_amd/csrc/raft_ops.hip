// Element-wise halves of RAFT's SepConvGRU fused around the convolutions, and the convex 8x up-sampling.
//
// Replaces (reference's RAFT submodule): core/RAFT/core/update.py SepConvGRU.forward
//     z = sigmoid(convz(hx)); r = sigmoid(convr(hx)); q = tanh(convq(cat[r*h, x])); h = (1-z)*h + z*q
// and core/RAFT/core/raft.py RAFT.upsample_flow (softmax over the 9 neighbours, weighted sum of 8*flow).
// On the tuned route these gate kernels are the epilogues of the convolutions (conv_wino1d.hip, conv.hip); stand-alone they serve the
// generic route (conv_direct.hip: map sizes the tuned kernels refuse) and remove the sigmoid / mul / cat / tanh / blend round trips
// through HBM between the convolutions.
#include "rpe_common.h"

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// zr_pre (b,2c,hw) ; h (b, h_ch, hw) channels [0,c) ; z_out (b,c,hw) ; rh_out (b, rh_ch, hw) channels [0,c)
template <int VEC>
__global__ __launch_bounds__(256) void k_gates_zr(const float* __restrict__ zr, const float* __restrict__ bias,
                                                  const float* __restrict__ add, const float* __restrict__ h, int c, int hw,
                                                  int h_ch, float* __restrict__ z_out, float* __restrict__ rh, int rh_ch) {
    const int bz = blockIdx.y;
    const size_t per = (size_t)c * hw;
    for (size_t e = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC; e < per; e += (size_t)gridDim.x * blockDim.x * VEC) {
        const int ch = (int)(e / hw);                       // hw % VEC == 0: all VEC elements share the channel
        const float bzv = bias ? bias[ch] : 0.0f, brv = bias ? bias[c + ch] : 0.0f;
        const float* zp = zr + (size_t)bz * 2 * per + e;
        const float* rp = zp + per;
        const float* hp = h + (size_t)bz * h_ch * hw + e;
        float* zo = z_out + (size_t)bz * per + e;
        float* ro = rh + (size_t)bz * rh_ch * hw + e;
        if (VEC == 4) {
            float4 zv = *(const float4*)zp, rv = *(const float4*)rp, hv = *(const float4*)hp;
            if (add) {
                const float4 az = *(const float4*)(add + (size_t)bz * 2 * per + e), ar = *(const float4*)(add + (size_t)bz * 2 * per + per + e);
                zv.x += az.x; zv.y += az.y; zv.z += az.z; zv.w += az.w; rv.x += ar.x; rv.y += ar.y; rv.z += ar.z; rv.w += ar.w;
            }
            float4 zz = make_float4(sigmoidf_(zv.x + bzv), sigmoidf_(zv.y + bzv), sigmoidf_(zv.z + bzv), sigmoidf_(zv.w + bzv));
            float4 rr = make_float4(sigmoidf_(rv.x + brv) * hv.x, sigmoidf_(rv.y + brv) * hv.y, sigmoidf_(rv.z + brv) * hv.z,
                                    sigmoidf_(rv.w + brv) * hv.w);
            *(float4*)zo = zz; *(float4*)ro = rr;
        } else {
            const float az = add ? add[(size_t)bz * 2 * per + e] : 0.0f, ar = add ? add[(size_t)bz * 2 * per + per + e] : 0.0f;
            zo[0] = sigmoidf_(zp[0] + az + bzv); ro[0] = sigmoidf_(rp[0] + ar + brv) * hp[0];
        }
    }
}

// h_out = (1 - z) * h + z * tanh(q_pre)
template <int VEC>
__global__ __launch_bounds__(256) void k_gates_h(const float* __restrict__ z, const float* __restrict__ q,
                                                 const float* __restrict__ bias, const float* __restrict__ add, const float* h, int c,
                                                 int hw, int h_ch, float* h_out, int ho_ch) {
    const int bz = blockIdx.y;
    const size_t per = (size_t)c * hw;
    for (size_t e = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC; e < per; e += (size_t)gridDim.x * blockDim.x * VEC) {
        const float bq = bias ? bias[(int)(e / hw)] : 0.0f;
        const float* zp = z + (size_t)bz * per + e;
        const float* qp = q + (size_t)bz * per + e;
        const float* hp = h + (size_t)bz * h_ch * hw + e;
        float* ho = h_out + (size_t)bz * ho_ch * hw + e;
        if (VEC == 4) {
            float4 zv = *(const float4*)zp, qv = *(const float4*)qp, hv = *(const float4*)hp;
            if (add) { const float4 aq = *(const float4*)(add + (size_t)bz * per + e); qv.x += aq.x; qv.y += aq.y; qv.z += aq.z; qv.w += aq.w; }
            float4 o;
            o.x = (1.0f - zv.x) * hv.x + zv.x * tanhf(qv.x + bq); o.y = (1.0f - zv.y) * hv.y + zv.y * tanhf(qv.y + bq);
            o.z = (1.0f - zv.z) * hv.z + zv.z * tanhf(qv.z + bq); o.w = (1.0f - zv.w) * hv.w + zv.w * tanhf(qv.w + bq);
            *(float4*)ho = o;
        } else {
            ho[0] = (1.0f - zp[0]) * hp[0] + zp[0] * tanhf(qp[0] + (add ? add[(size_t)bz * per + e] : 0.0f) + bq);
        }
    }
}

// y = act(x + bias[c]) written into channels [off, off+c) of one or two (b, C_total, hw) buffers: the conv
// library's separate bias-add and ReLU passes and the torch.cat / copy_ that follow them, in one pass.
template <int VEC>
__global__ __launch_bounds__(256) void k_bias_act(const float* x, const float* __restrict__ bias, int c, int hw, int relu,
                                                  float* o1, int o1_ch, int o1_off, float* o2, int o2_ch, int o2_off) {
    const int bz = blockIdx.y;
    const size_t per = (size_t)c * hw;
    for (size_t e = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC; e < per; e += (size_t)gridDim.x * blockDim.x * VEC) {
        const float bv = bias ? bias[(int)(e / hw)] : 0.0f;
        const float* xp = x + (size_t)bz * per + e;
        float v[VEC];
        if (VEC == 4) *(float4*)v = *(const float4*)xp; else v[0] = xp[0];
#pragma unroll
        for (int k = 0; k < VEC; ++k) { v[k] += bv; if (relu) v[k] = v[k] < 0.0f ? 0.0f : v[k]; }   // (NaN stays NaN, like torch.relu) 
        float* p1 = o1 + ((size_t)bz * o1_ch + o1_off) * hw + e;
        if (VEC == 4) *(float4*)p1 = *(float4*)v; else p1[0] = v[0];
        if (o2) {
            float* p2 = o2 + ((size_t)bz * o2_ch + o2_off) * hw + e;
            if (VEC == 4) *(float4*)p2 = *(float4*)v; else p2[0] = v[0];
        }
    }
}

// Channel-slice copy: c planes per batch item from a slice of one NCHW buffer to a slice of another (the copies torch's
// ``out[:, a:b] = x[:, c:d]`` / ``.clone()`` would launch around the update block: initial hidden state, returned hidden state,
// coords1 = coords0).
template <int VEC>
__global__ __launch_bounds__(256) void k_copy_planes(const float* __restrict__ src, long long sbs, float* __restrict__ dst, long long dbs, size_t per) {
    const int bz = blockIdx.y;
    const float* sp = src + (size_t)bz * sbs;
    float* dp = dst + (size_t)bz * dbs;
    for (size_t e = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC; e < per; e += (size_t)gridDim.x * blockDim.x * VEC) {
        if (VEC == 4) *(float4*)(dp + e) = *(const float4*)(sp + e); else dp[e] = sp[e];
    }
}

// ---- encoder epilogues (core/RAFT/core/extractor.py ResidualBlock / BasicEncoder): the normalisation, ReLU and
// residual add that follow every encoder convolution, fused so each activation plane crosses HBM once in and once out
// (torch runs statistics, normalise, bias add, ReLU and the residual add as 4-5 separate passes).
//   y = norm(x + bias[c]);  if (relu) y = max(y, 0);  if (residual) y = max(residual + y, 0)
// Instance norm (fnet): one workgroup per (b, c) plane, mean and biased variance by two passes over the plane (the
// re-reads hit L2: a plane is <= 328 KB), eps inside the square root as torch.nn.InstanceNorm2d does.
__global__ __launch_bounds__(512) void k_instnorm_act(const float* __restrict__ x, const float* __restrict__ bias, int c, int hw,
                                                      float eps, int relu, const float* __restrict__ residual, float* __restrict__ out) {
    const int plane = blockIdx.x;                      // b * c + ch
    const float bv = bias ? bias[plane % c] : 0.0f;
    const float* xp = x + (size_t)plane * hw;
    const float* rp = residual ? residual + (size_t)plane * hw : nullptr;
    float* op = out + (size_t)plane * hw;
    __shared__ float red[8];
    __shared__ float bc;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const bool v4 = (hw & 3) == 0;
    const int n4 = hw >> 2;
    auto block_sum = [&](float v) -> float {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wv] = v;
        __syncthreads();
        if (threadIdx.x == 0) { float t = 0.0f; for (int i = 0; i < nw; ++i) t += red[i]; bc = t; }
        __syncthreads();
        return bc;
    };
    float acc = 0.0f;
    if (v4) for (int i = threadIdx.x; i < n4; i += blockDim.x) { float4 v = ((const float4*)xp)[i]; acc += (v.x + bv) + (v.y + bv) + (v.z + bv) + (v.w + bv); }
    else for (int i = threadIdx.x; i < hw; i += blockDim.x) acc += xp[i] + bv;
    const float mean = block_sum(acc) / (float)hw;
    acc = 0.0f;
    if (v4) for (int i = threadIdx.x; i < n4; i += blockDim.x) {
        float4 v = ((const float4*)xp)[i];
        float a = v.x + bv - mean, b = v.y + bv - mean, cc = v.z + bv - mean, d = v.w + bv - mean;
        acc += a * a + b * b + cc * cc + d * d;
    } else for (int i = threadIdx.x; i < hw; i += blockDim.x) { float a = xp[i] + bv - mean; acc += a * a; }
    const float invstd = 1.0f / sqrtf(block_sum(acc) / (float)hw + eps);
    auto fin = [&](float v, float r) -> float {
        float y = (v + bv - mean) * invstd;
        if (relu) y = y < 0.0f ? 0.0f : y;
        if (rp) { y = r + y; y = y < 0.0f ? 0.0f : y; }
        return y;
    };
    if (v4) for (int i = threadIdx.x; i < n4; i += blockDim.x) {
        float4 v = ((const float4*)xp)[i];
        float4 r = rp ? ((const float4*)rp)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        ((float4*)op)[i] = make_float4(fin(v.x, r.x), fin(v.y, r.y), fin(v.z, r.z), fin(v.w, r.w));
    } else for (int i = threadIdx.x; i < hw; i += blockDim.x) op[i] = fin(xp[i], rp ? rp[i] : 0.0f);
}

// Frozen batch norm (cnet, eval mode) folded to a per-channel affine map y = x * scale[c] + shift[c].
template <int VEC>
__global__ __launch_bounds__(256) void k_affine_act(const float* __restrict__ x, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, int c, int hw, int relu,
                                                    const float* __restrict__ residual, float* __restrict__ out) {
    const int bz = blockIdx.y;
    const size_t per = (size_t)c * hw;
    for (size_t e = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC; e < per; e += (size_t)gridDim.x * blockDim.x * VEC) {
        const int ch = (int)(e / hw);
        const float sc = scale[ch], sh = shift[ch];
        const size_t o = (size_t)bz * per + e;
        float v[VEC], r[VEC];
        if (VEC == 4) { *(float4*)v = *(const float4*)(x + o); if (residual) *(float4*)r = *(const float4*)(residual + o); }
        else { v[0] = x[o]; if (residual) r[0] = residual[o]; }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float y = v[k] * sc + sh;
            if (relu) y = y < 0.0f ? 0.0f : y;
            if (residual) { y = r[k] + y; y = y < 0.0f ? 0.0f : y; }
            v[k] = y;
        }
        if (VEC == 4) *(float4*)(out + o) = *(float4*)v; else out[o] = v[0];
    }
}

// FlowHead.conv2 (core/RAFT/core/update.py): 3x3 convolution C -> 2 channels, padding 1, + bias, optionally added to the
// running coordinates (coords1 = coords1 + delta_flow, core/RAFT/core/raft.py).  With two output channels this is a
// bandwidth problem (read C planes once), not a GEMM: the conv library runs it at ~7 TFLOP/s.  One thread per output
// pixel, 64 consecutive x per wave: every tap is a coalesced 256-B row read and the 9 taps of a channel hit the same
// three rows (L1), weights are wave-uniform scalar loads; the channel loop is split over the 4 waves of a workgroup
// (each wave a quarter of the channels, combined through LDS) so the chip gets 4x more waves than pixels/64.
// Optional extra destinations of the flow head (rpe_conv3x3_to2_flow): with add = coords1 the result is the new coords1, and
// flow = coords1 - coords0 (coords0 = the integer pixel grid, core/RAFT/core/raft.py) is what the next iteration's motion encoder
// reads and what sits in the last two channels of the GRU's input buffers: written here instead of by separate subtract / copy
// launches.  p[i] = first of the two planes of destination i in batch item 0 (NULL = unused), bs[i] = its batch stride in floats.
struct FlowDst { float* p[3]; long long bs[3]; };

#define TO1_WAVES 8                            // channel slices per workgroup (small launches are chains of load latencies: the
                                              // more slices, the shorter each chain).  The four-pixel kernel below uses the SAME
                                              // slices, tap order and explicit fused multiply-adds: which of the two a launch takes
                                              // (by its size) never shows in the result
__global__ __launch_bounds__(64 * TO1_WAVES) void k_conv3x3_to2(const float* __restrict__ x, const float* __restrict__ wgt,
                                                     const float* __restrict__ bias, int C, int h, int w,
                                                     const float* __restrict__ add, float* __restrict__ out, FlowDst F) {
    __shared__ float part[TO1_WAVES][2][64];
    const int bz = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: weights by scalar loads
    const size_t hw = (size_t)h * w;
    const size_t p = (size_t)blockIdx.x * 64 + lane;
    const bool inside = p < hw;
    const int py = inside ? (int)(p / w) : 0, px = inside ? (int)(p - (size_t)py * w) : 0;
    const float* xb = x + (size_t)bz * C * hw;
    // neighbour offsets (clamped into the map) and validity flags are the same for every channel: no conditional loads; an out-of-map
    // tap is SELECTED to zero after the load (a product with 0 would turn an Inf / NaN centre pixel into NaN where torch's zero padding
    // contributes nothing)
    int off[9]; bool msk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = py + k / 3 - 1, xx = px + k % 3 - 1;
        const bool ok = inside && yy >= 0 && yy < h && xx >= 0 && xx < w;
        off[k] = ok ? yy * w + xx : py * w + px;
        msk[k] = ok;
    }
    const int cq = (C + TO1_WAVES - 1) / TO1_WAVES, c_lo = wv * cq, c_hi = min(C, c_lo + cq);
    float a0 = 0.0f, a1 = 0.0f;
    constexpr int UN = 4;                                            // channels whose 9 loads each are issued before any is used
    int c = c_lo;
    for (; c + UN <= c_hi; c += UN) {
        float v[UN][9];
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int k = 0; k < 9; ++k) v[u][k] = xb[(size_t)(c + u) * hw + off[k]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const float* w0 = wgt + (size_t)(c + u) * 9;             // (2, C, 3, 3)
            const float* w1 = wgt + (size_t)(C + c + u) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float t = msk[k] ? v[u][k] : 0.0f;
                a0 = fmaf(t, w0[k], a0);                             // (explicit: the four-pixel kernel must round identically)
                a1 = fmaf(t, w1[k], a1);
            }
        }
    }
    for (; c < c_hi; ++c) {
        const float* xc = xb + (size_t)c * hw;
        const float* w0 = wgt + (size_t)c * 9;
        const float* w1 = wgt + (size_t)(C + c) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float t = msk[k] ? xc[off[k]] : 0.0f;
            a0 = fmaf(t, w0[k], a0);
            a1 = fmaf(t, w1[k], a1);
        }
    }
    part[wv][0][lane] = a0; part[wv][1][lane] = a1;
    __syncthreads();
    if (wv == 0 && inside) {
        a0 = part[0][0][lane]; a1 = part[0][1][lane];
#pragma unroll
        for (int k = 1; k < TO1_WAVES; ++k) { a0 += part[k][0][lane]; a1 += part[k][1][lane]; }
        const size_t o = (size_t)bz * 2 * hw + p;
        a0 += bias ? bias[0] : 0.0f; a1 += bias ? bias[1] : 0.0f;
        if (add) { a0 += add[o]; a1 += add[o + hw]; }
        out[o] = a0; out[o + hw] = a1;
        const float f0 = a0 - (float)px, f1 = a1 - (float)py;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (F.p[i]) { float* d = F.p[i] + (size_t)bz * F.bs[i] + p; d[0] = f0; d[hw] = f1; }
    }
}

// The same for maps whose width is a multiple of 4: one thread per FOUR consecutive pixels of a row.  Per channel and
// input row it loads one aligned float4 plus its left and right neighbour (9 loads for 4 outputs instead of 36: the
// one-pixel kernel is bound by L1 request rate, 143 us for 168 MB), always from a clamped in-range address, and zeroes
// what lies outside the map by SELECTS after the loads -- no conditional loads (those compile to a branch around every load), and no
// products with zero (0 * Inf = NaN, where torch's zero padding contributes nothing).
#define TO2_WAVES TO1_WAVES                   // the same channel slices as the one-pixel kernel (bit-identical partial sums)
__global__ __launch_bounds__(64 * TO2_WAVES) void k_conv3x3_to2_x4(const float* __restrict__ x, const float* __restrict__ wgt,
                                                        const float* __restrict__ bias, int C, int h, int w,
                                                        const float* __restrict__ add, float* __restrict__ out, FlowDst F) {
    __shared__ float part[TO2_WAVES][8][64];
    const int bz = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: weights by scalar loads
    const int hw = h * w, w4 = w >> 2;
    const int g = blockIdx.x * 64 + lane;                            // group of 4 pixels
    const bool inside = g < (hw >> 2);
    const int py = inside ? g / w4 : 0, x0 = inside ? (g - py * w4) * 4 : 0;
    const float* xb = x + (size_t)bz * C * hw;
    int rowoff[3]; bool rowk[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int yy = py + d - 1;
        const bool ok = inside && yy >= 0 && yy < h;
        rowoff[d] = (ok ? yy : py) * w + x0;
        rowk[d] = ok;
    }
    const bool hasl = x0 > 0, hasr = x0 + 4 < w;
    // The pixel left of a lane's four is the last pixel of lane - 1's four (same image row whenever hasl), the one to the right the
    // first of lane + 1's: they come by DPP wave shifts instead of two more loads per row -- only lanes 0 and 63 load theirs (one
    // two-lane load per row).  9 -> 6 load instructions per channel, a third less L1 traffic.
    const bool edge = lane == 0 || lane == 63;
    const int eoff = lane == 0 ? (hasl ? -1 : 0) : (hasr ? 4 : 3);
    const int cq = (C + TO2_WAVES - 1) / TO2_WAVES, c_lo = wv * cq, c_hi = min(C, c_lo + cq);
    float a[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    auto accumulate = [&](int c, const float4 (&m)[3], const float (&ed)[3]) {
        const float* w0 = wgt + (size_t)c * 9;                       // (2, C, 3, 3): wave-uniform -> scalar loads
        const float* w1 = wgt + (size_t)(C + c) * 9;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const bool rk = rowk[d];
            const int ei = __builtin_bit_cast(int, ed[d]);
            const float lft = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ei, __builtin_bit_cast(int, m[d].w), 0x138, 0xF, 0xF, false));   // wave_shr:1
            const float rgt = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ei, __builtin_bit_cast(int, m[d].x), 0x130, 0xF, 0xF, false));   // wave_shl:1
            const float v[6] = {rk && hasl ? lft : 0.0f, rk ? m[d].x : 0.0f, rk ? m[d].y : 0.0f, rk ? m[d].z : 0.0f, rk ? m[d].w : 0.0f, rk && hasr ? rgt : 0.0f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {                     // tap order and rounding of the one-pixel kernel
                    a[0][e] = fmaf(v[e + dx], w0[3 * d + dx], a[0][e]);
                    a[1][e] = fmaf(v[e + dx], w1[3 * d + dx], a[1][e]);
                }
        }
    };
#ifndef TO2_UN
#define TO2_UN 2                                /* 2: 78 registers, three 8-wave workgroups per CU -- all 640 workgroups of a 32-pair launch are resident at once
                                                   (4: 113 registers, two per CU and a second, quarter-full round: 60.5 vs 54 us; 6 and 8: 83-90 us) */
#endif
    constexpr int UN = TO2_UN;                                       // channels whose loads are issued before any is used:
    int c = c_lo;                                                    // the loop is latency-bound
    for (; c + UN <= c_hi; c += UN) {
        float4 m[UN][3]; float ed[UN][3];
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float* r = xb + (size_t)(c + u) * hw + rowoff[d];
                m[u][d] = *(const float4*)r;
                ed[u][d] = 0.0f;
                if (edge) ed[u][d] = r[eoff];
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UN; ++u) accumulate(c + u, m[u], ed[u]);
    }
    for (; c < c_hi; ++c) {
        float4 m[3]; float ed[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float* r = xb + (size_t)c * hw + rowoff[d];
            m[d] = *(const float4*)r;
            ed[d] = 0.0f;
            if (edge) ed[d] = r[eoff];
        }
        accumulate(c, m, ed);
    }
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int e = 0; e < 4; ++e) part[wv][o * 4 + e][lane] = a[o][e];
    __syncthreads();
    if (wv == 0 && inside) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            float r4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
            {
                float t = part[0][o * 4 + e][lane];
#pragma unroll
                for (int k = 1; k < TO2_WAVES; ++k) t += part[k][o * 4 + e][lane];
                r4[e] = t + (bias ? bias[o] : 0.0f);
            }
            const size_t idx = ((size_t)bz * 2 + o) * hw + (size_t)py * w + x0;
            if (add) { const float4 ad = *(const float4*)(add + idx); r4[0] += ad.x; r4[1] += ad.y; r4[2] += ad.z; r4[3] += ad.w; }
            *(float4*)(out + idx) = make_float4(r4[0], r4[1], r4[2], r4[3]);
            const float g0 = o == 0 ? (float)x0 : (float)py, gs = o == 0 ? 1.0f : 0.0f;      // coords0: x along the row, y constant
            const float4 fl = make_float4(r4[0] - g0, r4[1] - (g0 + gs), r4[2] - (g0 + 2.0f * gs), r4[3] - (g0 + 3.0f * gs));
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (F.p[i]) *(float4*)(F.p[i] + (size_t)bz * F.bs[i] + (size_t)o * hw + (size_t)py * w + x0) = fl;
        }
    }
}

// One thread per 1/8-resolution cell and sub-pixel row; loops over the row's 8 sub-pixels.  Mask channel = k*64 + i*8 + j.
__global__ __launch_bounds__(256) void k_upsample_convex(const float* __restrict__ flow, const float* __restrict__ mask, int h8,
                                                         int w8, float* __restrict__ out) {
    const int bz = blockIdx.y;
    const int nq = h8 * w8;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int y = q / w8, x = q - y * w8;
    float fx[9], fy[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {                              // F.unfold(8*flow, 3, padding=1): zero padded
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        const bool ok = yy >= 0 && yy < h8 && xx >= 0 && xx < w8;
        fx[k] = ok ? 8.0f * flow[((size_t)bz * 2 + 0) * nq + (size_t)yy * w8 + xx] : 0.0f;
        fy[k] = ok ? 8.0f * flow[((size_t)bz * 2 + 1) * nq + (size_t)yy * w8 + xx] : 0.0f;
    }
    const float* mb = mask + (size_t)bz * 576 * nq + q;
    const int W = 8 * w8;
    float* ox = out + ((size_t)bz * 2 + 0) * 64 * nq;
    float* oy = out + ((size_t)bz * 2 + 1) * 64 * nq;
    // blockIdx.z = sub-pixel row i of the 8 x 8 block: eight times the threads of a thread-per-cell loop (a batch-2 launch was 40
    // workgroups walking 64 sub-pixels each: 96 us of load latency), the same arithmetic per sub-pixel
    {
        const int i = blockIdx.z;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float m[9], mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) { m[k] = mb[(size_t)(k * 64 + i * 8 + j) * nq]; mx = fmaxf(mx, m[k]); }
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) { m[k] = expf(m[k] - mx); sum += m[k]; }
            float ax = 0.0f, ay = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) { float p = m[k] / sum; ax += p * fx[k]; ay += p * fy[k]; }
            const size_t o = (size_t)(8 * y + i) * W + 8 * x + j;
            ox[o] = ax; oy[o] = ay;
        }
    }
}

static bool vec_ok(const void* p) { return ((uintptr_t)p % 16) == 0; }

extern "C" int rpe_instnorm_act(const float* x, const float* bias, int b, int c, int hw, float eps, int relu,
                                const float* residual, float* out, void* stream) {
    if (!x || !out || b <= 0 || c <= 0 || hw <= 0) return RPE_E_BADARG;
    bool v4 = vec_ok(x) && vec_ok(out) && (!residual || vec_ok(residual));
    if (!v4 && (hw & 3) == 0) return RPE_E_BADARG;     // planes of 16-B-aligned tensors with hw % 4 == 0 stay aligned
    hipLaunchKernelGGL(k_instnorm_act, dim3(b * c), dim3(hw >= 16384 ? 512 : 256), 0, (hipStream_t)stream, x, bias, c, hw, eps, relu,
                       residual, out);
    return rpe_check_launch();
}

extern "C" int rpe_affine_act(const float* x, const float* scale, const float* shift, int b, int c, int hw, int relu,
                              const float* residual, float* out, void* stream) {
    if (!x || !scale || !shift || !out || b <= 0 || c <= 0 || hw <= 0) return RPE_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    size_t per = (size_t)c * hw;
    bool v4 = hw % 4 == 0 && vec_ok(x) && vec_ok(out) && (!residual || vec_ok(residual));
    if (v4) hipLaunchKernelGGL(k_affine_act<4>, dim3(min(ceil_div(per / 4, 256), 2048), b), dim3(256), 0, s, x, scale, shift, c, hw, relu, residual, out);
    else hipLaunchKernelGGL(k_affine_act<1>, dim3(min(ceil_div(per, 256), 2048), b), dim3(256), 0, s, x, scale, shift, c, hw, relu, residual, out);
    return rpe_check_launch();
}

static int launch_to2(const float* x, const float* weight, const float* bias, int b, int c, int h, int w, const float* add, float* out,
                      const FlowDst& F, void* stream) {
    if (!x || !weight || !out || b <= 0 || c <= 0 || h <= 0 || w <= 0) return RPE_E_BADARG;
    bool fvec = true;
    for (int i = 0; i < 3; ++i) fvec = fvec && (!F.p[i] || (vec_ok(F.p[i]) && (F.bs[i] & 3) == 0));
    // (the four-pixel kernel has a quarter of the workgroups: small launches -- one frame of sequential tracking -- keep the one-pixel one)
    if ((w & 3) == 0 && vec_ok(x) && vec_ok(out) && (!add || vec_ok(add)) && fvec && (long long)ceil_div((size_t)h * w / 4, 64) * b >= 256)
        hipLaunchKernelGGL(k_conv3x3_to2_x4, dim3(ceil_div((size_t)h * w / 4, 64), b), dim3(64 * TO2_WAVES), 0, (hipStream_t)stream, x, weight, bias,
                           c, h, w, add, out, F);
    else
        hipLaunchKernelGGL(k_conv3x3_to2, dim3(ceil_div((size_t)h * w, 64), b), dim3(64 * TO1_WAVES), 0, (hipStream_t)stream, x, weight, bias,
                           c, h, w, add, out, F);
    return rpe_check_launch();
}

extern "C" int rpe_conv3x3_to2(const float* x, const float* weight, const float* bias, int b, int c, int h, int w,
                               const float* add, float* out, void* stream) {
    FlowDst F = {{nullptr, nullptr, nullptr}, {0, 0, 0}};
    return launch_to2(x, weight, bias, b, c, h, w, add, out, F, stream);
}

extern "C" int rpe_conv3x3_to2_flow(const float* x, const float* weight, const float* bias, int b, int c, int h, int w,
                                    const float* coords, float* coords_out, float* flow_out, float* dst1, long long dst1_batch_stride,
                                    float* dst2, long long dst2_batch_stride, void* stream) {
    if (!coords) return RPE_E_BADARG;
    FlowDst F = {{flow_out, dst1, dst2}, {2LL * h * w, dst1_batch_stride, dst2_batch_stride}};
    return launch_to2(x, weight, bias, b, c, h, w, coords, coords_out, F, stream);
}

extern "C" int rpe_bias_act(const float* x, const float* bias, int b, int c, int hw, int relu, float* out1, int out1_channels,
                            int out1_offset, float* out2, int out2_channels, int out2_offset, void* stream) {
    if (!x || !out1 || b <= 0 || c <= 0 || hw <= 0 || out1_offset < 0 || out1_offset + c > out1_channels) return RPE_E_BADARG;
    if (out2 && (out2_offset < 0 || out2_offset + c > out2_channels)) return RPE_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    size_t per = (size_t)c * hw;
    bool v4 = hw % 4 == 0 && vec_ok(x) && vec_ok(out1) && (!out2 || vec_ok(out2));
    if (v4) hipLaunchKernelGGL(k_bias_act<4>, dim3(min(ceil_div(per / 4, 256), 2048), b), dim3(256), 0, s, x, bias, c, hw, relu, out1, out1_channels, out1_offset, out2, out2_channels, out2_offset);
    else hipLaunchKernelGGL(k_bias_act<1>, dim3(min(ceil_div(per, 256), 2048), b), dim3(256), 0, s, x, bias, c, hw, relu, out1, out1_channels, out1_offset, out2, out2_channels, out2_offset);
    return rpe_check_launch();
}

extern "C" int rpe_copy_planes(const float* src, long long src_batch_stride, float* dst, long long dst_batch_stride, int b, int c, int hw,
                               void* stream) {
    if (!src || !dst || b <= 0 || c <= 0 || hw <= 0) return RPE_E_BADARG;
    const size_t per = (size_t)c * hw;
    const bool v4 = per % 4 == 0 && vec_ok(src) && vec_ok(dst) && (src_batch_stride & 3) == 0 && (dst_batch_stride & 3) == 0;
    if (v4) hipLaunchKernelGGL(k_copy_planes<4>, dim3(min(ceil_div(per / 4, 256), 2048), b), dim3(256), 0, (hipStream_t)stream, src, src_batch_stride, dst, dst_batch_stride, per);
    else hipLaunchKernelGGL(k_copy_planes<1>, dim3(min(ceil_div(per, 256), 2048), b), dim3(256), 0, (hipStream_t)stream, src, src_batch_stride, dst, dst_batch_stride, per);
    return rpe_check_launch();
}

extern "C" int rpe_gru_gates_zr(const float* zr_pre, const float* zr_bias, const float* zr_add, const float* h, int h_channels,
                                int b, int c, int hw, float* z_out, float* rh_out, int rh_channels, void* stream) {
    if (!zr_pre || !h || !z_out || !rh_out || b <= 0 || c <= 0 || hw <= 0 || h_channels < c || rh_channels < c) return RPE_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    size_t per = (size_t)c * hw;
    bool v4 = hw % 4 == 0 && vec_ok(zr_pre) && vec_ok(h) && vec_ok(z_out) && vec_ok(rh_out) && (!zr_add || vec_ok(zr_add));
    if (v4) hipLaunchKernelGGL(k_gates_zr<4>, dim3(min(ceil_div(per / 4, 256), 2048), b), dim3(256), 0, s, zr_pre, zr_bias, zr_add, h, c, hw, h_channels, z_out, rh_out, rh_channels);
    else hipLaunchKernelGGL(k_gates_zr<1>, dim3(min(ceil_div(per, 256), 2048), b), dim3(256), 0, s, zr_pre, zr_bias, zr_add, h, c, hw, h_channels, z_out, rh_out, rh_channels);
    return rpe_check_launch();
}

extern "C" int rpe_gru_gates_h(const float* z, const float* q_pre, const float* q_bias, const float* q_add, const float* h,
                               int h_channels, int b, int c, int hw, float* h_out, int hout_channels, void* stream) {
    if (!z || !q_pre || !h || !h_out || b <= 0 || c <= 0 || hw <= 0 || h_channels < c || hout_channels < c) return RPE_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    size_t per = (size_t)c * hw;
    bool v4 = hw % 4 == 0 && vec_ok(z) && vec_ok(q_pre) && vec_ok(h) && vec_ok(h_out) && (!q_add || vec_ok(q_add));
    if (v4) hipLaunchKernelGGL(k_gates_h<4>, dim3(min(ceil_div(per / 4, 256), 2048), b), dim3(256), 0, s, z, q_pre, q_bias, q_add, h, c, hw, h_channels, h_out, hout_channels);
    else hipLaunchKernelGGL(k_gates_h<1>, dim3(min(ceil_div(per, 256), 2048), b), dim3(256), 0, s, z, q_pre, q_bias, q_add, h, c, hw, h_channels, h_out, hout_channels);
    return rpe_check_launch();
}

extern "C" int rpe_upsample_convex(const float* flow, const float* mask, int b, int h8, int w8, float* out, void* stream) {
    if (!flow || !mask || !out || b <= 0 || h8 <= 0 || w8 <= 0) return RPE_E_BADARG;
    hipLaunchKernelGGL(k_upsample_convex, dim3(ceil_div((size_t)h8 * w8, 256), b, 8), dim3(256), 0, (hipStream_t)stream, flow, mask, h8, w8, out);
    return rpe_check_launch();
}
