// Row normalisations for gfx950: wavefront reductions, one 64-lane wave per row, rows of C = 64*VPL.
//
//   dtlr_layernorm : y = LayerNorm(x [+ residual]) * gamma + beta    (post-norm blocks of the
//                    encoder/decoder: deformable_transformer.py:804-823, 876-959; eps 1e-5)
// HBM-bound elementwise/reduction work: each element is read once and
// written once; the residual add is fused so the sum never round-trips through HBM.  Statistics
// are fp32 two-pass (mean, then centred variance) like ATen's LayerNorm.
#include "dtlr_common.h"

namespace dtlr {

template <typename T> struct IO;
template <> struct IO<float> {
    static __device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct IO<uint16_t> {   // bf16
    static __device__ __forceinline__ void load4(const uint16_t* p, float (&v)[4]) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = h16_lo(t.x); v[1] = h16_hi(t.x);
        v[2] = h16_lo(t.y); v[3] = h16_hi(t.y); }
    static __device__ __forceinline__ void store4(uint16_t* p, const float (&v)[4]) {
        *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])); }
};

// C = 256 * CH: each lane owns CH groups of 4 contiguous channels; group g of lane l covers
// channels [g*256 + 4l, g*256 + 4l + 4) so every wave-level load/store is one contiguous segment.
template <typename T, int CH>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        T* __restrict__ y, long rows, float eps)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int C = 256 * CH;
    const T* xr = x + row * C;
    float v[CH][4];
#pragma unroll
    for (int g = 0; g < CH; ++g) IO<T>::load4(xr + g * 256 + 4 * lane, v[g]);
    if (res) {
        const T* rr = res + row * C;
#pragma unroll
        for (int g = 0; g < CH; ++g) {
            float r[4]; IO<T>::load4(rr + g * 256 + 4 * lane, r);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[g][i] += r[i];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < CH; ++g) s += (v[g][0] + v[g][1]) + (v[g][2] + v[g][3]);
    const float mean = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int g = 0; g < CH; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = v[g][i] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / C) + eps);
    T* yr = y + row * C;
#pragma unroll
    for (int g = 0; g < CH; ++g) {
        const float4 ga = *reinterpret_cast<const float4*>(gamma + g * 256 + 4 * lane);
        const float4 be = *reinterpret_cast<const float4*>(beta + g * 256 + 4 * lane);
        float o[4] = {(v[g][0] - mean) * rstd * ga.x + be.x, (v[g][1] - mean) * rstd * ga.y + be.y,
                      (v[g][2] - mean) * rstd * ga.z + be.z, (v[g][3] - mean) * rstd * ga.w + be.w};
        IO<T>::store4(yr + g * 256 + 4 * lane, o);
    }
}

// any C with C % 4 == 0 and C <= 3072 (Swin stages: 96 / 192 / 384 / 768 / 1536 ...): lane l owns elements 4l + 256 g, g < ceil(C/256)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_any_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            T* __restrict__ y, long rows, int C, float eps)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + row * C;
    float v[12][4];
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 12; ++g) {
        const int e = g * 256 + 4 * lane;
        v[g][0] = v[g][1] = v[g][2] = v[g][3] = 0.f;
        if (e < C) {
            IO<T>::load4(xr + e, v[g]);
            if (res) {
                float r[4]; IO<T>::load4(res + row * C + e, r);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[g][i] += r[i];
            }
            s += (v[g][0] + v[g][1]) + (v[g][2] + v[g][3]);
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int g = 0; g < 12; ++g)
        if (g * 256 + 4 * lane < C) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float d = v[g][i] - mean; q += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    T* yr = y + row * C;
#pragma unroll
    for (int g = 0; g < 12; ++g) {
        const int e = g * 256 + 4 * lane;
        if (e < C) {
            const float4 ga = *reinterpret_cast<const float4*>(gamma + e), be = *reinterpret_cast<const float4*>(beta + e);
            float o[4] = {(v[g][0] - mean) * rstd * ga.x + be.x, (v[g][1] - mean) * rstd * ga.y + be.y,
                          (v[g][2] - mean) * rstd * ga.z + be.z, (v[g][3] - mean) * rstd * ga.w + be.w};
            IO<T>::store4(yr + e, o);
        }
    }
}

template <typename T>
static int launch_ln(const void* x, const void* res, const float* gamma, const float* beta, void* y,
                     long rows, int C, float eps, hipStream_t st) {
    const int block = 256, rpb = block / 64;
    const long grid = (rows + rpb - 1) / rpb;
    if (grid > 0x7fffffffL) return DTLR_ESHAPE;
#define LN_CASE(CH) case CH: hipLaunchKernelGGL((layernorm_kernel<T, CH>), dim3((unsigned)grid), dim3(block), 0, st, \
        (const T*)x, (const T*)res, gamma, beta, (T*)y, rows, eps); break;
    switch (C / 256) { LN_CASE(1) LN_CASE(2) LN_CASE(4) LN_CASE(8) default: return DTLR_ESHAPE; }
#undef LN_CASE
    return check_launch();
}

}  // namespace dtlr

using namespace dtlr;

extern "C" int dtlr_layernorm(const void* x, const void* residual, const float* gamma, const float* beta,
                              void* y, long rows, int C, float eps, int dtype, void* stream)
{
    clear_stale_error();
    if (!x || !gamma || !beta || !y) return DTLR_EINVAL;
    if (rows <= 0 || C <= 0) return DTLR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int ch256 = C / 256;
    if (C % 256 != 0 || !(ch256 == 1 || ch256 == 2 || ch256 == 4 || ch256 == 8)) {     // generic row kernel (Swin widths: 96 .. 1536, 768, 3072)
        if ((C & 3) || C > 3072) return DTLR_ESHAPE;
        const long grid = (rows + 3) / 4;
        if (grid > 0x7fffffffL) return DTLR_ESHAPE;
        if (dtype == DTLR_F32)
            hipLaunchKernelGGL(layernorm_any_kernel<float>, dim3((unsigned)grid), dim3(256), 0, st, (const float*)x, (const float*)residual, gamma, beta, (float*)y, rows, C, eps);
        else if (dtype == DTLR_H16)
            hipLaunchKernelGGL(layernorm_any_kernel<uint16_t>, dim3((unsigned)grid), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)residual, gamma, beta, (uint16_t*)y, rows, C, eps);
        else return DTLR_EDTYPE;
        return check_launch();
    }
    switch (dtype) {
    case DTLR_F32: return launch_ln<float>(x, residual, gamma, beta, y, rows, C, eps, st);
    case DTLR_H16: return launch_ln<uint16_t>(x, residual, gamma, beta, y, rows, C, eps, st);
    default: return DTLR_EDTYPE;
    }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm(32, 256) over the tokens of ONE feature level: x [B, T, 256], statistics per (sample, group)
// over T positions x 8 channels (models/dino/dino.py:121-134, eps 1e-5, affine).
//   pass 1 (gn_partial): grid (slabs, B); a wave owns rows of its slab, lane l owns channels 4l..4l+3, so
//           lanes 2g and 2g+1 hold group g; per-lane fp32 sum / sum of squares -> one (sum, sumsq) per
//           (block, group) in the workspace.
//   pass 2 (gn_apply):   each block first reduces the slab partials of its sample in fp64 (32 groups),
//           then normalises rows (one wave per row, 16-byte loads).
// ---------------------------------------------------------------------------------------------
namespace dtlr {

template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, float2* __restrict__ part, int T_tokens, int rows_per_slab)
{
    __shared__ float2 red[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slab = blockIdx.x, b = blockIdx.y;
    const int r0 = slab * rows_per_slab, r1 = min(r0 + rows_per_slab, T_tokens);
    float s = 0.f, q = 0.f;
    for (int r = r0 + wave; r < r1; r += 4) {
        float v[4];
        IO<T>::load4(x + ((long)b * T_tokens + r) * 256 + 4 * lane, v);
#pragma unroll
        for (int i = 0; i < 4; ++i) { s += v[i]; q += v[i] * v[i]; }
    }
    s += __shfl_xor(s, 1, 64);
    q += __shfl_xor(q, 1, 64);
    if ((lane & 1) == 0) red[wave][lane >> 1] = make_float2(s, q);
    __syncthreads();
    if (threadIdx.x < 32) {
        float2 a = red[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < 4; ++w) { a.x += red[w][threadIdx.x].x; a.y += red[w][threadIdx.x].y; }
        part[((long)b * gridDim.x + slab) * 32 + threadIdx.x] = a;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const float2* __restrict__ part,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ y, int T_tokens, int nslab, int rows_per_block, float eps,
                                                       long y_bstride)
{
    __shared__ float2 stat[32];                     // (mean, rstd) per group
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    if (threadIdx.x < 32) {
        double s = 0.0, q = 0.0;
        for (int k = 0; k < nslab; ++k) { const float2 p = part[((long)b * nslab + k) * 32 + threadIdx.x]; s += (double)p.x; q += (double)p.y; }
        const double n = (double)T_tokens * 8.0;
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        stat[threadIdx.x] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
    }
    __syncthreads();
    const float2 st = stat[lane >> 1];
    const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * lane);
    const float4 be = *reinterpret_cast<const float4*>(beta + 4 * lane);
    const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, T_tokens);
    // four rows per pass, loads before stores (a load issued after a store waits for that store's round trip: vmcnt is shared)
    for (int r = r0 + wave; r < r1; r += 16) {
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = min(r + 4 * u, r1 - 1);
            IO<T>::load4(x + ((long)b * T_tokens + rr) * 256 + 4 * lane, v[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r + 4 * u >= r1) break;
            float o[4] = {(v[u][0] - st.x) * st.y * ga.x + be.x, (v[u][1] - st.x) * st.y * ga.y + be.y,
                          (v[u][2] - st.x) * st.y * ga.z + be.z, (v[u][3] - st.x) * st.y * ga.w + be.w};
            IO<T>::store4(y + (long)b * y_bstride + (long)(r + 4 * u) * 256 + 4 * lane, o);
        }
    }
}

// 3x3 stride-2 pad-1 max pooling on NHWC (torchvision resnet50.maxpool as run by backbone.py:97-106);
// a thread owns VEC channels of one output pixel; padding behaves as -inf.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ bias,
                                                           int relu, int H, int W, int C, int Ho, int Wo, long total)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= total) return;
    const int cv = C / VEC;
    const int c0 = (int)(tid % cv) * VEC;
    long p = tid / cv;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const long b = p / Ho;
    float m[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) m[i] = -INFINITY;
    // branch-free: the nine loads are always issued (addresses clamped into the map) and an out-of-map tap is replaced by -inf with a
    // select -- with `if (outside) continue` hipcc put every load in its own exec-masked block behind a wait (nine serialised round trips)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hi = 2 * ho - 1 + kh;
        const bool hok = hi >= 0 && hi < H;
        const int hc = min(max(hi, 0), H - 1);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int wi = 2 * wo - 1 + kw;
            const bool ok = hok && wi >= 0 && wi < W;
            float v[VEC];
            const T* src = x + ((b * H + hc) * W + min(max(wi, 0), W - 1)) * C + c0;
            if constexpr (VEC == 8 && sizeof(T) == 2) {               // one 16-byte load (8-byte accesses run at ~0.6x the 16-byte rate)
                const uint4 t = *reinterpret_cast<const uint4*>(src);
                const uint32_t w4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[2 * i] = h16_lo(w4[i]); v[2 * i + 1] = h16_hi(w4[i]); }
            } else if (VEC == 4) IO<T>::load4(src, *reinterpret_cast<float (*)[4]>(v));
            else { IO<T>::load4(src, *reinterpret_cast<float (*)[4]>(v)); IO<T>::load4(src + 4, *reinterpret_cast<float (*)[4]>(v + 4)); }
#pragma unroll
            for (int i = 0; i < VEC; ++i) m[i] = fmaxf(m[i], ok ? v[i] : -INFINITY);
        }
    }
    // optional per-channel bias + ReLU applied AFTER the max: max_i(x_i + b) == max_i(x_i) + b exactly (rounding is
    // monotone) and ReLU commutes with max, so conv -> +bias -> ReLU -> maxpool (backbone.py:97-106 with the folded
    // FrozenBN shift as bias) costs one pass over the full-resolution map instead of three
    if (bias) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) m[i] += bias[c0 + i];
    }
    if (relu) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) m[i] = fmaxf(m[i], 0.f);
    }
    T* dst = y + ((b * Ho + ho) * Wo + wo) * C + c0;
    if constexpr (VEC == 8 && sizeof(T) == 2) {
        *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]), pack_bf16x2(m[4], m[5]), pack_bf16x2(m[6], m[7]));
    } else {
        IO<T>::store4(dst, *reinterpret_cast<float (*)[4]>(m));
        if (VEC == 8) IO<T>::store4(dst + 4, *reinterpret_cast<float (*)[4]>(m + 4));
    }
}

}  // namespace dtlr

extern "C" long dtlr_groupnorm_workspace_bytes(int B, int T_tokens)
{
    const int rows_per_slab = 64;
    const long nslab = (T_tokens + rows_per_slab - 1) / rows_per_slab;
    return (long)B * nslab * 32 * (long)sizeof(float2);
}

extern "C" int dtlr_groupnorm_tokens(const void* x, const float* gamma, const float* beta, void* y, void* workspace,
                                     int B, int T_tokens, int C, int groups, float eps, int dtype, void* stream)
{
    return dtlr_groupnorm_tokens_strided(x, gamma, beta, y, 0, workspace, B, T_tokens, C, groups, eps, dtype, stream);
}

extern "C" int dtlr_groupnorm_tokens_strided(const void* x, const float* gamma, const float* beta, void* y, long y_batch_stride,
                                             void* workspace, int B, int T_tokens, int C, int groups, float eps, int dtype, void* stream)
{
    clear_stale_error();
    const long ybs = y_batch_stride > 0 ? y_batch_stride : (long)T_tokens * 256;
    if (ybs < (long)T_tokens * 256) return DTLR_EINVAL;
    if (!x || !gamma || !beta || !y || !workspace) return DTLR_EINVAL;
    if (B <= 0 || T_tokens <= 0) return DTLR_EINVAL;
    if (C != 256 || groups != 32) return DTLR_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int rows_per_slab = 64;
    const int nslab = (T_tokens + rows_per_slab - 1) / rows_per_slab;
    const int rows_per_block = 32;
    const int nblk = (T_tokens + rows_per_block - 1) / rows_per_block;
    if (dtype == DTLR_F32) {
        hipLaunchKernelGGL((gn_partial_kernel<float>), dim3(nslab, B), dim3(256), 0, st, (const float*)x, (float2*)workspace, T_tokens, rows_per_slab);
        hipLaunchKernelGGL((gn_apply_kernel<float>), dim3(nblk, B), dim3(256), 0, st, (const float*)x, (const float2*)workspace, gamma, beta, (float*)y, T_tokens, nslab, rows_per_block, eps, ybs);
    } else if (dtype == DTLR_H16) {
        hipLaunchKernelGGL((gn_partial_kernel<uint16_t>), dim3(nslab, B), dim3(256), 0, st, (const uint16_t*)x, (float2*)workspace, T_tokens, rows_per_slab);
        hipLaunchKernelGGL((gn_apply_kernel<uint16_t>), dim3(nblk, B), dim3(256), 0, st, (const uint16_t*)x, (const float2*)workspace, gamma, beta, (uint16_t*)y, T_tokens, nslab, rows_per_block, eps, ybs);
    } else return DTLR_EDTYPE;
    return check_launch();
}

extern "C" int dtlr_maxpool3x3s2_nhwc(const void* x, void* y, const float* bias, int relu, int B, int H, int W, int C, int dtype, void* stream)
{
    clear_stale_error();
    if (!x || !y) return DTLR_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return DTLR_EINVAL;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DTLR_H16 && C % 8 == 0) {
        const long total = (long)B * Ho * Wo * (C / 8);
        hipLaunchKernelGGL((maxpool3x3s2_kernel<uint16_t, 8>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           (const uint16_t*)x, (uint16_t*)y, bias, relu, H, W, C, Ho, Wo, total);
    } else if (dtype == DTLR_F32 && C % 4 == 0) {
        const long total = (long)B * Ho * Wo * (C / 4);
        hipLaunchKernelGGL((maxpool3x3s2_kernel<float, 4>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           (const float*)x, (float*)y, bias, relu, H, W, C, Ho, Wo, total);
    } else return (dtype == DTLR_H16 || dtype == DTLR_F32) ? DTLR_ESHAPE : DTLR_EDTYPE;
    return check_launch();
}
