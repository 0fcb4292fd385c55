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
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u); }
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
    if (!x || !gamma || !beta || !y) return DTLR_EINVAL;
    if (rows <= 0 || C <= 0) return DTLR_EINVAL;
    if (C % 256 != 0) return DTLR_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
    case DTLR_F32: return launch_ln<float>(x, residual, gamma, beta, y, rows, C, eps, st);
    case DTLR_BF16: return launch_ln<uint16_t>(x, residual, gamma, beta, y, rows, C, eps, st);
    default: return DTLR_EDTYPE;
    }
}
