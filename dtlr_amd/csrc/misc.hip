// Small fused elementwise kernels of the decoder loop (each replaces a dozen tiny launches per layer).
#include "dtlr_common.h"

namespace dtlr {

// TransformerDecoder.forward per-layer query preparation (deformable_transformer.py:684-690) +
// gen_sineembed_for_position (models/dino/utils.py:141-167):
//   ref_in[q, l, :] = ref[q, :] * (vr[b,l,0], vr[b,l,1], vr[b,l,0], vr[b,l,1])
//   sine[q, :]      = [emb(y) | emb(x) | emb(w) | emb(h)] of ref_in[q, 0, :],  emb(c)[i] = sin/cos(c * 2pi / dim_t[i])
// one thread per (query, frequency pair): 64 pairs x 4 coordinates; dim_t comes from the host so it is
// bit-identical to the table torch builds (10000 ** (2*(i//2)/128) in fp32).
template <typename OT>
__global__ __launch_bounds__(256) void query_prep_kernel(const float* __restrict__ ref, const float* __restrict__ vr,
                                                         const float* __restrict__ dim_t, float* __restrict__ ref_in,
                                                         OT* __restrict__ sine, int nq, int L, long total)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= total) return;
    const int pair = (int)(tid & 63);
    const long q = tid >> 6;
    const int b = (int)(q / nq);
    const float4 r = *reinterpret_cast<const float4*>(ref + q * 4);
    if (pair < L) {                                   // lanes 0..L-1 also write the per-level reference boxes
        const float vx = vr[(b * L + pair) * 2], vy = vr[(b * L + pair) * 2 + 1];
        *reinterpret_cast<float4*>(ref_in + (q * L + pair) * 4) = make_float4(r.x * vx, r.y * vy, r.z * vx, r.w * vy);
    }
    const float vx0 = vr[(b * L) * 2], vy0 = vr[(b * L) * 2 + 1];
    const float scale = 6.283185307179586f;           // 2*pi, as `scale = 2 * math.pi` rounded to fp32 by torch
    const float c[4] = {r.y * vy0 * scale, r.x * vx0 * scale, r.z * vx0 * scale, r.w * vy0 * scale};   // order y, x, w, h
    const float d0 = dim_t[2 * pair], d1 = dim_t[2 * pair + 1];
    const float r0 = __frcp_rn(d0), r1 = __frcp_rn(d1);          // bf16 path: one reciprocal per frequency instead of eight divisions
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // bf16 engine: the hardware sine (v_sin_f32, ~1e-6 absolute on these arguments of at most a few turns) -- the result is rounded to
        // 2^-9 relative anyway; the library sinf / cosf with their software range reduction made this the slowest pure-write kernel
        // of the decoder (20 us per layer for 29 MB).  fp32 (parity) engine: the library functions, as torch computes them.
        const float s = sizeof(OT) == 2 ? __sinf(c[k] * r0) : sinf(c[k] / d0);
        const float co = sizeof(OT) == 2 ? __cosf(c[k] * r1) : cosf(c[k] / d1);
        OT* o = sine + q * 512 + k * 128 + 2 * pair;
        if (sizeof(OT) == 2) *reinterpret_cast<uint32_t*>(o) = pack_bf16x2(s, co);
        else { reinterpret_cast<float*>(o)[0] = s; reinterpret_cast<float*>(o)[1] = co; }
    }
}

// iterative box refinement (deformable_transformer.py:734-756; inverse_sigmoid util/misc.py:575-579):
//   out = sigmoid(delta + log(clamp(x,1e-3) / clamp(1-x,1e-3))),  x = clamp(ref, 0, 1)
__global__ __launch_bounds__(256) void box_refine_kernel(const float* __restrict__ delta, const float* __restrict__ ref,
                                                         float* __restrict__ out, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = fminf(fmaxf(ref[i], 0.f), 1.f);
    const float x1 = fmaxf(x, 1e-3f), x2 = fmaxf(1.f - x, 1e-3f);
    const float u = delta[i] + logf(x1 / x2);
    out[i] = 1.f / (1.f + expf(-u));
}

// Output layer of a 3-layer box MLP (256 -> 4, models/dino/dino.py MLP) fused with what consumes it:
//   mode 0: out = sigmoid(h W^T + b + inverse_sigmoid(ref))      iterative refinement (deformable_transformer.py:734-756)
//   mode 1: out = h W^T + b + ref                                two-stage initial boxes, unsigmoided (:352-356)
// One wavefront per row: lane l holds channels 4l..4l+3 of h and of the four weight rows; four wave reductions.
// (As a GEMM with N = 4 this was a 24 us launch of the fp32 MFMA kernel per decoder layer, plus the refine launch.)
__global__ __launch_bounds__(256) void box_head_refine_kernel(const float* __restrict__ h, const float* __restrict__ W,
                                                              const float* __restrict__ bias, const float* __restrict__ ref,
                                                              float* __restrict__ out, long rows, int mode)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4 x = *reinterpret_cast<const float4*>(h + row * 256 + 4 * lane);
    float d[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const float4 w = *reinterpret_cast<const float4*>(W + o * 256 + 4 * lane);
        d[o] = wave_sum(x.x * w.x + x.y * w.y + x.z * w.z + x.w * w.w);
    }
    if (lane < 4) {
        const float delta = (lane == 0 ? d[0] : lane == 1 ? d[1] : lane == 2 ? d[2] : d[3]) + bias[lane];
        const float r = ref[row * 4 + lane];
        float y;
        if (mode == 0) {
            const float xx = fminf(fmaxf(r, 0.f), 1.f);
            const float x1 = fmaxf(xx, 1e-3f), x2 = fmaxf(1.f - xx, 1e-3f);
            const float u = delta + logf(x1 / x2);
            y = 1.f / (1.f + expf(-u));
        } else y = delta + r;
        out[row * 4 + lane] = y;
    }
}


// Two-stage query selection, the gathers after the top-k (deformable_transformer.py:347-356; utils.py proposals): for every
// selected token copy its output_memory row (bf16 engine: the [hi | lo | hi] image, 768 elements, plus bf16(hi + lo) as the box
// MLP's input; fp32 engine: the 256 fp32 channels), its proposal logits, and sigmoid(proposal) (`init_box_proposal`, :355).
// One wavefront per selected row; replaces two torch.gather launches, a slice-add, a cast and a sigmoid.
template <bool SPLIT>
__global__ __launch_bounds__(256) void two_stage_gather_kernel(const void* __restrict__ om, const float* __restrict__ proposals,
                                                               const long* __restrict__ idx, void* __restrict__ sel_raw,
                                                               uint16_t* __restrict__ sel_x, float* __restrict__ prop_sel,
                                                               float* __restrict__ init_box, int S, int k, long rows)
{
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const long b = r / k;
    const long src = b * S + idx[r];
    if (SPLIT) {
        const uint4* s4 = reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(om) + src * 768);
        uint4* d4 = reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(sel_raw) + r * 768);
        const uint4 a = s4[lane];                                  // chunks 0..63: hi (0..31) and lo (32..63)
        d4[lane] = a;
        if (lane < 32) d4[64 + lane] = s4[64 + lane];              // chunks 64..95: hi again
        // lane < 32 holds hi[8 lane ..], lane + 32 holds lo[8 lane ..]
        const uint32_t av[4] = {a.x, a.y, a.z, a.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t lo = (uint32_t)__shfl(av[e], (lane & 31) + 32, 64);
            const float x0 = h16_lo(av[e]) + h16_lo(lo);
            const float x1 = h16_hi(av[e]) + h16_hi(lo);
            o[e] = pack_bf16x2(x0, x1);
        }
        if (lane < 32) reinterpret_cast<uint4*>(sel_x + r * 256)[lane] = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
        const float4* s4 = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(om) + src * 256);
        reinterpret_cast<float4*>(reinterpret_cast<float*>(sel_raw) + r * 256)[lane] = s4[lane];
    }
    if (lane == 0) {
        const float4 p = *reinterpret_cast<const float4*>(proposals + src * 4);
        *reinterpret_cast<float4*>(prop_sel + r * 4) = p;
        *reinterpret_cast<float4*>(init_box + r * 4) = make_float4(1.f / (1.f + expf(-p.x)), 1.f / (1.f + expf(-p.y)),
                                                                  1.f / (1.f + expf(-p.z)), 1.f / (1.f + expf(-p.w)));
    }
}

}  // namespace dtlr

using namespace dtlr;

extern "C" int dtlr_two_stage_gather(const void* om, const float* proposals, const long* idx, void* sel_raw, void* sel_x,
                                     float* prop_sel, float* init_box, int B, int S, int k, int dtype, void* stream)
{
    clear_stale_error();
    if (!om || !proposals || !idx || !sel_raw || !prop_sel || !init_box) return DTLR_EINVAL;
    if (B <= 0 || S <= 0 || k <= 0) return DTLR_EINVAL;
    const long rows = (long)B * k;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if (dtype == DTLR_H16) {
        if (!sel_x) return DTLR_EINVAL;
        hipLaunchKernelGGL(two_stage_gather_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, om, proposals, idx, sel_raw,
                           (uint16_t*)sel_x, prop_sel, init_box, S, k, rows);
    } else if (dtype == DTLR_F32) {
        hipLaunchKernelGGL(two_stage_gather_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, om, proposals, idx, sel_raw,
                           (uint16_t*)nullptr, prop_sel, init_box, S, k, rows);
    } else return DTLR_EDTYPE;
    return check_launch();
}

extern "C" int dtlr_box_head_refine(const float* h, const float* W, const float* bias, const float* ref, float* out,
                                    long rows, int hidden, int mode, void* stream)
{
    clear_stale_error();
    if (!h || !W || !bias || !ref || !out) return DTLR_EINVAL;
    if (rows <= 0 || (mode != 0 && mode != 1)) return DTLR_EINVAL;
    if (hidden != 256) return DTLR_ESHAPE;
    hipLaunchKernelGGL(box_head_refine_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, h, W, bias, ref, out, rows, mode);
    return check_launch();
}

extern "C" int dtlr_decoder_query_prep(const float* ref, const float* valid_ratios, const float* dim_t,
                                       float* ref_in, void* sine, int B, int nq, int L, int sine_dtype, void* stream)
{
    clear_stale_error();
    if (!ref || !valid_ratios || !dim_t || !ref_in || !sine) return DTLR_EINVAL;
    if (B <= 0 || nq <= 0 || L <= 0 || L > 64) return DTLR_EINVAL;
    const long total = (long)B * nq * 64;
    const long grid = (total + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    if (sine_dtype == DTLR_H16)
        hipLaunchKernelGGL((query_prep_kernel<uint16_t>), dim3((unsigned)grid), dim3(256), 0, st, ref, valid_ratios, dim_t, ref_in, (uint16_t*)sine, nq, L, total);
    else if (sine_dtype == DTLR_F32)
        hipLaunchKernelGGL((query_prep_kernel<float>), dim3((unsigned)grid), dim3(256), 0, st, ref, valid_ratios, dim_t, ref_in, (float*)sine, nq, L, total);
    else return DTLR_EDTYPE;
    return check_launch();
}

extern "C" int dtlr_box_refine(const float* delta, const float* ref, float* out, long n, void* stream)
{
    clear_stale_error();
    if (!delta || !ref || !out) return DTLR_EINVAL;
    if (n <= 0) return DTLR_EINVAL;
    hipLaunchKernelGGL(box_refine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, delta, ref, out, n);
    return check_launch();
}
