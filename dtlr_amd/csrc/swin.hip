// Swin Transformer backbone kernels (models/dino/swin_transformer.py; selected by `backbone = 'swin_*'`, models/dino/backbone.py:
// 172-205 -- SURVEY.md section 8 row f.4).  Token-major NHWC like the rest of the engine: a stage's activations are [B, H, W, C].
//   dtlr_swin_patch_embed   PatchEmbed (swin_transformer.py:393-432): 4x4 / stride 4 convolution of the NCHW fp32 image (zero padding
//                           to a multiple of 4) + LayerNorm(E), one launch
//   dtlr_swin_window_attn   the attention core of a SwinTransformerBlock (:116-147, 191-236): cyclic shift, window partition, padding
//                           to a multiple of the window, q k^T + relative position bias + shifted-window mask, softmax, p v, window
//                           reverse, un-shift -- all by index arithmetic inside one kernel, on the matrix cores (bf16: 16x16x32;
//                           fp32 engine: exact 16x16x4 f32).  The qkv projection before it and the output projection (+ residual)
//                           after it are GEMMs (gemm.hip).
//   dtlr_swin_patch_merge   PatchMerging (:250-288): 2x2 neighbourhood gather (zero padding for odd sizes) + LayerNorm(4C); the
//                           4C -> 2C reduction that follows is a GEMM.
// The MLP's GELU is an epilogue of the GEMM (EPI_GELU), LayerNorm of arbitrary C the generic row kernel of norm.hip.
#include "dtlr_common.h"

namespace dtlr {

typedef __attribute__((ext_vector_type(8))) h16_hw_t sw_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float sw_f32x4_t;

// ------------------------------------------------------------------------------------------------------------- patch embed
// x [B,3,H,W] fp32 ; w [48][E] fp32 (k = (c*4 + dy)*4 + dx, k-major) ; b, gamma, beta [E] ; out [B,Hp,Wp,E], Hp = ceil(H/4).
// Workgroup = 64 output pixels of one row; thread (pixel, quarter) accumulates EQ = E/4 channels; LayerNorm over the 4 threads.
template <typename OT, int EQ>
__global__ __launch_bounds__(256) void swin_patch_embed_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, OT* __restrict__ out,
                                                               int H, int W, int Hp, int Wp, float eps)
{
    constexpr int E = 4 * EQ;
    extern __shared__ __attribute__((aligned(16))) float sw_pe[];
    float* patch = sw_pe;                 // [64][49] (padded)
    float* wl = sw_pe + 64 * 49;          // [48][E]
    const int b = blockIdx.z, i = blockIdx.y, j0 = blockIdx.x * 64;
    for (int t = threadIdx.x; t < 64 * 48; t += 256) {
        const int px = t / 48, k = t % 48;
        const int c = k >> 4, dy = (k >> 2) & 3, dx = k & 3;
        const int y = 4 * i + dy, xx = 4 * (j0 + px) + dx;
        patch[px * 49 + k] = (y < H && xx < W && j0 + px < Wp) ? x[(((long)b * 3 + c) * H + y) * W + xx] : 0.f;
    }
    for (int t = threadIdx.x; t < 48 * E; t += 256) wl[t] = w[t];
    __syncthreads();
    const int px = threadIdx.x >> 2, q = threadIdx.x & 3;
    float acc[EQ];
#pragma unroll
    for (int e = 0; e < EQ; ++e) acc[e] = bias[q * EQ + e];
    for (int k = 0; k < 48; ++k) {
        const float xv = patch[px * 49 + k];
        const float* wr = wl + k * E + q * EQ;
#pragma unroll
        for (int e = 0; e < EQ; ++e) acc[e] = fmaf(xv, wr[e], acc[e]);
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EQ; ++e) s += acc[e];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    const float mean = s * (1.0f / E);
    float v = 0.f;
#pragma unroll
    for (int e = 0; e < EQ; ++e) { const float d = acc[e] - mean; v += d * d; }
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    const float rstd = rsqrtf(v * (1.0f / E) + eps);
    if (j0 + px < Wp) {
        OT* o = out + (((long)b * Hp + i) * Wp + j0 + px) * E + q * EQ;
#pragma unroll
        for (int e = 0; e < EQ; ++e) {
            const float y = (acc[e] - mean) * rstd * gamma[q * EQ + e] + beta[q * EQ + e];
            if constexpr (sizeof(OT) == 2) o[e] = f32_to_bf16(y);
            else o[e] = y;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ patch merging
// x [B,H,W,C] -> y [B,H2,W2,4C] = LayerNorm(cat(x[2i,2j], x[2i+1,2j], x[2i,2j+1], x[2i+1,2j+1])) (zeros beyond H, W).
// One wavefront per output token; lane l owns elements 4l + 256 g of the 4C-long row (C % 4 == 0, 4C <= 3072).
template <typename T>
__global__ __launch_bounds__(256) void swin_patch_merge_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, T* __restrict__ y,
                                                               int H, int W, int C, int H2, int W2, long rows, float eps)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int C4 = 4 * C;
    const int j = (int)(row % W2), i = (int)((row / W2) % H2);
    const long b = row / ((long)W2 * H2);
    float v[12][4];
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 12; ++g) {
        const int e = g * 256 + 4 * lane;
        v[g][0] = v[g][1] = v[g][2] = v[g][3] = 0.f;
        if (e < C4) {
            const int part = e / C, c = e % C;
            const int yy = 2 * i + (part & 1), xx = 2 * j + (part >> 1);
            if (yy < H && xx < W) {
                const T* p = x + ((b * H + yy) * W + xx) * (long)C + c;
                if constexpr (sizeof(T) == 2) {
                    const uint2 t = *reinterpret_cast<const uint2*>(p);
                    v[g][0] = h16_lo(t.x); v[g][1] = h16_hi(t.x);
                    v[g][2] = h16_lo(t.y); v[g][3] = h16_hi(t.y);
                } else {
                    const float4 t = *reinterpret_cast<const float4*>(p);
                    v[g][0] = t.x; v[g][1] = t.y; v[g][2] = t.z; v[g][3] = t.w;
                }
            }
            s += (v[g][0] + v[g][1]) + (v[g][2] + v[g][3]);
        }
    }
    const float mean = wave_sum(s) / (float)C4;
    float q = 0.f;
#pragma unroll
    for (int g = 0; g < 12; ++g)
        if (g * 256 + 4 * lane < C4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = v[g][r] - mean; q += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)C4 + eps);
    T* yr = y + row * C4;
#pragma unroll
    for (int g = 0; g < 12; ++g) {
        const int e = g * 256 + 4 * lane;
        if (e < C4) {
            const float4 ga = *reinterpret_cast<const float4*>(gamma + e), be = *reinterpret_cast<const float4*>(beta + e);
            const float o0 = (v[g][0] - mean) * rstd * ga.x + be.x, o1 = (v[g][1] - mean) * rstd * ga.y + be.y;
            const float o2 = (v[g][2] - mean) * rstd * ga.z + be.z, o3 = (v[g][3] - mean) * rstd * ga.w + be.w;
            if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(yr + e) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
            else *reinterpret_cast<float4*>(yr + e) = make_float4(o0, o1, o2, o3);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- window attention
// qkv [B,H,W,3C] (q | k | v, each head-major with head_dim 32) ; qkv_b [3C] fp32 (what a padded position projects to: the reference pads
// AFTER norm1, so a padded token's q/k/v are the bare biases) ; rpb [nH][NQ][NK] fp32 = relative position bias, zero padded
// (NQ = ceil16(N), NK = ceil32(N), N = ws*ws) ; out [B,H,W,C].  Workgroup = (window, head); 4 wavefronts take the 16-query tiles.
//   S^T = K Q^T  ->  lane (n = query, g): keys 4g..4g+3 of each 16-key tile: the row softmax is in-lane + two xor shuffles over g;
//   O^T += V^T P^T with the P registers used AS the B operand (the k index of the second product is a permutation of the keys that
//   matches the accumulator layout of the first: no data movement between the two products).
struct SwinAttnP { int H, W, C, nH, ws, shift, Hp, Wp, nWw, N, NQ, NK; float scale; };

template <typename T> struct SwinLds;
template <> struct SwinLds<uint16_t> {                        // bf16 operands
    static constexpr int VT_PAD = 4;
    static __host__ __device__ int bytes(int NQ, int NK) { return NQ * 64 + NK * 64 + 32 * (NK + VT_PAD) * 2 + NK * 8; }
};
template <> struct SwinLds<float> {
    static __host__ __device__ int bytes(int NQ, int NK) { return (NQ + 2 * NK) * 33 * 4 + NK * 8; }
};

template <typename T>
__global__ __launch_bounds__(256) void swin_window_attn_kernel(const T* __restrict__ qkv, const float* __restrict__ qkv_b,
                                                               const float* __restrict__ rpb, T* __restrict__ out, SwinAttnP P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sw_smem[];
    constexpr bool BF = sizeof(T) == 2;
    const int N = P.N, NQ = P.NQ, NK = P.NK, ws = P.ws, C = P.C;
    const int win = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int wy = win / P.nWw, wx = win % P.nWw;
    // LDS carve-up
    unsigned char* Qs = sw_smem;
    unsigned char* Ks = Qs + (BF ? NQ * 64 : NQ * 33 * 4);
    unsigned char* Vs = Ks + (BF ? NK * 64 : NK * 33 * 4);
    int* tok = reinterpret_cast<int*>(Vs + (BF ? 32 * (NK + SwinLds<uint16_t>::VT_PAD) * 2 : NK * 33 * 4));
    int* reg = tok + NK;

    // ---- window geometry: token of every in-window position (shifted frame -> original frame), region id for the shift mask ----
    for (int i = threadIdx.x; i < NK; i += 256) {
        int t = -1, r = 0;
        if (i < N) {
            const int a = i / ws, c = i % ws;
            const int ys = wy * ws + a, xs = wx * ws + c;                     // shifted-frame coordinates
            const int y = (ys + P.shift) % P.Hp, xx = (xs + P.shift) % P.Wp;   // torch.roll(x, -shift): shifted[i] = x[(i + shift) mod Hp]
            t = (y < P.H && xx < P.W) ? y * P.W + xx : -2;                     // -2: padding position (projects to the biases)
            if (P.shift > 0) {
                const int rh = ys < P.Hp - ws ? 0 : (ys < P.Hp - P.shift ? 1 : 2);
                const int rw = xs < P.Wp - ws ? 0 : (xs < P.Wp - P.shift ? 1 : 2);
                r = 3 * rh + rw;
            }
        }
        tok[i] = t;
        reg[i] = r;
    }
    __syncthreads();
    // ---- gather q (scaled), k, v of head h into LDS ---------------------------------------------------------------------------
    const T* base = qkv + (long)b * P.H * P.W * 3 * C;
    for (int it = threadIdx.x; it < NK * 12; it += 256) {
        const int i = it / 12, rem = it % 12, part = rem >> 2, ch = rem & 3;      // part 0 q, 1 k, 2 v ; 8 channels per chunk
        if (part == 0 && i >= NQ) continue;
        const int t = tok[i];
        float v[8];
        const int col = part * C + h * 32 + ch * 8;
        if (t >= 0) {
            const T* p = base + (long)t * 3 * C + col;
            if constexpr (BF) {
                const uint4 d = *reinterpret_cast<const uint4*>(p);
                const uint32_t wv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] = h16_lo(wv[e]); v[2 * e + 1] = h16_hi(wv[e]); }
            } else {
                const float4 d0 = reinterpret_cast<const float4*>(p)[0], d1 = reinterpret_cast<const float4*>(p)[1];
                v[0] = d0.x; v[1] = d0.y; v[2] = d0.z; v[3] = d0.w; v[4] = d1.x; v[5] = d1.y; v[6] = d1.z; v[7] = d1.w;
            }
        } else if (t == -2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = qkv_b[col + e];
            if constexpr (BF) {                                                    // the engine's qkv GEMM would have rounded these to bf16
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = bf16_to_f32(f32_to_bf16(v[e]));
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        if (part == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= P.scale;
        }
        if constexpr (BF) {
            const uint4 pk = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
            if (part < 2) {
                unsigned char* dst = (part == 0 ? Qs : Ks) + i * 64 + ((ch ^ ((i >> 2) & 3)) * 16);   // swizzled 16-byte slot
                *reinterpret_cast<uint4*>(dst) = pk;
            } else {
                uint16_t* vt = reinterpret_cast<uint16_t*>(Vs);
                const int stride = NK + SwinLds<uint16_t>::VT_PAD;
#pragma unroll
                for (int e = 0; e < 8; ++e) vt[(ch * 8 + e) * stride + i] = f32_to_bf16(v[e]);
            }
        } else {
            float* dst = reinterpret_cast<float*>(part == 0 ? Qs : part == 1 ? Ks : Vs) + i * 33 + ch * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[e] = v[e];
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int KT = NK / 16;
    const float* bias_h = rpb + (long)h * NQ * NK;
    for (int qt = wave; qt < NQ / 16; qt += 4) {
        const int qi = 16 * qt + n;                               // this lane's query
        const int rq = reg[min(qi, NK - 1)];
        float s[12][4];                                           // up to 12 key tiles (NK <= 192: window 13)
        float mx = -__builtin_huge_valf();
#pragma unroll
        for (int kt = 0; kt < 12; ++kt) {
            if (kt < KT) {
                sw_f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                if constexpr (BF) {
                    const int km = 16 * kt + n;                   // A-operand row = key (lane index n plays m)
                    const uint4 kf = *reinterpret_cast<const uint4*>(Ks + km * 64 + ((g ^ ((km >> 2) & 3)) * 16));
                    const uint4 qf = *reinterpret_cast<const uint4*>(Qs + qi * 64 + ((g ^ ((qi >> 2) & 3)) * 16));
                    acc = DTLR_MFMA_16x16x32_H16(__builtin_bit_cast(sw_bf16x8_t, kf), __builtin_bit_cast(sw_bf16x8_t, qf), acc, 0, 0, 0);
                } else {
                    const float* kr = reinterpret_cast<const float*>(Ks) + (16 * kt + n) * 33;
                    const float* qr = reinterpret_cast<const float*>(Qs) + qi * 33;
#pragma unroll
                    for (int st = 0; st < 8; ++st) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[4 * st + g], qr[4 * st + g], acc, 0, 0, 0);
                }
                const float4 bb = *reinterpret_cast<const float4*>(bias_h + (long)qi * NK + 16 * kt + 4 * g);
                const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kj = 16 * kt + 4 * g + r;
                    float v = acc[r] + bv[r];
                    if (P.shift > 0 && reg[kj] != rq) v += -100.0f;
                    if (kj >= N) v = -__builtin_huge_valf();
                    s[kt][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 12; ++kt)
            if (kt < KT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[kt][r] = __expf(s[kt][r] - mx); sum += s[kt][r]; }
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        sw_f32x4_t o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if constexpr (BF) {
            const uint16_t* vt = reinterpret_cast<const uint16_t*>(Vs);
            const int stride = NK + SwinLds<uint16_t>::VT_PAD;
#pragma unroll
            for (int kb = 0; kb < 6; ++kb) {
                if (2 * kb < KT) {
                    const uint4 pf = make_uint4(pack_bf16x2(s[2 * kb][0], s[2 * kb][1]), pack_bf16x2(s[2 * kb][2], s[2 * kb][3]),
                                                pack_bf16x2(s[2 * kb + 1][0], s[2 * kb + 1][1]), pack_bf16x2(s[2 * kb + 1][2], s[2 * kb + 1][3]));
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const uint16_t* vr = vt + (16 * dt + n) * stride + 32 * kb + 4 * g;          // V^T row d = 16 dt + n
                        const uint2 lo = *reinterpret_cast<const uint2*>(vr), hi = *reinterpret_cast<const uint2*>(vr + 16);
                        const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
                        o[dt] = DTLR_MFMA_16x16x32_H16(__builtin_bit_cast(sw_bf16x8_t, vf), __builtin_bit_cast(sw_bf16x8_t, pf), o[dt], 0, 0, 0);
                    }
                }
            }
        } else {
            const float* vs = reinterpret_cast<const float*>(Vs);
#pragma unroll
            for (int kt = 0; kt < 12; ++kt)
                if (kt < KT) {
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const float* vr = vs + (16 * kt + 4 * g + st) * 33;                          // key 16 kt + 4 g + st
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[16 * dt + n], s[kt][st], o[dt], 0, 0, 0);
                    }
                }
        }
        const int t = qi < NK ? tok[qi] : -1;
        if (qi < N && t >= 0) {
            T* op = out + ((long)b * P.H * P.W + t) * C + h * 32 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const float v0 = o[dt][0] * inv, v1 = o[dt][1] * inv, v2 = o[dt][2] * inv, v3 = o[dt][3] * inv;
                if constexpr (BF) *reinterpret_cast<uint2*>(op + 16 * dt) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                else *reinterpret_cast<float4*>(op + 16 * dt) = make_float4(v0, v1, v2, v3);
            }
        }
    }
}

}  // namespace dtlr

using namespace dtlr;

extern "C" int dtlr_swin_patch_embed(const float* x, const float* w_kE, const float* bias, const float* gamma, const float* beta,
                                     void* out, int B, int H, int W, int E, float eps, int out_dtype, void* stream)
{
    clear_stale_error();
    if (!x || !w_kE || !bias || !gamma || !beta || !out) return DTLR_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0) return DTLR_EINVAL;
    if (out_dtype != DTLR_F32 && out_dtype != DTLR_H16) return DTLR_EDTYPE;
    const int Hp = (H + 3) / 4, Wp = (W + 3) / 4;
    if (Hp > 65535 || B > 65535) return DTLR_ESHAPE;
    const dim3 grid((Wp + 63) / 64, Hp, B);
    const size_t lds = (size_t)(64 * 49 + 48 * E) * 4;
    hipStream_t st = (hipStream_t)stream;
#define PE_LAUNCH(OT, EQ) hipLaunchKernelGGL((swin_patch_embed_kernel<OT, EQ>), grid, dim3(256), lds, st, x, w_kE, bias, gamma, beta, (OT*)out, H, W, Hp, Wp, eps)
#define PE_CASE(EQ) case 4 * EQ: if (out_dtype == DTLR_H16) PE_LAUNCH(uint16_t, EQ); else PE_LAUNCH(float, EQ); break;
    switch (E) {
        PE_CASE(8) PE_CASE(16) PE_CASE(24) PE_CASE(32) PE_CASE(48)
    default: return DTLR_ESHAPE;                               // embed_dim 32 / 64 / 96 / 128 / 192
    }
#undef PE_CASE
#undef PE_LAUNCH
    return check_launch();
}

extern "C" int dtlr_swin_patch_merge(const void* x, const float* gamma, const float* beta, void* y, int B, int H, int W, int C,
                                     float eps, int dtype, void* stream)
{
    clear_stale_error();
    if (!x || !gamma || !beta || !y) return DTLR_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return DTLR_EINVAL;
    if ((C & 3) || 4 * C > 3072) return DTLR_ESHAPE;
    const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    const long rows = (long)B * H2 * W2;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DTLR_H16)
        hipLaunchKernelGGL(swin_patch_merge_kernel<uint16_t>, dim3(grid), dim3(256), 0, st, (const uint16_t*)x, gamma, beta, (uint16_t*)y, H, W, C, H2, W2, rows, eps);
    else if (dtype == DTLR_F32)
        hipLaunchKernelGGL(swin_patch_merge_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, gamma, beta, (float*)y, H, W, C, H2, W2, rows, eps);
    else return DTLR_EDTYPE;
    return check_launch();
}

extern "C" int dtlr_swin_window_attn(const void* qkv, const float* qkv_bias, const float* rpb, void* out,
                                     int B, int H, int W, int C, int n_heads, int window, int shift, int dtype, void* stream)
{
    clear_stale_error();
    if (!qkv || !qkv_bias || !rpb || !out) return DTLR_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || window <= 0 || shift < 0 || shift >= window || n_heads <= 0) return DTLR_EINVAL;
    if (C != 32 * n_heads) return DTLR_ESHAPE;                  // head_dim 32 (every reference variant)
    SwinAttnP P;
    P.H = H; P.W = W; P.C = C; P.nH = n_heads; P.ws = window; P.shift = shift;
    P.Hp = (H + window - 1) / window * window; P.Wp = (W + window - 1) / window * window;
    P.nWw = P.Wp / window;
    P.N = window * window; P.NQ = (P.N + 15) / 16 * 16; P.NK = (P.N + 31) / 32 * 32;
    P.scale = 0.17677669529663687f;                            // 32 ** -0.5
    if (P.NK > 192) return DTLR_ESHAPE;                         // window <= 13
    const int nW = (P.Hp / window) * P.nWw;
    if (n_heads > 65535 || B > 65535) return DTLR_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DTLR_H16) {
        const int lds = SwinLds<uint16_t>::bytes(P.NQ, P.NK);
        hipLaunchKernelGGL(swin_window_attn_kernel<uint16_t>, dim3(nW, n_heads, B), dim3(256), lds, st, (const uint16_t*)qkv, qkv_bias, rpb, (uint16_t*)out, P);
    } else if (dtype == DTLR_F32) {
        const int lds = SwinLds<float>::bytes(P.NQ, P.NK);
        static DevOnce once;
        if (once.first()) { (void)hipFuncSetAttribute((const void*)swin_window_attn_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); (void)hipGetLastError(); }
        if (lds > 96 * 1024) return DTLR_ESHAPE;
        hipLaunchKernelGGL(swin_window_attn_kernel<float>, dim3(nW, n_heads, B), dim3(256), lds, st, (const float*)qkv, qkv_bias, rpb, (float*)out, P);
    } else return DTLR_EDTYPE;
    return check_launch();
}
