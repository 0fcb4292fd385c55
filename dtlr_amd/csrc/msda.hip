// Multi-scale deformable attention forward for gfx950.
//
// What it computes (reference: models/dino/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-84,237-299):
//   out[b,q,m,:] = sum_l sum_p A[b,q,m,l,p] * bilinear(value_l[b,:,m,:], (x*W_l-0.5, y*H_l-0.5))
// with zero padding outside the map.  The reference runs one thread per output SCALAR (block 1024);
// here a thread owns VEC contiguous channels of one (b,q,m) so every corner fetch is one 16-byte
// load and the D/VEC lanes of a head form one coalesced segment: with M=8, D=32 (fp32, VEC=4) a
// 64-lane wavefront is exactly one query -- its 8 heads x 8 lanes -- so the 384 floats of
// loc/attn of that query are one contiguous 1.5 KB region read through same-address broadcast
// loads, and the 256-float output row is one coalesced 1 KB store.
//
// HBM roofline: compulsory bytes per call = value + loc + attn + out.  The
// 4-corner gather re-reads value ~18x (Lq*M*L*P*4*D elements); that traffic is served by L2 (the
// per-image value map is 5.6 MB fp32 / 2.8 MB bf16, blocks of one image are launched adjacently),
// so HBM traffic stays near compulsory -- see DESIGN.md for the measured FETCH_SIZE.
#include "dtlr_common.h"

namespace dtlr {

thread_local int g_last_hip_error = 0;

template <typename T> struct Elem;
template <> struct Elem<float>  { using acc = float;  static __device__ __forceinline__ float ld(const float* p) { return *p; } };
template <> struct Elem<double> { using acc = double; static __device__ __forceinline__ double ld(const double* p) { return *p; } };

// ---- vector fetch of VEC contiguous channels into accumulator-typed registers ------------------
template <typename T, int VEC> struct Vec;
template <> struct Vec<float, 4> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
    static __device__ __forceinline__ void load_sel(const float* p, bool ok, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = ok ? t.x : 0.f; v[1] = ok ? t.y : 0.f; v[2] = ok ? t.z : 0.f; v[3] = ok ? t.w : 0.f; }
};
template <> struct Vec<float, 2> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[2]) {
        const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[2]) {
        *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
    static __device__ __forceinline__ void load_sel(const float* p, bool ok, float (&v)[2]) {
        const float2 t = *reinterpret_cast<const float2*>(p); v[0] = ok ? t.x : 0.f; v[1] = ok ? t.y : 0.f; }
};
template <> struct Vec<float, 1> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[1]) { *p = v[0]; }
    static __device__ __forceinline__ void load_sel(const float* p, bool ok, float (&v)[1]) { const float t = *p; v[0] = ok ? t : 0.f; }
};
template <> struct Vec<double, 1> {
    static __device__ __forceinline__ void load(const double* p, double (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void store(double* p, const double (&v)[1]) { *p = v[0]; }
    static __device__ __forceinline__ void load_sel(const double* p, bool ok, double (&v)[1]) { const double t = *p; v[0] = ok ? t : 0.0; }
};
template <> struct Vec<double, 2> {
    static __device__ __forceinline__ void load(const double* p, double (&v)[2]) {
        const double2 t = *reinterpret_cast<const double2*>(p); v[0] = t.x; v[1] = t.y; }
    static __device__ __forceinline__ void store(double* p, const double (&v)[2]) {
        *reinterpret_cast<double2*>(p) = make_double2(v[0], v[1]); }
    static __device__ __forceinline__ void load_sel(const double* p, bool ok, double (&v)[2]) {
        const double2 t = *reinterpret_cast<const double2*>(p); v[0] = ok ? t.x : 0.0; v[1] = ok ? t.y : 0.0; }
};
// bf16 storage (uint16_t), f32 arithmetic: 8 channels = one 16-byte load
template <> struct Vec<uint16_t, 8> {
    static __device__ __forceinline__ void load(const uint16_t* p, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = h16_lo(w[i]); v[2 * i + 1] = h16_hi(w[i]); }
    }
    static __device__ __forceinline__ void store(uint16_t* p, const float (&v)[8]) {
        *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]),
                                                  pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])); }
    static __device__ __forceinline__ void load_sel(const uint16_t* p, bool ok, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {ok ? t.x : 0u, ok ? t.y : 0u, ok ? t.z : 0u, ok ? t.w : 0u};   // select on packed words
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = h16_lo(w[i]); v[2 * i + 1] = h16_hi(w[i]); }
    }
};
template <> struct Vec<uint16_t, 4> {
    static __device__ __forceinline__ void load(const uint16_t* p, float (&v)[4]) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = h16_lo(t.x); v[1] = h16_hi(t.x);
        v[2] = h16_lo(t.y); v[3] = h16_hi(t.y); }
    static __device__ __forceinline__ void store(uint16_t* p, const float (&v)[4]) {
        *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])); }
    static __device__ __forceinline__ void load_sel(const uint16_t* p, bool ok, float (&v)[4]) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        const uint32_t a = ok ? t.x : 0u, b = ok ? t.y : 0u;
        v[0] = h16_lo(a); v[1] = h16_hi(a);
        v[2] = h16_lo(b); v[3] = h16_hi(b); }
};
template <> struct Vec<uint16_t, 1> {
    static __device__ __forceinline__ void load(const uint16_t* p, float (&v)[1]) { v[0] = bf16_to_f32(*p); }
    static __device__ __forceinline__ void store(uint16_t* p, const float (&v)[1]) { *p = f32_to_bf16(v[0]); }
    static __device__ __forceinline__ void load_sel(const uint16_t* p, bool ok, float (&v)[1]) { const uint16_t t = *p; v[0] = ok ? bf16_to_f32(t) : 0.f; }
};

// One sampling point: adds A * bilinear(...) for VEC channels.  `base` points at
// value[b, level_start, m, c0]; row stride = W*M*D elements, column stride = M*D elements.
// BRANCH-FREE on purpose: the four corner fetches are always issued (addresses clamped into the map)
// and an invalid corner's data is replaced by zero with v_cndmask.  With `if (valid) load` hipcc puts
// each load in its own exec-masked block followed by s_waitcnt vmcnt(0) -- 64 serialized L2 round
// trips per thread (first version of this kernel: 0.56 ms per encoder call); unconditional loads let
// the 4 corners x several points be in flight together.  Same arithmetic as the reference
// (cuh:33-84): v_i = 0 for corners outside the map, val = w1 v1 + w2 v2 + w3 v3 + w4 v4, col += val * A.
template <typename T, typename A, int VEC>
__device__ __forceinline__ void sample_point(const T* __restrict__ base, int H, int W, int MD,
                                             A loc_w, A loc_h, A weight, A (&col)[VEC]) {
    const A h_im = loc_h * (A)H - (A)0.5;
    const A w_im = loc_w * (A)W - (A)0.5;
    const bool inside = h_im > (A)-1 && w_im > (A)-1 && h_im < (A)H && w_im < (A)W;
    const A hf = floor(h_im), wf = floor(w_im);
    const A lh = h_im - hf, lw = w_im - wf, hh = (A)1 - lh, hw = (A)1 - lw;
    // clamp in floating point first so the int conversion is defined for far-away / non-finite locations
    const int h_low = (int)fmin(fmax(hf, (A)-1), (A)H), w_low = (int)fmin(fmax(wf, (A)-1), (A)W);
    const int h_high = h_low + 1, w_high = w_low + 1;
    const bool top = inside && h_low >= 0, bot = inside && h_high <= H - 1;
    const bool left = w_low >= 0, right = w_high <= W - 1;
    // clamp BOTH ways: for a far-outside sample h_low can be H (then h_high = H+1): data unused, address must stay in the map
    const int h0 = min(max(h_low, 0), H - 1), h1 = max(min(h_high, H - 1), 0), w0 = min(max(w_low, 0), W - 1), w1 = max(min(w_high, W - 1), 0);
    const long row_stride = (long)W * MD;
    const T* r0 = base + (long)h0 * row_stride;
    const T* r1 = base + (long)h1 * row_stride;
    A v1[VEC], v2[VEC], v3[VEC], v4[VEC];
    Vec<T, VEC>::load_sel(r0 + (long)w0 * MD, top && left, v1);
    Vec<T, VEC>::load_sel(r0 + (long)w1 * MD, top && right, v2);
    Vec<T, VEC>::load_sel(r1 + (long)w0 * MD, bot && left, v3);
    Vec<T, VEC>::load_sel(r1 + (long)w1 * MD, bot && right, v4);
    const A w1_ = hh * hw, w2_ = hh * lw, w3_ = lh * hw, w4_ = lh * lw;
#pragma unroll
    for (int i = 0; i < VEC; ++i) col[i] += (w1_ * v1[i] + w2_ * v2[i] + w3_ * v3[i] + w4_ * v4[i]) * weight;
}

// Hot path (P = 4): one LEVEL at a time -- geometry of its 4 points first, then all 16 corner fetches
// issued back to back (64 data VGPRs in flight per lane), then the arithmetic.  A scheduling barrier
// after each level stops hipcc from hoisting the next level's loads too (unbounded hoisting costs 256
// VGPRs or kilobytes of scratch); 16 independent 16-byte loads per lane x 3-4 waves per SIMD is what
// covers the L2 latency of the gather.
template <typename T, int VEC> struct Raw;
template <> struct Raw<float, 4> {
    using raw_t = float4;
    static constexpr int GROUP = 4;        // points whose corner fetches are in flight together
    static __device__ __forceinline__ raw_t ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void unpack(raw_t t, bool ok, float (&v)[4]) {
        v[0] = ok ? t.x : 0.f; v[1] = ok ? t.y : 0.f; v[2] = ok ? t.z : 0.f; v[3] = ok ? t.w : 0.f; }
};
template <> struct Raw<uint16_t, 4> {      // bf16: 4 channels = one 8-byte load; same register shape as fp32
    using raw_t = uint2;
    static constexpr int GROUP = 4;
    static __device__ __forceinline__ raw_t ld(const uint16_t* p) { return *reinterpret_cast<const uint2*>(p); }
    static __device__ __forceinline__ void unpack(raw_t t, bool ok, float (&v)[4]) {
        const uint32_t a = ok ? t.x : 0u, b = ok ? t.y : 0u;          // select on the packed words
        v[0] = h16_lo(a); v[1] = h16_hi(a);
        v[2] = h16_lo(b); v[3] = h16_hi(b); }
};

template <typename T, int VEC>
__device__ __forceinline__ void sample_level4(const T* __restrict__ base, int H, int W, int MD,
                                              const float (&lx)[4], const float (&ly)[4], const float (&aw)[4],
                                              float (&col)[VEC]) {
    using R = Raw<T, VEC>;
    constexpr int G = R::GROUP;
    const int row_stride = W * MD;            // intra-level element offsets fit 32 bits (one image's map)
#pragma unroll
    for (int p0 = 0; p0 < 4; p0 += G) {
        typename R::raw_t d[G][4];
        float wgt[G][4];
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const int p = p0 + q;
            // Round 4: the validity of a corner no longer travels as a lane mask next to its data.  The first form kept `ok[q][c]`
            // (inside && h_low >= 0 && ...: chains of s_and_b64 on 64-bit SGPR masks) alive across the 16 gathers and selected the DATA
            // with them afterwards -- the pattern that lost lanes 48..63 of such masks under multi-process contention in the 16-bit decoder
            // kernel (tools/mask_lint.py counted 22..30 masks read after an s_waitcnt vmcnt in these kernels).  As in
            // msda_fused_quad_bf16_kernel the coordinate is clamped to [-1, size] first (identical inside the map; outside it every factor
            // below becomes zero, which is the reference's `inside` test) and each separable factor depends on ONE unsigned compare that is
            // consumed by the next instruction; an invalid corner has a zero WEIGHT on a clamped (valid, finite) address.  Products of the
            // valid corners are bit-identical to the first form (same expression order: hh * hw, ...).
            // CONTRACT that follows (include/dtlr_hip.h, dtlr_msda_forward): `value` must be FINITE.  A corner outside the map multiplies the
            // clamped BORDER pixel by a zero weight, where the reference does not read at all (cuh:49-70): an inf / NaN stored in a border
            // pixel therefore reaches outputs the reference would keep finite (0 * inf = NaN).  Every producer of `value` in this engine
            // is a projection of a normalised stream; the fp16 engine saturates instead of overflowing (FP16_OVFL / the staging clamps).
            const float h_im = __builtin_amdgcn_fmed3f(ly[p] * (float)H - 0.5f, -1.f, (float)H);      // NaN -> -1: zero weights
            const float w_im = __builtin_amdgcn_fmed3f(lx[p] * (float)W - 0.5f, -1.f, (float)W);
            const float hf = floorf(h_im), wf = floorf(w_im);
            const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
            const int h_low = (int)hf, w_low = (int)wf;
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float wy0 = (unsigned)h_low < (unsigned)H ? hh : 0.f, wy1 = (unsigned)h_high < (unsigned)H ? lh : 0.f;
            const float wx0 = (unsigned)w_low < (unsigned)W ? hw : 0.f, wx1 = (unsigned)w_high < (unsigned)W ? lw : 0.f;
            // clamp BOTH ways: h_low ranges over [-1, H], h_high over [0, H + 1]: data unused there, the address must stay in the map
            const int h0 = min(max(h_low, 0), H - 1), h1 = max(min(h_high, H - 1), 0), w0 = min(max(w_low, 0), W - 1), w1 = max(min(w_high, W - 1), 0);
            const int r0 = h0 * row_stride, r1 = h1 * row_stride, c0 = w0 * MD, c1 = w1 * MD;
            d[q][0] = R::ld(base + (r0 + c0));
            d[q][1] = R::ld(base + (r0 + c1));
            d[q][2] = R::ld(base + (r1 + c0));
            d[q][3] = R::ld(base + (r1 + c1));
            wgt[q][0] = wy0 * wx0; wgt[q][1] = wy0 * wx1; wgt[q][2] = wy1 * wx0; wgt[q][3] = wy1 * wx1;
        }
#pragma unroll
        for (int q = 0; q < G; ++q) {
            float v1[VEC], v2[VEC], v3[VEC], v4[VEC];
            R::unpack(d[q][0], true, v1);
            R::unpack(d[q][1], true, v2);
            R::unpack(d[q][2], true, v3);
            R::unpack(d[q][3], true, v4);
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                col[i] += (wgt[q][0] * v1[i] + wgt[q][1] * v2[i] + wgt[q][2] * v3[i] + wgt[q][3] * v4[i]) * aw[p0 + q];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// XCD-aware work placement: the dispatcher puts workgroup b on XCD b % 8 and each XCD has a private
// 4 MiB L2.  The gather re-reads an image's value map (2.8 MB bf16 / 5.6 MB fp32) ~18x, so all
// workgroups of image i are sent to XCD i % 8: that L2 then holds one image's map instead of slices
// of every image in flight.  Used when a workgroup never straddles two images and N % 8 == 0;
// otherwise the identity map.  (Speed only -- results do not depend on placement.)
__device__ __forceinline__ long xcd_block_map(long bid, int bpi /*blocks per image*/, int N) {
    if (bpi <= 0 || (N & 7)) return bid;
    const long xcd = bid & 7, idx = bid >> 3;
    const long img = (idx / bpi) * 8 + xcd, blk = idx % bpi;
    return img * bpi + blk;
}

// T: storage type of value/out; LT: type of loc/attn; VEC channels per thread.
template <typename T, typename LT, typename A, int VEC>
__global__ __launch_bounds__(256) void msda_fwd_kernel(
    const T* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const LT* __restrict__ loc, const LT* __restrict__ attn,
    int S, int M, int D, int L, int Lq, int P, T* __restrict__ out, long total)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= total) return;
    const int cpv = D / VEC;                       // lanes per (b,q,m)
    const int c0 = (int)(tid % cpv) * VEC;
    const long si = tid / cpv;                     // ((b*Lq)+q)*M + m
    const int m = (int)(si % M);
    const long bq = si / M;
    const int b = (int)(bq / Lq);
    const int MD = M * D;
    const LT* lp = loc + si * (long)(L * P) * 2;
    const LT* ap = attn + si * (long)(L * P);
    const T* vb = value + (long)b * S * MD + m * D + c0;
    A col[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) col[i] = 0;
    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const T* base = vb + (long)lsi[l] * MD;
        for (int p = 0; p < P; ++p) {
            sample_point<T, A, VEC>(base, H, W, MD, (A)lp[0], (A)lp[1], (A)ap[0], col);
            lp += 2; ap += 1;
        }
    }
    Vec<T, VEC>::store(out + si * D + c0, col);
}

// Hot shape: L=4 levels x P=4 points, loc/attn fp32.  Fully unrolled; the 32 loc floats and the 16
// attn floats of this (b,q,m) come in as 8 + 4 16-byte loads (all lanes of a head read the same
// addresses: one request each) issued before any gather so their latency overlaps.
template <typename T, int VEC>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 3) void msda_fwd_l4p4_kernel(
    const T* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ attn,
    int S, int M, int D, int Lq, T* __restrict__ out, long total, int bpi, int nimg)
{
    const long tid = xcd_block_map(blockIdx.x, bpi, nimg) * blockDim.x + threadIdx.x;
    if (tid >= total) return;
    const int cpv = D / VEC;
    const int c0 = (int)(tid % cpv) * VEC;
    const long si = tid / cpv;
    const int m = (int)(si % M);
    const int b = (int)((si / M) / Lq);
    const int MD = M * D;
    const float4* lp = reinterpret_cast<const float4*>(loc + si * 32);
    const float4* ap = reinterpret_cast<const float4*>(attn + si * 16);
    float4 lv[8], av[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) lv[i] = lp[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) av[i] = ap[i];
    int Hs[4], Ws[4]; long st[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) { Hs[l] = (int)shapes[2 * l]; Ws[l] = (int)shapes[2 * l + 1]; st[l] = (long)lsi[l]; }
    const T* vb = value + (long)b * S * MD + m * D + c0;
    float col[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) col[i] = 0.f;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const T* base = vb + st[l] * MD;
        const float a[4] = {av[l].x, av[l].y, av[l].z, av[l].w};
        const float lx[4] = {lv[2 * l].x, lv[2 * l].z, lv[2 * l + 1].x, lv[2 * l + 1].z};
        const float ly[4] = {lv[2 * l].y, lv[2 * l].w, lv[2 * l + 1].y, lv[2 * l + 1].w};
        sample_level4<T, VEC>(base, Hs[l], Ws[l], MD, lx, ly, a, col);
    }
    Vec<T, VEC>::store(out + si * D + c0, col);
}


// ---------------------------------------------------------------------------------------------
// Fused front end of MSDeformAttn.forward (ops/modules/ms_deform_attn.py:97-124) for L=4, P=4:
// reads the raw projection row ow[b,q,:] = [offsets (M,L,P,2) | logits (M,L*P)] and the reference
// points, and does softmax(16) + sampling-location arithmetic in registers, so `sampling_locations`
// and `attention_weights` (1.5 KB per query, written then re-read by the reference) never exist in
// HBM.  REFD = 2: loc = ref + off / (W_l, H_l) (encoder);  REFD = 4: loc = ref_xy + off / P * ref_wh
// * 0.5 (decoder) -- same operation order as the reference so fp32 results agree to rounding.
// ---------------------------------------------------------------------------------------------
template <typename OT> __device__ __forceinline__ void load_row16(const OT* p, float (&v)[16]);
template <> __device__ __forceinline__ void load_row16<float>(const float* p, float (&v)[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 t = reinterpret_cast<const float4*>(p)[i];
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w; }
}
template <> __device__ __forceinline__ void load_row16<uint16_t>(const uint16_t* p, float (&v)[16]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { const uint4 t = reinterpret_cast<const uint4*>(p)[i];
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[8 * i + 2 * j] = h16_lo(w[j]); v[8 * i + 2 * j + 1] = h16_hi(w[j]); } }
}

template <typename T, typename OT, int VEC, int REFD>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 3) void msda_fused_l4p4_kernel(
    const T* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const OT* __restrict__ ow, const float* __restrict__ ref,
    int S, int M, int D, int Lq, T* __restrict__ out, long total, int bpi, int nimg, int vstride)
{
    const long tid = xcd_block_map(blockIdx.x, bpi, nimg) * blockDim.x + threadIdx.x;
    if (tid >= total) return;
    const int cpv = D / VEC;
    const int c0 = (int)(tid % cpv) * VEC;
    const long si = tid / cpv;               // (b*Lq + q)*M + m
    const int m = (int)(si % M);
    const long bq = si / M;
    const int b = (int)(bq / Lq);
    (void)0;
    const OT* row = ow + bq * (long)(M * 48);
    float off[32], lg[16];
    load_row16<OT>(row + m * 32, *reinterpret_cast<float (*)[16]>(&off[0]));
    load_row16<OT>(row + m * 32 + 16, *reinterpret_cast<float (*)[16]>(&off[16]));
    load_row16<OT>(row + M * 32 + m * 16, lg);
    float rf[4 * REFD];
    const float4* rp = reinterpret_cast<const float4*>(ref + bq * (4 * REFD));
#pragma unroll
    for (int i = 0; i < REFD; ++i) { const float4 t = rp[i]; rf[4 * i] = t.x; rf[4 * i + 1] = t.y; rf[4 * i + 2] = t.z; rf[4 * i + 3] = t.w; }
    // softmax over the 16 (level, point) logits of this head (ms_deform_attn.py:99-100)
    float mx = lg[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, lg[i]);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { lg[i] = expf(lg[i] - mx); sum += lg[i]; }
    const float inv = 1.0f / sum;
    const T* vb = value + (long)b * S * vstride + m * D + c0;      // vstride: elements between spatial positions (>= M*D)
    float col[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) col[i] = 0.f;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const T* base = vb + (long)lsi[l] * vstride;
        float lx[4], ly[4], aw4[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float ox = off[(l * 4 + p) * 2], oy = off[(l * 4 + p) * 2 + 1];
            if (REFD == 2) {
                lx[p] = rf[2 * l] + ox / (float)W;
                ly[p] = rf[2 * l + 1] + oy / (float)H;
            } else {
                lx[p] = rf[4 * l] + ox / 4.0f * rf[4 * l + 2] * 0.5f;
                ly[p] = rf[4 * l + 1] + oy / 4.0f * rf[4 * l + 3] * 0.5f;
            }
            aw4[p] = lg[l * 4 + p] * inv;
        }
        sample_level4<T, VEC>(base, H, W, vstride, lx, ly, aw4, col);
    }
    Vec<T, VEC>::store(out + si * D + c0, col);
}

// ---- bf16, D = 32: quad = (query, head); lane p owns channel piece p (8 channels) AND does the geometry of level p ---
// The kernel above repeats the 16-point geometry + 16-logit softmax in all 8 lanes of a head (2/3 of its VALU work).
// Here a quad shares it: lane p computes the 4 points of level p (corner offsets + bilinear x attention weights), and
// every lane gets the other levels' 32 values by quad-broadcast DPP.  Each corner is then ONE 16-byte load per lane and
// the four lanes of a quad read the 64 contiguous bytes of a pixel row (coalesced -- a first attempt that gave each lane a
// whole level of its own made every load touch a private cache line and was 30% slower than the original).
template <int CTRL> __device__ __forceinline__ float quad_bcast_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ int quad_bcast_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ void unpack8_bf16(const uint4& t, float (&v)[8]) {
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = h16_lo(w[i]); v[2 * i + 1] = h16_hi(w[i]); }
}
template <typename OT> __device__ __forceinline__ void load8f(const OT* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8f<float>(const float* p, float (&v)[8]) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8f<uint16_t>(const uint16_t* p, float (&v)[8]) { unpack8_bf16(*reinterpret_cast<const uint4*>(p), v); }
template <typename OT> __device__ __forceinline__ void load4f(const OT* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4f<float>(const float* p, float (&v)[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <> __device__ __forceinline__ void load4f<uint16_t>(const uint16_t* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = h16_lo(t.x); v[1] = h16_hi(t.x); v[2] = h16_lo(t.y); v[3] = h16_hi(t.y);
}

template <int L_, typename OT, int REFD>
__device__ __forceinline__ void quad_level(const uint16_t* __restrict__ lbase, const int (&og)[4][4], const float (&kg)[4][4], float (&acc)[8])
{
    constexpr int CTRL = L_ * 0x55;                          // quad_perm [L,L,L,L]
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        uint4 d[4];
        float k[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int o = quad_bcast_i<CTRL>(og[pt][c]);
            k[c] = quad_bcast_f<CTRL>(kg[pt][c]);
            d[c] = *reinterpret_cast<const uint4*>(lbase + o);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v[8];
            unpack8_bf16(d[c], v);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += k[c] * v[i];
        }
    }
}

template <typename OT, int REFD>
__global__ __launch_bounds__(128, 2) void msda_fused_quad_bf16_kernel(
    const uint16_t* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const OT* __restrict__ ow, const float* __restrict__ ref,
    int S, int M, int Lq, uint16_t* __restrict__ out, long total, int bpi, int nimg, int vstride)
{
    const long tid = xcd_block_map(blockIdx.x, bpi, nimg) * blockDim.x + threadIdx.x;
    if (tid >= total) return;                       // total is a multiple of 4: quads are never split
    const int p = (int)(tid & 3);
    const long si = tid >> 2;                       // (b*Lq + q)*M + m
    const int m = (int)(si % M);
    const long bq = si / M;
    const int b = (int)(bq / Lq);
    const int Hl = (int)(p == 0 ? shapes[0] : p == 1 ? shapes[2] : p == 2 ? shapes[4] : shapes[6]);
    const int Wl = (int)(p == 0 ? shapes[1] : p == 1 ? shapes[3] : p == 2 ? shapes[5] : shapes[7]);
    const float fH = (float)Hl, fW = (float)Wl;
    const OT* row = ow + bq * (long)(M * 48);
    float off[8], lg[4];
    load8f<OT>(row + m * 32 + p * 8, off);
    load4f<OT>(row + M * 32 + m * 16 + p * 4, lg);
    float rf[4];
    if (REFD == 2) { const float2 t = *reinterpret_cast<const float2*>(ref + bq * 8 + 2 * p); rf[0] = t.x; rf[1] = t.y; rf[2] = rf[3] = 0.f; }
    else { const float4 t = *reinterpret_cast<const float4*>(ref + bq * 16 + 4 * p); rf[0] = t.x; rf[1] = t.y; rf[2] = t.z; rf[3] = t.w; }
    // softmax over the quad's 16 logits (ms_deform_attn.py:99-100)
    float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
    mx = fmaxf(mx, quad_bcast_f<0xB1>(mx));
    mx = fmaxf(mx, quad_bcast_f<0x4E>(mx));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { lg[i] = __expf(lg[i] - mx); sum += lg[i]; }
    sum += quad_bcast_f<0xB1>(sum);
    sum += quad_bcast_f<0x4E>(sum);
    const float inv = 1.0f / sum;
    // geometry of level p: element offsets of the 4 corners relative to the level base, weights = bilinear x attention
    int og[4][4];
    float kg[4][4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        float lx, ly;
        if (REFD == 2) { lx = rf[0] + off[2 * pt] / fW; ly = rf[1] + off[2 * pt + 1] / fH; }
        else { lx = rf[0] + off[2 * pt] / 4.0f * rf[2] * 0.5f; ly = rf[1] + off[2 * pt + 1] / 4.0f * rf[3] * 0.5f; }
        // Bilinear corners and their validity, one UNSIGNED compare per select (cuh:33-84: a corner counts iff it is inside the map and the
        // point satisfies -1 < h_im < H, -1 < w_im < W).  The coordinate is first clamped to [-1, size]: inside that range nothing
        // changes (same floor, same fractions -- bit-identical weights); at or beyond -1 the clamped point sits ON row/column -1 (its
        // only in-map neighbour, 0, gets the fraction 0) and at or beyond `size` both neighbours are outside, so the reference's
        // `inside` test needs no separate lane mask.  Each weight then depends on ONE comparison (v_cmp -> VCC -> v_cndmask).  The first
        // version combined five comparisons per corner with s_and_b64 into lane masks that lived in SGPR pairs across the whole
        // kernel; under GPU contention (two processes on one device) lanes 48..63 of scattered waves then dropped corner terms --
        // DESIGN.md section 6 -- and this form never has.
        const float h_im = fminf(fmaxf(ly * fH - 0.5f, -1.f), fH), w_im = fminf(fmaxf(lx * fW - 0.5f, -1.f), fW);
        const float hf = floorf(h_im), wf = floorf(w_im);
        const int h_low = (int)hf, w_low = (int)wf;
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
        const float wy0 = (unsigned)h_low < (unsigned)Hl ? hh : 0.f, wy1 = (unsigned)h_high < (unsigned)Hl ? lh : 0.f;
        const float wx0 = (unsigned)w_low < (unsigned)Wl ? hw : 0.f, wx1 = (unsigned)w_high < (unsigned)Wl ? lw : 0.f;
        const int h0 = min(max(h_low, 0), Hl - 1), h1 = max(min(h_high, Hl - 1), 0);
        const int w0 = min(max(w_low, 0), Wl - 1), w1c = max(min(w_high, Wl - 1), 0);
        const float a = lg[pt] * inv;
        kg[pt][0] = wy0 * wx0 * a;  kg[pt][1] = wy0 * wx1 * a;
        kg[pt][2] = wy1 * wx0 * a;  kg[pt][3] = wy1 * wx1 * a;
        og[pt][0] = (h0 * Wl + w0) * vstride;  og[pt][1] = (h0 * Wl + w1c) * vstride;
        og[pt][2] = (h1 * Wl + w0) * vstride;  og[pt][3] = (h1 * Wl + w1c) * vstride;
    }
    const uint16_t* vb = value + (long)b * S * vstride + m * 32 + p * 8;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    quad_level<0, OT, REFD>(vb + lsi[0] * vstride, og, kg, acc);
    __builtin_amdgcn_sched_barrier(0);
    quad_level<1, OT, REFD>(vb + lsi[1] * vstride, og, kg, acc);
    __builtin_amdgcn_sched_barrier(0);
    quad_level<2, OT, REFD>(vb + lsi[2] * vstride, og, kg, acc);
    __builtin_amdgcn_sched_barrier(0);
    quad_level<3, OT, REFD>(vb + lsi[3] * vstride, og, kg, acc);
    *reinterpret_cast<uint4*>(out + si * 32 + p * 8) =
        make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7]));
}

template <typename OT>
static int launch_fused_quad(const void* value, const int64_t* shapes, const int64_t* lsi, const void* ow, const float* ref,
                             int ref_dim, int N, int S, int M, int Lq, void* out, hipStream_t st, int vstride) {
    const long total = (long)N * Lq * M * 4;
    const int block = 128;
    const long grid = (total + block - 1) / block;
    if (grid > 0x7fffffffL) return DTLR_ESHAPE;
    const long per_img = (long)Lq * M * 4;
    const int bpi = (per_img % block == 0 && N % 8 == 0) ? (int)(per_img / block) : 0;
    if (ref_dim == 2)
        hipLaunchKernelGGL((msda_fused_quad_bf16_kernel<OT, 2>), dim3((unsigned)grid), dim3(block), 0, st,
                           (const uint16_t*)value, shapes, lsi, (const OT*)ow, ref, S, M, Lq, (uint16_t*)out, total, bpi, N, vstride);
    else
        hipLaunchKernelGGL((msda_fused_quad_bf16_kernel<OT, 4>), dim3((unsigned)grid), dim3(block), 0, st,
                           (const uint16_t*)value, shapes, lsi, (const OT*)ow, ref, S, M, Lq, (uint16_t*)out, total, bpi, N, vstride);
    return check_launch();
}

template <typename T, typename OT, int VEC>
static int launch_fused(const void* value, const int64_t* shapes, const int64_t* lsi, const void* ow, const float* ref,
                        int ref_dim, int N, int S, int M, int D, int Lq, void* out, hipStream_t st, int vstride) {
    const long total = (long)N * Lq * M * (D / VEC);
    const int block = 256;
    const long grid = (total + block - 1) / block;
    if (grid > 0x7fffffffL) return DTLR_ESHAPE;
    const long per_img = (long)Lq * M * (D / VEC);
    const int bpi = (per_img % block == 0 && N % 8 == 0) ? (int)(per_img / block) : 0;
    if (ref_dim == 2)
        hipLaunchKernelGGL((msda_fused_l4p4_kernel<T, OT, VEC, 2>), dim3((unsigned)grid), dim3(block), 0, st,
                           (const T*)value, shapes, lsi, (const OT*)ow, ref, S, M, D, Lq, (T*)out, total, bpi, N, vstride);
    else
        hipLaunchKernelGGL((msda_fused_l4p4_kernel<T, OT, VEC, 4>), dim3((unsigned)grid), dim3(block), 0, st,
                           (const T*)value, shapes, lsi, (const OT*)ow, ref, S, M, D, Lq, (T*)out, total, bpi, N, vstride);
    return check_launch();
}

template <typename T, typename LT, typename A, int VEC>
static int launch_generic(const void* value, const int64_t* shapes, const int64_t* lsi, const void* loc, const void* attn,
                          int N, int S, int M, int D, int L, int Lq, int P, void* out, hipStream_t st) {
    const long total = (long)N * Lq * M * (D / VEC);
    const int block = 256;
    const long grid = (total + block - 1) / block;
    if (grid > 0x7fffffffL) return DTLR_ESHAPE;
    hipLaunchKernelGGL((msda_fwd_kernel<T, LT, A, VEC>), dim3((unsigned)grid), dim3(block), 0, st,
                       (const T*)value, shapes, lsi, (const LT*)loc, (const LT*)attn, S, M, D, L, Lq, P, (T*)out, total);
    return check_launch();
}

template <typename T, int VEC>
static int launch_l4p4(const void* value, const int64_t* shapes, const int64_t* lsi, const void* loc, const void* attn,
                       int N, int S, int M, int D, int Lq, void* out, hipStream_t st) {
    const long total = (long)N * Lq * M * (D / VEC);
    const int block = 256;
    const long grid = (total + block - 1) / block;
    if (grid > 0x7fffffffL) return DTLR_ESHAPE;
    const long per_img = (long)Lq * M * (D / VEC);
    const int bpi = (per_img % block == 0 && N % 8 == 0) ? (int)(per_img / block) : 0;
    hipLaunchKernelGGL((msda_fwd_l4p4_kernel<T, VEC>), dim3((unsigned)grid), dim3(block), 0, st,
                       (const T*)value, shapes, lsi, (const float*)loc, (const float*)attn, S, M, D, Lq, (T*)out, total, bpi, N);
    return check_launch();
}

}  // namespace dtlr

using namespace dtlr;

extern "C" int dtlr_msda_forward(const void* value, const int64_t* shapes, const int64_t* lsi,
                                 const void* loc, const void* attn,
                                 int N, int S, int M, int D, int L, int Lq, int P,
                                 int dtype, void* out, void* stream)
{
    clear_stale_error();
    if (!value || !shapes || !lsi || !loc || !attn || !out) return DTLR_EINVAL;
    if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0) return DTLR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool hot = (L == 4 && P == 4);
    switch (dtype) {
    case DTLR_F32:
        if (hot && D % 4 == 0) return launch_l4p4<float, 4>(value, shapes, lsi, loc, attn, N, S, M, D, Lq, out, st);
        if (D % 4 == 0) return launch_generic<float, float, float, 4>(value, shapes, lsi, loc, attn, N, S, M, D, L, Lq, P, out, st);
        if (D % 2 == 0) return launch_generic<float, float, float, 2>(value, shapes, lsi, loc, attn, N, S, M, D, L, Lq, P, out, st);
        return launch_generic<float, float, float, 1>(value, shapes, lsi, loc, attn, N, S, M, D, L, Lq, P, out, st);
    case DTLR_F64:
        if (D % 2 == 0) return launch_generic<double, double, double, 2>(value, shapes, lsi, loc, attn, N, S, M, D, L, Lq, P, out, st);
        return launch_generic<double, double, double, 1>(value, shapes, lsi, loc, attn, N, S, M, D, L, Lq, P, out, st);
    case DTLR_H16:
        if (hot && D % 4 == 0) return launch_l4p4<uint16_t, 4>(value, shapes, lsi, loc, attn, N, S, M, D, Lq, out, st);
        if (D % 8 == 0) return launch_generic<uint16_t, float, float, 8>(value, shapes, lsi, loc, attn, N, S, M, D, L, Lq, P, out, st);
        return launch_generic<uint16_t, float, float, 1>(value, shapes, lsi, loc, attn, N, S, M, D, L, Lq, P, out, st);
    default:
        return DTLR_EDTYPE;
    }
}

extern "C" int dtlr_msda_fused_forward(const void* value, const int64_t* shapes, const int64_t* lsi,
                                       const void* ow, const float* ref, int ref_dim,
                                       int N, int S, int M, int D, int L, int Lq, int P,
                                       int dtype, int ow_dtype, void* out, void* stream)
{
    return dtlr_msda_fused_forward_strided(value, 0, shapes, lsi, ow, ref, ref_dim, N, S, M, D, L, Lq, P, dtype, ow_dtype, out, stream);
}

extern "C" int dtlr_msda_fused_forward_strided(const void* value, int value_row_stride, const int64_t* shapes, const int64_t* lsi,
                                               const void* ow, const float* ref, int ref_dim,
                                               int N, int S, int M, int D, int L, int Lq, int P,
                                               int dtype, int ow_dtype, void* out, void* stream)
{
    clear_stale_error();
    if (!value || !shapes || !lsi || !ow || !ref || !out) return DTLR_EINVAL;
    if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || Lq <= 0) return DTLR_EINVAL;
    if (L != 4 || P != 4 || (ref_dim != 2 && ref_dim != 4)) return DTLR_ESHAPE;
    const int vstride = value_row_stride > 0 ? value_row_stride : M * D;
    if (vstride < M * D || (vstride & 7)) return DTLR_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DTLR_F32 && D % 4 == 0) {
        if (ow_dtype == DTLR_F32) return launch_fused<float, float, 4>(value, shapes, lsi, ow, ref, ref_dim, N, S, M, D, Lq, out, st, vstride);
        return DTLR_EDTYPE;
    }
    if (dtype == DTLR_H16 && D == 32 && (long)S * vstride < (1L << 31)) {       // the hot configuration
        if (ow_dtype == DTLR_F32) return launch_fused_quad<float>(value, shapes, lsi, ow, ref, ref_dim, N, S, M, Lq, out, st, vstride);
        if (ow_dtype == DTLR_H16) return launch_fused_quad<uint16_t>(value, shapes, lsi, ow, ref, ref_dim, N, S, M, Lq, out, st, vstride);
        return DTLR_EDTYPE;
    }
    if (dtype == DTLR_H16 && D % 4 == 0) {
        if (ow_dtype == DTLR_F32) return launch_fused<uint16_t, float, 4>(value, shapes, lsi, ow, ref, ref_dim, N, S, M, D, Lq, out, st, vstride);
        if (ow_dtype == DTLR_H16) return launch_fused<uint16_t, uint16_t, 4>(value, shapes, lsi, ow, ref, ref_dim, N, S, M, D, Lq, out, st, vstride);
        return DTLR_EDTYPE;
    }
    return (dtype == DTLR_F32 || dtype == DTLR_H16) ? DTLR_ESHAPE : DTLR_EDTYPE;
}

extern "C" const char* dtlr_strerror(int code) {
    switch (code) {
    case DTLR_OK: return "ok";
    case DTLR_EINVAL: return "invalid argument (null pointer or non-positive size)";
    case DTLR_EDTYPE: return "unsupported dtype code";
    case DTLR_ESHAPE: return "shape not supported by the gfx950 kernels";
    case DTLR_ELAUNCH: return "HIP launch/runtime error (see dtlr_last_hip_error)";
    default: return "unknown error";
    }
}
extern "C" int dtlr_last_hip_error(void) {
    clear_stale_error(); return g_last_hip_error; }
extern "C" int dtlr_abi_version(void) {
    clear_stale_error(); return 1; }
