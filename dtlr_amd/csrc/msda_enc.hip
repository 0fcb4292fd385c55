// LDS-staged multi-scale deformable attention for the ENCODER (queries = the pixels of the 4 levels).
//
// Why: the gather form (msda.hip) moves Lq*M*L*P*4 corner rows = 356 MB/line (fp32) through the
// vector L1 at <= 64 B/clk/CU; measured it is L1/TA-bound at ~0.6 ms per call (B=32), 10x above the
// HBM time of the compulsory bytes.  Text lines are wide and short (level 0 is 16 x 256 at
// 128x2048), sampling offsets are a few pixels, so the value rows a block of neighbouring queries
// touches form a narrow column window of each level: stage it in LDS once (coalesced 16-byte
// loads, each byte of `value` fetched ~1.5x instead of ~18x) and serve the 64 corner reads per
// (query, head) from LDS at 256 B/clk/CU.
//
// Decomposition: grid = (x-tiles, heads, images).  A workgroup owns the queries of ALL levels whose
// pixel centre falls in one slab [t, t+1) * TW0 / W_0 of the normalised x axis (exact integer
// partition, any level widths), for ONE head, and holds for that head the full-height windows
//     level l: columns [floor(t*TW0*W_l/W_0) - R, ceil((t+1)*TW0*W_l/W_0) + R)  x  32 channels.
// A sample whose needed columns are not all inside the window takes the global-memory path (same
// arithmetic) -- correctness never depends on the offsets being small; only speed does.
// Front end fused as in msda_fused_l4p4: softmax(16) + location arithmetic from the raw projection row.
// Arithmetic order = the reference's (cuh:33-84,237-299): fp32 results agree to rounding.
#include "dtlr_common.h"
#include <cstdlib>

namespace dtlr {

struct EncLevels { int H[4], W[4], start[4], wmax[4], loff[4]; };   // loff: LDS offset of level l, in pixels

template <typename T> struct ET;
template <> struct ET<float> {
    static constexpr int VEC = 4, CP = 8;              // channels per 16-byte chunk; chunks per 32-channel pixel row
    static __device__ __forceinline__ void unpack(uint4 t, float (&v)[4]) {
        v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w); }
    static __device__ __forceinline__ uint4 pack(const float (&v)[4]) {
        return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])); }
};
template <> struct ET<uint16_t> {
    static constexpr int VEC = 8, CP = 4;
    static __device__ __forceinline__ void unpack(uint4 t, float (&v)[8]) {
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = h16_lo(w[i]); v[2 * i + 1] = h16_hi(w[i]); } }
    static __device__ __forceinline__ uint4 pack(const float (&v)[8]) {
        return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])); }
};

template <typename OT> __device__ __forceinline__ void ld16f(const OT* p, float (&v)[16]);
template <> __device__ __forceinline__ void ld16f<float>(const float* p, float (&v)[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 t = reinterpret_cast<const float4*>(p)[i];
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w; }
}
template <> __device__ __forceinline__ void ld16f<uint16_t>(const uint16_t* p, float (&v)[16]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { const uint4 t = reinterpret_cast<const uint4*>(p)[i];
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[8 * i + 2 * j] = h16_lo(w[j]); v[8 * i + 2 * j + 1] = h16_hi(w[j]); } }
}

// number of columns j of a W_l-wide level whose centre (j+0.5)/W_l lies left of tile boundary t*TW0/W_0:
// exact integers: (2j+1)*W_0 < 2*t*TW0*W_l
__device__ __forceinline__ int cols_left_of(int t, int TW0, int W0, int Wl) {
    const long num = 2L * t * TW0 * Wl - W0;                  // j < num / (2*W0)
    if (num <= 0) return 0;
    const long c = (num + 2L * W0 - 1) / (2L * W0);            // ceil
    return (int)(c < Wl ? c : Wl);
}

// The bf16 kernel keeps its LDS windows in FP16: every bf16 value in fp16's range converts exactly (8 significant bits into
// 11), and the query phase can then feed the halves of a register straight into v_fma_mix_f32 (fp32 weight x fp16 value +
// fp32 accumulator, one instruction per channel) instead of unpacking bf16 with a shift/mask per channel first -- the
// unpack was 512 of the ~1100 VALU instructions per lane.  Each staged element is converted once and read ~18 times.
// (|x| >= 65520 would become inf and |x| < 6e-8 zero; the sampled tensor is value_proj(LayerNorm output): O(1).)
template <typename T> __device__ __forceinline__ uint4 stage_convert(const uint4& d);
template <> __device__ __forceinline__ uint4 stage_convert<float>(const uint4& d) { return d; }
#ifdef DTLR_HALF_IS_F16
template <> __device__ __forceinline__ uint4 stage_convert<uint16_t>(const uint4& d) { return d; }      // the fp16 library's values ARE fp16
#else
template <> __device__ __forceinline__ uint4 stage_convert<uint16_t>(const uint4& d) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const uint32_t w[4] = {d.x, d.y, d.z, d.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // saturate at fp16's largest finite value: a bf16 value beyond it (never seen: the sampled tensor is value_proj of a
        // LayerNorm output) must not turn into inf and poison a whole window through 0 * inf
        const f2_t f = {fminf(fmaxf(h16_lo(w[j]), -65504.f), 65504.f), fminf(fmaxf(h16_hi(w[j]), -65504.f), 65504.f)};
        o[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, h2_t));
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
#endif
// the same conversion for a kernel that runs with MODE.FP16_OVFL = 1: v_cvt_pk_f16_f32 itself saturates at +-65504
template <typename T> __device__ __forceinline__ uint4 stage_convert_ovfl(const uint4& d);
#ifdef DTLR_HALF_IS_F16
template <> __device__ __forceinline__ uint4 stage_convert_ovfl<uint16_t>(const uint4& d) { return d; }
#else
template <> __device__ __forceinline__ uint4 stage_convert_ovfl<uint16_t>(const uint4& d) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const uint32_t w[4] = {d.x, d.y, d.z, d.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f2_t f = {h16_lo(w[j]), h16_hi(w[j])};
        o[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, h2_t));
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
#endif
// acc += w * fp16 half of `data` (plain asm, not volatile: a pure function of its inputs, the compiler schedules it freely)
__device__ __forceinline__ float fma_mix_lo(float w, uint32_t data, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(w), "v"(data), "v"(acc));
    return d;
}
__device__ __forceinline__ float fma_mix_hi(float w, uint32_t data, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(w), "v"(data), "v"(acc));
    return d;
}

// ---- bf16 query phase: one lane per (query, head, LEVEL) ---------------------------------------------
// The four lanes of a quad are the four levels of one (query, head).  Each lane does the geometry of its own
// 4 points only (the first version gave a lane 8 channels of ALL 16 points: every lane of the quad repeated
// the same 16-point geometry + 16-logit softmax, ~45% of the kernel's VALU instructions, and the kernel is
// VALU-bound: ~2000 instr/lane), accumulates all 32 channels of its level, and the quad is combined with
// 24 DPP adds.  LDS reads of a pixel row (64 B = 4 x 16-byte pieces) are issued in the rotated piece order
// (jj + level) & 3 so that the lanes of a quad hit different 16-byte slots (same bank behaviour as the old
// layout); the rotation cancels in the quad reduce-scatter: lane i ends up owning piece i.
template <int CTRL> __device__ __forceinline__ float quad_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <typename OT> __device__ __forceinline__ void ld8f(const OT* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8f<float>(const float* p, float (&v)[8]) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void ld8f<uint16_t>(const uint16_t* p, float (&v)[8]) {
    ET<uint16_t>::unpack(*reinterpret_cast<const uint4*>(p), v);
}
template <typename OT> __device__ __forceinline__ void ld4f(const OT* p, float (&v)[4]);
template <> __device__ __forceinline__ void ld4f<float>(const float* p, float (&v)[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <> __device__ __forceinline__ void ld4f<uint16_t>(const uint16_t* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = h16_lo(t.x); v[1] = h16_hi(t.x); v[2] = h16_lo(t.y); v[3] = h16_hi(t.y);
}

template <typename OT>
__device__ __forceinline__ void enc_queries_bf16(
    const unsigned char* smem, const uint16_t* __restrict__ vimg, const OT* __restrict__ ow, const float* __restrict__ ref,
    uint16_t* __restrict__ out, const EncLevels lv, const int (&qc0)[4], const int (&qn)[4], const int (&wc0)[4],
    const int (&wc1)[4], const int (&qbase)[5], int nq, int S, int M, int m, int b)
{
    const int tid = threadIdx.x, p = tid & 3;
    const int MD = M * 32;
    // this lane's level (loop invariant: the loop stride 256 keeps tid & 3)
    const int Hl = p == 0 ? lv.H[0] : p == 1 ? lv.H[1] : p == 2 ? lv.H[2] : lv.H[3];
    const int Wl = p == 0 ? lv.W[0] : p == 1 ? lv.W[1] : p == 2 ? lv.W[2] : lv.W[3];
    const int startl = p == 0 ? lv.start[0] : p == 1 ? lv.start[1] : p == 2 ? lv.start[2] : lv.start[3];
    const int loffl = p == 0 ? lv.loff[0] : p == 1 ? lv.loff[1] : p == 2 ? lv.loff[2] : lv.loff[3];
    const int wstride = p == 0 ? lv.wmax[0] : p == 1 ? lv.wmax[1] : p == 2 ? lv.wmax[2] : lv.wmax[3];
    const int wc0l = p == 0 ? wc0[0] : p == 1 ? wc0[1] : p == 2 ? wc0[2] : wc0[3];
    const int wc1l = p == 0 ? wc1[0] : p == 1 ? wc1[1] : p == 2 ? wc1[2] : wc1[3];
    const int wwl = wc1l - wc0l;
    const float fH = (float)Hl, fW = (float)Wl;
    const float invH = 1.0f / fH, invW = 1.0f / fW;          // exact for power-of-two maps; else the product differs from the quotient by <= 1 ulp
    const unsigned char* win = smem + (long)loffl * 64;
    const uint16_t* gsrc = vimg + (long)startl * MD;
    int rot[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) rot[jj] = ((jj + p) & 3) * 16;

    for (int it = tid; it < nq * 4; it += 256) {
        const int q = it >> 2;
        int lq = 0;
        if (q >= qbase[1]) lq = 1;
        if (q >= qbase[2]) lq = 2;
        if (q >= qbase[3]) lq = 3;
        int r, nc, c0q, Wq, stq;
        switch (lq) {
        case 0: r = q - qbase[0]; nc = qn[0]; c0q = qc0[0]; Wq = lv.W[0]; stq = lv.start[0]; break;
        case 1: r = q - qbase[1]; nc = qn[1]; c0q = qc0[1]; Wq = lv.W[1]; stq = lv.start[1]; break;
        case 2: r = q - qbase[2]; nc = qn[2]; c0q = qc0[2]; Wq = lv.W[2]; stq = lv.start[2]; break;
        default: r = q - qbase[3]; nc = qn[3]; c0q = qc0[3]; Wq = lv.W[3]; stq = lv.start[3]; break;
        }
        const int qi = r / nc, qj = c0q + r % nc;
        const long bq = (long)b * S + stq + qi * Wq + qj;
        const OT* row = ow + bq * (long)(M * 48);
        float off[8], lg[4];
        ld8f<OT>(row + m * 32 + p * 8, off);
        ld4f<OT>(row + M * 32 + m * 16 + p * 4, lg);
        const float2 rf = *reinterpret_cast<const float2*>(ref + bq * 8 + 2 * p);
        // softmax over the quad's 16 logits
        float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
        mx = fmaxf(mx, quad_dpp<0xB1>(mx));
        mx = fmaxf(mx, quad_dpp<0x4E>(mx));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { lg[i] = __expf(lg[i] - mx); sum += lg[i]; }
        sum += quad_dpp<0xB1>(sum);
        sum += quad_dpp<0x4E>(sum);
        const float inv = 1.0f / sum;

        float acc[4][8];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[jj][i] = 0.f;
#define DTLR_ACCUM(D, WGT)                                                                         \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                         \
            const uint32_t q_[4] = {D[jj].x, D[jj].y, D[jj].z, D[jj].w};                           \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                        \
                acc[jj][2 * i] = fma_mix_lo((WGT), q_[i], acc[jj][2 * i]);                         \
                acc[jj][2 * i + 1] = fma_mix_hi((WGT), q_[i], acc[jj][2 * i + 1]);                 \
            }                                                                                      \
        }
        // one point at a time: its geometry, then its 16 reads (4 corners x 4 pieces, 64 registers in flight), then the
        // accumulate; the scheduling barrier keeps the compiler from hoisting the next points' reads (it spills ~200
        // registers when it does)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const float lx = rf.x + off[2 * pt] * invW;
            const float ly = rf.y + off[2 * pt + 1] * invH;
            const float h_im = ly * fH - 0.5f, w_im = lx * fW - 0.5f;
            const bool inside = h_im > -1.f && w_im > -1.f && h_im < fH && w_im < fW;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_low = (int)fminf(fmaxf(hf, -1.f), fH), w_low = (int)fminf(fmaxf(wf, -1.f), fW);
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
            const bool top = inside && h_low >= 0, bot = inside && h_high <= Hl - 1, left = w_low >= 0, right = w_high <= Wl - 1;
            const int h0 = min(max(h_low, 0), Hl - 1), h1 = max(min(h_high, Hl - 1), 0);
            const int w0 = min(max(w_low, 0), Wl - 1), w1c = max(min(w_high, Wl - 1), 0);
            const bool staged = (w0 >= wc0l) && (w1c < wc1l);
            const float a = lg[pt] * inv;                       // attention weight folded into the bilinear weights
            const float k1 = (top && left) ? hh * hw * a : 0.f, k2 = (top && right) ? hh * lw * a : 0.f;
            const float k3 = (bot && left) ? lh * hw * a : 0.f, k4 = (bot && right) ? lh * lw * a : 0.f;
            if (staged || !inside) {
                // a point outside the map has all-zero weights but must still read initialised LDS (0 * NaN = NaN): clamp
                const int a0 = min(max(w0 - wc0l, 0), wwl - 1), a1 = min(max(w1c - wc0l, 0), wwl - 1);
                const unsigned char* r0 = win + (h0 * wstride) * 64;
                const unsigned char* r1 = win + (h1 * wstride) * 64;
                uint4 d1[4], d2[4], d3[4], d4[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    d1[jj] = *reinterpret_cast<const uint4*>(r0 + a0 * 64 + rot[jj]);
                    d2[jj] = *reinterpret_cast<const uint4*>(r0 + a1 * 64 + rot[jj]);
                    d3[jj] = *reinterpret_cast<const uint4*>(r1 + a0 * 64 + rot[jj]);
                    d4[jj] = *reinterpret_cast<const uint4*>(r1 + a1 * 64 + rot[jj]);
                }
                DTLR_ACCUM(d1, k1) DTLR_ACCUM(d2, k2) DTLR_ACCUM(d3, k3) DTLR_ACCUM(d4, k4)
            } else {                                            // inside the map, outside the staged window: global path
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int pe = rot[jj] >> 1;                // piece offset in elements
                    const uint4 e1 = *reinterpret_cast<const uint4*>(gsrc + (long)(h0 * Wl + w0) * MD + pe);
                    const uint4 e2 = *reinterpret_cast<const uint4*>(gsrc + (long)(h0 * Wl + w1c) * MD + pe);
                    const uint4 e3 = *reinterpret_cast<const uint4*>(gsrc + (long)(h1 * Wl + w0) * MD + pe);
                    const uint4 e4 = *reinterpret_cast<const uint4*>(gsrc + (long)(h1 * Wl + w1c) * MD + pe);
                    float v1[8], v2[8], v3[8], v4[8];
                    ET<uint16_t>::unpack(e1, v1); ET<uint16_t>::unpack(e2, v2); ET<uint16_t>::unpack(e3, v3); ET<uint16_t>::unpack(e4, v4);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[jj][i] += k1 * v1[i] + k2 * v2[i] + k3 * v3[i] + k4 * v4[i];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef DTLR_ACCUM
        // quad reduce-scatter: lane i's acc[jj] is piece (jj + i) & 3 of its level; lane i collects piece i
        float res[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            res[i] = (acc[0][i] + quad_dpp<0x39>(acc[3][i])) + (quad_dpp<0x4E>(acc[2][i]) + quad_dpp<0x93>(acc[1][i]));
        *reinterpret_cast<uint4*>(out + bq * MD + m * 32 + p * 8) = ET<uint16_t>::pack(res);
    }
}


// ---- bf16 query phase, second form: per-level accumulation in PACKED FP16 ------------------------------------------------
// The first form spends 512 of its ~850 VALU instructions per lane on v_fma_mix_f32, one per (corner, channel).  The windows
// already hold fp16 pairs, so v_pk_fma_f16 blends TWO channels of a corner per instruction: 256 instructions, 16 accumulator
// registers instead of 32.  Precision: a lane only accumulates the 16 corner terms of ITS level in fp16 (weights sum to <= 1,
// values O(1): ~2^-11 relative per term); the four levels are combined in fp32 in the quad reduce-scatter, and the result is
// rounded to bf16 (2^-9) anyway.  The next query's projection row / reference point are loaded one iteration ahead (the first
// form started every iteration with a dependent global load at two waves per SIMD), and the lower register count admits
// NT = 512 threads per workgroup (4 waves per SIMD with two workgroups per CU).
typedef _Float16 enc_h2_t __attribute__((ext_vector_type(2)));
typedef float enc_f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_fma_h2(uint32_t w2, uint32_t d, uint32_t acc) {
    uint32_t r;
    asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(r) : "v"(w2), "v"(d), "v"(acc));
    return r;
}
__device__ __forceinline__ uint32_t h2_splat(float k) {
    const enc_f2_t f = {k, k};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, enc_h2_t));
}
template <int CTRL> __device__ __forceinline__ uint32_t quad_dpp_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <typename OT> struct RowRaw;                    // the lane's slice of the projection row, as loaded (conversion deferred)
template <> struct RowRaw<uint16_t> {
    uint4 off; uint2 lg;
    __device__ __forceinline__ void load(const uint16_t* row, int M, int m, int p) {
        off = *reinterpret_cast<const uint4*>(row + m * 32 + p * 8);
        lg = *reinterpret_cast<const uint2*>(row + M * 32 + m * 16 + p * 4);
    }
    __device__ __forceinline__ void get(float (&o)[8], float (&l)[4]) const {
        ET<uint16_t>::unpack(off, o);
        l[0] = h16_lo(lg.x); l[1] = h16_hi(lg.x);
        l[2] = h16_lo(lg.y); l[3] = h16_hi(lg.y);
    }
};
template <> struct RowRaw<float> {
    float4 o0, o1, l4;
    __device__ __forceinline__ void load(const float* row, int M, int m, int p) {
        o0 = reinterpret_cast<const float4*>(row + m * 32 + p * 8)[0];
        o1 = reinterpret_cast<const float4*>(row + m * 32 + p * 8)[1];
        l4 = *reinterpret_cast<const float4*>(row + M * 32 + m * 16 + p * 4);
    }
    __device__ __forceinline__ void get(float (&o)[8], float (&l)[4]) const {
        o[0] = o0.x; o[1] = o0.y; o[2] = o0.z; o[3] = o0.w; o[4] = o1.x; o[5] = o1.y; o[6] = o1.z; o[7] = o1.w;
        l[0] = l4.x; l[1] = l4.y; l[2] = l4.z; l[3] = l4.w;
    }
};

template <typename OT, int NT>
__device__ __forceinline__ void enc_queries_bf16_h(
    const unsigned char* smem, const int* tok, const uint16_t* __restrict__ vimg, const OT* __restrict__ ow, const float* __restrict__ ref,
    uint16_t* __restrict__ out, const EncLevels lv, const int (&qc0)[4], const int (&qn)[4], const int (&wc0)[4],
    const int (&wc1)[4], const int (&qbase)[5], int nq, int S, int M, int m, int b)
{
    const int tid = threadIdx.x, p = tid & 3;
    const int MD = M * 32;
    const int Hl = p == 0 ? lv.H[0] : p == 1 ? lv.H[1] : p == 2 ? lv.H[2] : lv.H[3];
    const int Wl = p == 0 ? lv.W[0] : p == 1 ? lv.W[1] : p == 2 ? lv.W[2] : lv.W[3];
    const int startl = p == 0 ? lv.start[0] : p == 1 ? lv.start[1] : p == 2 ? lv.start[2] : lv.start[3];
    const int loffl = p == 0 ? lv.loff[0] : p == 1 ? lv.loff[1] : p == 2 ? lv.loff[2] : lv.loff[3];
    const int wstride = p == 0 ? lv.wmax[0] : p == 1 ? lv.wmax[1] : p == 2 ? lv.wmax[2] : lv.wmax[3];
    const int wc0l = p == 0 ? wc0[0] : p == 1 ? wc0[1] : p == 2 ? wc0[2] : wc0[3];
    const int wc1l = p == 0 ? wc1[0] : p == 1 ? wc1[1] : p == 2 ? wc1[2] : wc1[3];
    const int wwl = wc1l - wc0l;
    const float fH = (float)Hl, fW = (float)Wl;
    const float invH = 1.0f / fH, invW = 1.0f / fW;
    const unsigned char* win = smem + (long)loffl * 64;
    const uint16_t* gsrc = vimg + (long)startl * MD;
    int rot[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) rot[jj] = ((jj + p) & 3) * 16;

    // token index of work item `it`: a table the workgroup filled while staging (the integer divisions that map a tile-local
    // query number to (level, row, column) cost ~60 VALU instructions per lookup when done in the loop)
    auto token_of = [&](int it) -> long { return (long)b * S + tok[it >> 2]; };
    const int total = nq * 4;
    if (tid >= total) return;
    long bq = token_of(tid);
    RowRaw<OT> cur, nxt;
    float2 rf, rf_n;
    cur.load(ow + bq * (long)(M * 48), M, m, p);
    rf = *reinterpret_cast<const float2*>(ref + bq * 8 + 2 * p);

    for (int it = tid; it < total; it += NT) {
        // issue the next item's loads before this item's arithmetic (clamped to a valid item: the last iteration re-reads its own)
        const int itn = it + NT < total ? it + NT : it;
        const long bqn = token_of(itn);
        nxt.load(ow + bqn * (long)(M * 48), M, m, p);
        rf_n = *reinterpret_cast<const float2*>(ref + bqn * 8 + 2 * p);

        float off[8], lg[4];
        cur.get(off, lg);
        float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
        mx = fmaxf(mx, quad_dpp<0xB1>(mx));
        mx = fmaxf(mx, quad_dpp<0x4E>(mx));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { lg[i] = __expf(lg[i] - mx); sum += lg[i]; }
        sum += quad_dpp<0xB1>(sum);
        sum += quad_dpp<0x4E>(sum);
        const float inv = 1.0f / sum;

        uint32_t acc[4][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[jj][i] = 0u;
#define DTLR_ACC_H(D, W2)                                                                          \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                         \
            acc[jj][0] = pk_fma_h2((W2), D[jj].x, acc[jj][0]);                                     \
            acc[jj][1] = pk_fma_h2((W2), D[jj].y, acc[jj][1]);                                     \
            acc[jj][2] = pk_fma_h2((W2), D[jj].z, acc[jj][2]);                                     \
            acc[jj][3] = pk_fma_h2((W2), D[jj].w, acc[jj][3]);                                     \
        }
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const float lx = rf.x + off[2 * pt] * invW;
            const float ly = rf.y + off[2 * pt + 1] * invH;
            const float h_im = ly * fH - 0.5f, w_im = lx * fW - 0.5f;
            const bool inside = h_im > -1.f && w_im > -1.f && h_im < fH && w_im < fW;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_low = (int)fminf(fmaxf(hf, -1.f), fH), w_low = (int)fminf(fmaxf(wf, -1.f), fW);
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
            const bool top = inside && h_low >= 0, bot = inside && h_high <= Hl - 1, left = w_low >= 0, right = w_high <= Wl - 1;
            const int h0 = min(max(h_low, 0), Hl - 1), h1 = max(min(h_high, Hl - 1), 0);
            const int w0 = min(max(w_low, 0), Wl - 1), w1c = max(min(w_high, Wl - 1), 0);
            const bool staged = (w0 >= wc0l) && (w1c < wc1l);
            const float a = lg[pt] * inv;
            const uint32_t k1 = h2_splat((top && left) ? hh * hw * a : 0.f), k2 = h2_splat((top && right) ? hh * lw * a : 0.f);
            const uint32_t k3 = h2_splat((bot && left) ? lh * hw * a : 0.f), k4 = h2_splat((bot && right) ? lh * lw * a : 0.f);
            // two corners (one map row) at a time: 32 data registers in flight instead of 64, so that the kernel fits 128 VGPRs
            // (four waves per SIMD); the scheduling barriers keep the compiler from hoisting the second row's reads
            if (staged || !inside) {
                const int a0 = min(max(w0 - wc0l, 0), wwl - 1), a1 = min(max(w1c - wc0l, 0), wwl - 1);
                const unsigned char* r0 = win + (h0 * wstride) * 64;
                const unsigned char* r1 = win + (h1 * wstride) * 64;
                {
                    uint4 d1[4], d2[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        d1[jj] = *reinterpret_cast<const uint4*>(r0 + a0 * 64 + rot[jj]);
                        d2[jj] = *reinterpret_cast<const uint4*>(r0 + a1 * 64 + rot[jj]);
                    }
                    DTLR_ACC_H(d1, k1) DTLR_ACC_H(d2, k2)
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    uint4 d3[4], d4[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        d3[jj] = *reinterpret_cast<const uint4*>(r1 + a0 * 64 + rot[jj]);
                        d4[jj] = *reinterpret_cast<const uint4*>(r1 + a1 * 64 + rot[jj]);
                    }
                    DTLR_ACC_H(d3, k3) DTLR_ACC_H(d4, k4)
                }
            } else {                                            // inside the map, outside the staged window: global path
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int pe = rot[jj] >> 1;
                    const uint4 e1 = stage_convert<uint16_t>(*reinterpret_cast<const uint4*>(gsrc + (long)(h0 * Wl + w0) * MD + pe));
                    const uint4 e2 = stage_convert<uint16_t>(*reinterpret_cast<const uint4*>(gsrc + (long)(h0 * Wl + w1c) * MD + pe));
                    const uint4 e3 = stage_convert<uint16_t>(*reinterpret_cast<const uint4*>(gsrc + (long)(h1 * Wl + w0) * MD + pe));
                    const uint4 e4 = stage_convert<uint16_t>(*reinterpret_cast<const uint4*>(gsrc + (long)(h1 * Wl + w1c) * MD + pe));
                    acc[jj][0] = pk_fma_h2(k4, e4.x, pk_fma_h2(k3, e3.x, pk_fma_h2(k2, e2.x, pk_fma_h2(k1, e1.x, acc[jj][0]))));
                    acc[jj][1] = pk_fma_h2(k4, e4.y, pk_fma_h2(k3, e3.y, pk_fma_h2(k2, e2.y, pk_fma_h2(k1, e1.y, acc[jj][1]))));
                    acc[jj][2] = pk_fma_h2(k4, e4.z, pk_fma_h2(k3, e3.z, pk_fma_h2(k2, e2.z, pk_fma_h2(k1, e1.z, acc[jj][2]))));
                    acc[jj][3] = pk_fma_h2(k4, e4.w, pk_fma_h2(k3, e3.w, pk_fma_h2(k2, e2.w, pk_fma_h2(k1, e1.w, acc[jj][3]))));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef DTLR_ACC_H
        // quad reduce-scatter in fp32: lane i's acc[jj] is piece (jj + i) & 3 of its level; lane i collects piece i of all four
        float res[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t t3 = quad_dpp_u<0x39>(acc[3][i]), t2 = quad_dpp_u<0x4E>(acc[2][i]), t1 = quad_dpp_u<0x93>(acc[1][i]);
            res[2 * i] = fma_mix_lo(1.f, t1, fma_mix_lo(1.f, t2, fma_mix_lo(1.f, t3, fma_mix_lo(1.f, acc[0][i], 0.f))));
            res[2 * i + 1] = fma_mix_hi(1.f, t1, fma_mix_hi(1.f, t2, fma_mix_hi(1.f, t3, fma_mix_hi(1.f, acc[0][i], 0.f))));
        }
        *reinterpret_cast<uint4*>(out + bq * MD + m * 32 + p * 8) = ET<uint16_t>::pack(res);
        cur = nxt; rf = rf_n; bq = bqn;
    }
}

// ---- bf16 query phase, third form: the second form with fewer VALU instructions per lane -------------------------------------
// The DEFAULT since round 4 (timed then: same-box A/B of the bench step 9.24 -> 9.12 ms; tests green on hardware).  The
// kernel is VALU-issue-bound (SQ counters: ~96% of its duration), so the instruction count of this loop is its cost model
// (tools/isa_mix.py).  Changes against the second form, none of which touches the data layout:
//   * geometry as in msda_fused_quad_bf16_kernel: the coordinate is clamped to [-1, size] (one v_med3), a corner's validity is ONE
//     unsigned compare per axis, applied to the separable weights -- no `inside` mask, no per-corner compare / s_and chains; a point
//     outside the map gets zero weights on valid (clamped) addresses, and a NaN offset clamps to -1 = zero weights as before;
//   * the softmax normalisation is v_rcp_f32 (1 ulp) instead of the IEEE division sequence (12 instructions);
//   * two corner weights per v_cvt_pk_f16_f32, broadcast into v_pk_fma_f16 through op_sel instead of four splat conversions.
// Results: same tolerance against the oracle as the second form; not bit-identical to it (fp32 product order of the weights).
__device__ __forceinline__ uint32_t pk_fma_h2_bl(uint32_t w2, uint32_t d, uint32_t acc) {      // both halves of d times the LOW half of w2
    uint32_t r;
    asm("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(w2), "v"(d), "v"(acc));
    return r;
}
__device__ __forceinline__ uint32_t pk_fma_h2_bh(uint32_t w2, uint32_t d, uint32_t acc) {      // ... times the HIGH half of w2
    uint32_t r;
    asm("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(w2), "v"(d), "v"(acc));
    return r;
}
__device__ __forceinline__ uint32_t h2_pair(float a, float b) {
    const enc_f2_t f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, enc_h2_t));
}
__device__ __forceinline__ int clamp0_i32(int v, int hi) {                                    // min(max(v, 0), hi), hi >= 0: one v_med3_i32
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "v"(hi));
    return r;
}

template <typename OT, int NT>
__device__ __forceinline__ void enc_queries_bf16_h3(
    const unsigned char* smem, const int* tok, const uint16_t* __restrict__ vimg, const OT* __restrict__ ow, const float* __restrict__ ref,
    uint16_t* __restrict__ out, const EncLevels lv, const int (&wc0)[4], const int (&wc1)[4], int nq, int S, int M, int m, int b)
{
    const int tid = threadIdx.x, p = tid & 3;
    const int MD = M * 32;
    const int Hl = p == 0 ? lv.H[0] : p == 1 ? lv.H[1] : p == 2 ? lv.H[2] : lv.H[3];
    const int Wl = p == 0 ? lv.W[0] : p == 1 ? lv.W[1] : p == 2 ? lv.W[2] : lv.W[3];
    const int startl = p == 0 ? lv.start[0] : p == 1 ? lv.start[1] : p == 2 ? lv.start[2] : lv.start[3];
    const int loffl = p == 0 ? lv.loff[0] : p == 1 ? lv.loff[1] : p == 2 ? lv.loff[2] : lv.loff[3];
    const int wstride = p == 0 ? lv.wmax[0] : p == 1 ? lv.wmax[1] : p == 2 ? lv.wmax[2] : lv.wmax[3];
    const int wc0l = p == 0 ? wc0[0] : p == 1 ? wc0[1] : p == 2 ? wc0[2] : wc0[3];
    const int wc1l = p == 0 ? wc1[0] : p == 1 ? wc1[1] : p == 2 ? wc1[2] : wc1[3];
    const float fH = (float)Hl, fW = (float)Wl;
    const float invH = 1.0f / fH, invW = 1.0f / fW;
    const unsigned char* win = smem + (long)loffl * 64;
    const uint16_t* gsrc = vimg + (long)startl * MD;
    int rot[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) rot[jj] = ((jj + p) & 3) * 16;

    auto token_of = [&](int it) -> long { return (long)b * S + tok[it >> 2]; };
    const int total = nq * 4;
    if (tid >= total) return;
    long bq = token_of(tid);
    RowRaw<OT> cur, nxt;
    float2 rf, rf_n;
    cur.load(ow + bq * (long)(M * 48), M, m, p);
    rf = *reinterpret_cast<const float2*>(ref + bq * 8 + 2 * p);

    for (int it = tid; it < total; it += NT) {
        const int itn = it + NT < total ? it + NT : it;
        const long bqn = token_of(itn);
        nxt.load(ow + bqn * (long)(M * 48), M, m, p);
        rf_n = *reinterpret_cast<const float2*>(ref + bqn * 8 + 2 * p);

        float off[8], lg[4];
        cur.get(off, lg);
        float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
        mx = fmaxf(mx, quad_dpp<0xB1>(mx));
        mx = fmaxf(mx, quad_dpp<0x4E>(mx));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { lg[i] = __expf(lg[i] - mx); sum += lg[i]; }
        sum += quad_dpp<0xB1>(sum);
        sum += quad_dpp<0x4E>(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);

        uint32_t acc[4][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[jj][i] = 0u;
#define DTLR_ACC_H3(FMA, D, W2)                                                                    \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                         \
            acc[jj][0] = FMA((W2), D[jj].x, acc[jj][0]);                                           \
            acc[jj][1] = FMA((W2), D[jj].y, acc[jj][1]);                                           \
            acc[jj][2] = FMA((W2), D[jj].z, acc[jj][2]);                                           \
            acc[jj][3] = FMA((W2), D[jj].w, acc[jj][3]);                                           \
        }
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const float lx = rf.x + off[2 * pt] * invW;
            const float ly = rf.y + off[2 * pt + 1] * invH;
            // v_med3_f32: a NaN operand yields min3 of the others = -1 (zero weights), like fminf(fmaxf(x, -1), size)
            const float h_im = __builtin_amdgcn_fmed3f(ly * fH - 0.5f, -1.f, fH), w_im = __builtin_amdgcn_fmed3f(lx * fW - 0.5f, -1.f, fW);
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_low = (int)hf, w_low = (int)wf;
            const float lh = h_im - hf, lw = w_im - wf;
            const float a = lg[pt] * inv;
            const float wy0 = (unsigned)h_low < (unsigned)Hl ? 1.f - lh : 0.f, wy1 = (unsigned)(h_low + 1) < (unsigned)Hl ? lh : 0.f;
            const float wx0 = (unsigned)w_low < (unsigned)Wl ? (1.f - lw) * a : 0.f, wx1 = (unsigned)(w_low + 1) < (unsigned)Wl ? lw * a : 0.f;
            const uint32_t k12 = h2_pair(wy0 * wx0, wy0 * wx1), k34 = h2_pair(wy1 * wx0, wy1 * wx1);
            const int h0 = clamp0_i32(h_low, Hl - 1), h1 = clamp0_i32(h_low + 1, Hl - 1);
            const int w0 = clamp0_i32(w_low, Wl - 1), w1c = clamp0_i32(w_low + 1, Wl - 1);
            if ((w0 >= wc0l) && (w1c < wc1l)) {
                const unsigned char* r0 = win + (h0 * wstride) * 64;
                const unsigned char* r1 = win + (h1 * wstride) * 64;
                const int a0 = (w0 - wc0l) * 64, a1 = (w1c - wc0l) * 64;
                {
                    uint4 d1[4], d2[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        d1[jj] = *reinterpret_cast<const uint4*>(r0 + a0 + rot[jj]);
                        d2[jj] = *reinterpret_cast<const uint4*>(r0 + a1 + rot[jj]);
                    }
                    DTLR_ACC_H3(pk_fma_h2_bl, d1, k12) DTLR_ACC_H3(pk_fma_h2_bh, d2, k12)
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    uint4 d3[4], d4[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        d3[jj] = *reinterpret_cast<const uint4*>(r1 + a0 + rot[jj]);
                        d4[jj] = *reinterpret_cast<const uint4*>(r1 + a1 + rot[jj]);
                    }
                    DTLR_ACC_H3(pk_fma_h2_bl, d3, k34) DTLR_ACC_H3(pk_fma_h2_bh, d4, k34)
                }
            } else {                                            // outside the staged window: global path (clamped, valid addresses; MODE.FP16_OVFL is set)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int pe = rot[jj] >> 1;
                    const uint4 e1 = stage_convert_ovfl<uint16_t>(*reinterpret_cast<const uint4*>(gsrc + (long)(h0 * Wl + w0) * MD + pe));
                    const uint4 e2 = stage_convert_ovfl<uint16_t>(*reinterpret_cast<const uint4*>(gsrc + (long)(h0 * Wl + w1c) * MD + pe));
                    const uint4 e3 = stage_convert_ovfl<uint16_t>(*reinterpret_cast<const uint4*>(gsrc + (long)(h1 * Wl + w0) * MD + pe));
                    const uint4 e4 = stage_convert_ovfl<uint16_t>(*reinterpret_cast<const uint4*>(gsrc + (long)(h1 * Wl + w1c) * MD + pe));
                    acc[jj][0] = pk_fma_h2_bh(k34, e4.x, pk_fma_h2_bl(k34, e3.x, pk_fma_h2_bh(k12, e2.x, pk_fma_h2_bl(k12, e1.x, acc[jj][0]))));
                    acc[jj][1] = pk_fma_h2_bh(k34, e4.y, pk_fma_h2_bl(k34, e3.y, pk_fma_h2_bh(k12, e2.y, pk_fma_h2_bl(k12, e1.y, acc[jj][1]))));
                    acc[jj][2] = pk_fma_h2_bh(k34, e4.z, pk_fma_h2_bl(k34, e3.z, pk_fma_h2_bh(k12, e2.z, pk_fma_h2_bl(k12, e1.z, acc[jj][2]))));
                    acc[jj][3] = pk_fma_h2_bh(k34, e4.w, pk_fma_h2_bl(k34, e3.w, pk_fma_h2_bh(k12, e2.w, pk_fma_h2_bl(k12, e1.w, acc[jj][3]))));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef DTLR_ACC_H3
        float res[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t t3 = quad_dpp_u<0x39>(acc[3][i]), t2 = quad_dpp_u<0x4E>(acc[2][i]), t1 = quad_dpp_u<0x93>(acc[1][i]);
            res[2 * i] = fma_mix_lo(1.f, t1, fma_mix_lo(1.f, t2, fma_mix_lo(1.f, t3, fma_mix_lo(1.f, acc[0][i], 0.f))));
            res[2 * i + 1] = fma_mix_hi(1.f, t1, fma_mix_hi(1.f, t2, fma_mix_hi(1.f, t3, fma_mix_hi(1.f, acc[0][i], 0.f))));
        }
        *reinterpret_cast<uint4*>(out + bq * MD + m * 32 + p * 8) = ET<uint16_t>::pack(res);
        cur = nxt; rf = rf_n; bq = bqn;
    }
}

// ---- bf16 query phase, fourth form (round 6): the third form with the window reads one half-point AHEAD of the FMAs -------------------
// tools/experiments/valu_rate.hip put numbers on the third form: per lane-iteration ~880 ns of VALU issue per SIMD and ~750 ns of LDS time
// per CU (64 ds_read_b128 at a 1.65x bank-conflict factor) -- 75 us + 64 us per launch back to back, 139 us measured: the kernel is bound by
// the SERIALISATION of the two inside a wave (eight read -> wait -> FMA phases per lane-iteration; four waves per SIMD and 127 of 128 VGPRs
// leave nothing to read ahead into).  Here: 384 threads per workgroup (three waves per SIMD at two workgroups per CU: 168 VGPRs), the
// geometry of all four points first, then the eight half-points (a pixel row pair of one point: 8 reads, 32 packed FMAs) through TWO register
// buffers -- the reads of half-point h + 2 are issued right after the FMAs of half-point h freed their buffer and land under the FMAs of h + 1.
// The loop has no branch: a point whose columns are not all inside the staged window reads a clamped (valid) window address with zero
// weights and is added afterwards through the global path (same arithmetic; those lanes sum their points in a different order: fp16
// rounding-order noise, the oracle tolerance is unchanged).
// MEASURED (profiles/r06_msda_v4_c16.txt, same box, separate processes): 206 us per launch against 142-145 us for the third form, the bench
// step 9.96 against 9.63 ms -- the fourth wave per SIMD hides more than the read-ahead does.  Experiment builds only (DTLR_MSDA_ENC_V=4); it was
// timed, its results were NOT compared with the oracle (no test selects it).
template <typename OT, int NT>
__device__ __forceinline__ void enc_queries_bf16_h4(
    const unsigned char* smem, const int* tok, const uint16_t* __restrict__ vimg, const OT* __restrict__ ow, const float* __restrict__ ref,
    uint16_t* __restrict__ out, const EncLevels lv, const int (&wc0)[4], const int (&wc1)[4], int nq, int S, int M, int m, int b)
{
    const int tid = threadIdx.x, p = tid & 3;
    const int MD = M * 32;
    const int Hl = p == 0 ? lv.H[0] : p == 1 ? lv.H[1] : p == 2 ? lv.H[2] : lv.H[3];
    const int Wl = p == 0 ? lv.W[0] : p == 1 ? lv.W[1] : p == 2 ? lv.W[2] : lv.W[3];
    const int startl = p == 0 ? lv.start[0] : p == 1 ? lv.start[1] : p == 2 ? lv.start[2] : lv.start[3];
    const int loffl = p == 0 ? lv.loff[0] : p == 1 ? lv.loff[1] : p == 2 ? lv.loff[2] : lv.loff[3];
    const int wstride = p == 0 ? lv.wmax[0] : p == 1 ? lv.wmax[1] : p == 2 ? lv.wmax[2] : lv.wmax[3];
    const int wc0l = p == 0 ? wc0[0] : p == 1 ? wc0[1] : p == 2 ? wc0[2] : wc0[3];
    const int wc1l = p == 0 ? wc1[0] : p == 1 ? wc1[1] : p == 2 ? wc1[2] : wc1[3];
    const float fH = (float)Hl, fW = (float)Wl;
    const float invH = 1.0f / fH, invW = 1.0f / fW;
    const unsigned char* win = smem + (long)loffl * 64;
    const uint16_t* gsrc = vimg + (long)startl * MD;
    int rot[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) rot[jj] = ((jj + p) & 3) * 16;

    auto token_of = [&](int it) -> long { return (long)b * S + tok[it >> 2]; };
    const int total = nq * 4;
    if (tid >= total) return;
    long bq = token_of(tid);
    RowRaw<OT> cur, nxt;
    float2 rf, rf_n;
    cur.load(ow + bq * (long)(M * 48), M, m, p);
    rf = *reinterpret_cast<const float2*>(ref + bq * 8 + 2 * p);

    for (int it = tid; it < total; it += NT) {
        const int itn = it + NT < total ? it + NT : it;
        const long bqn = token_of(itn);
        nxt.load(ow + bqn * (long)(M * 48), M, m, p);
        rf_n = *reinterpret_cast<const float2*>(ref + bqn * 8 + 2 * p);

        float off[8], lg[4];
        cur.get(off, lg);
        float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
        mx = fmaxf(mx, quad_dpp<0xB1>(mx));
        mx = fmaxf(mx, quad_dpp<0x4E>(mx));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { lg[i] = __expf(lg[i] - mx); sum += lg[i]; }
        sum += quad_dpp<0xB1>(sum);
        sum += quad_dpp<0x4E>(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);

        // geometry of the four points: packed corner weights (real ones: the global path uses them), window byte offsets of the four corners
        // (clamped into the window for a point outside it), and the mask of such points
        uint32_t k12[4], k34[4];
        int ad[4][4];                                           // [point][row * 2 + column]
        int hw[4][4];                                           // h0, h1, w0, w1c of the far points' global path
        unsigned far = 0u;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const float lx = rf.x + off[2 * pt] * invW;
            const float ly = rf.y + off[2 * pt + 1] * invH;
            const float h_im = __builtin_amdgcn_fmed3f(ly * fH - 0.5f, -1.f, fH), w_im = __builtin_amdgcn_fmed3f(lx * fW - 0.5f, -1.f, fW);
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_low = (int)hf, w_low = (int)wf;
            const float lh = h_im - hf, lw = w_im - wf;
            const float a = lg[pt] * inv;
            const float wy0 = (unsigned)h_low < (unsigned)Hl ? 1.f - lh : 0.f, wy1 = (unsigned)(h_low + 1) < (unsigned)Hl ? lh : 0.f;
            const float wx0 = (unsigned)w_low < (unsigned)Wl ? (1.f - lw) * a : 0.f, wx1 = (unsigned)(w_low + 1) < (unsigned)Wl ? lw * a : 0.f;
            k12[pt] = h2_pair(wy0 * wx0, wy0 * wx1); k34[pt] = h2_pair(wy1 * wx0, wy1 * wx1);
            const int h0 = clamp0_i32(h_low, Hl - 1), h1 = clamp0_i32(h_low + 1, Hl - 1);
            const int w0 = clamp0_i32(w_low, Wl - 1), w1c = clamp0_i32(w_low + 1, Wl - 1);
            hw[pt][0] = h0; hw[pt][1] = h1; hw[pt][2] = w0; hw[pt][3] = w1c;
            const bool inw = (w0 >= wc0l) && (w1c < wc1l);
            if (!inw) far |= 1u << pt;
            const int c0 = inw ? w0 - wc0l : 0, c1 = inw ? w1c - wc0l : 0;
            const int r0 = h0 * wstride, r1 = h1 * wstride;
            ad[pt][0] = (r0 + c0) * 64; ad[pt][1] = (r0 + c1) * 64; ad[pt][2] = (r1 + c0) * 64; ad[pt][3] = (r1 + c1) * 64;
        }

        uint32_t acc[4][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[jj][i] = 0u;
        uint4 bufA[8], bufB[8];
#define DTLR_H4_LOAD(BUF, HP)                                                                      \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                         \
            BUF[jj] = *reinterpret_cast<const uint4*>(win + ad[(HP) >> 1][2 * ((HP) & 1)] + rot[jj]);      \
            BUF[4 + jj] = *reinterpret_cast<const uint4*>(win + ad[(HP) >> 1][2 * ((HP) & 1) + 1] + rot[jj]); \
        }
#define DTLR_H4_FMA(BUF, HP)                                                                       \
        {                                                                                          \
            const uint32_t kreal_ = ((HP) & 1) ? k34[(HP) >> 1] : k12[(HP) >> 1];                  \
            const uint32_t kk_ = (far >> ((HP) >> 1)) & 1u ? 0u : kreal_;                          \
            _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                     \
                acc[jj][0] = pk_fma_h2_bl(kk_, BUF[jj].x, acc[jj][0]);                             \
                acc[jj][1] = pk_fma_h2_bl(kk_, BUF[jj].y, acc[jj][1]);                             \
                acc[jj][2] = pk_fma_h2_bl(kk_, BUF[jj].z, acc[jj][2]);                             \
                acc[jj][3] = pk_fma_h2_bl(kk_, BUF[jj].w, acc[jj][3]);                             \
            }                                                                                      \
            _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                     \
                acc[jj][0] = pk_fma_h2_bh(kk_, BUF[4 + jj].x, acc[jj][0]);                         \
                acc[jj][1] = pk_fma_h2_bh(kk_, BUF[4 + jj].y, acc[jj][1]);                         \
                acc[jj][2] = pk_fma_h2_bh(kk_, BUF[4 + jj].z, acc[jj][2]);                         \
                acc[jj][3] = pk_fma_h2_bh(kk_, BUF[4 + jj].w, acc[jj][3]);                         \
            }                                                                                      \
        }
        DTLR_H4_LOAD(bufA, 0)
        DTLR_H4_LOAD(bufB, 1)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int hp = 0; hp < 8; ++hp) {
            if (hp & 1) { DTLR_H4_FMA(bufB, hp) } else { DTLR_H4_FMA(bufA, hp) }
            if (hp + 2 < 8) { if (hp & 1) { DTLR_H4_LOAD(bufB, hp + 2) } else { DTLR_H4_LOAD(bufA, hp + 2) } }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef DTLR_H4_FMA
#undef DTLR_H4_LOAD
        if (far) {                                              // points outside the staged window: global path (MODE.FP16_OVFL is set)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                if (!((far >> pt) & 1u)) continue;
                const int h0 = hw[pt][0], h1 = hw[pt][1], w0 = hw[pt][2], w1c = hw[pt][3];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int pe = rot[jj] >> 1;
                    const uint4 e1 = stage_convert_ovfl<uint16_t>(*reinterpret_cast<const uint4*>(gsrc + (long)(h0 * Wl + w0) * MD + pe));
                    const uint4 e2 = stage_convert_ovfl<uint16_t>(*reinterpret_cast<const uint4*>(gsrc + (long)(h0 * Wl + w1c) * MD + pe));
                    const uint4 e3 = stage_convert_ovfl<uint16_t>(*reinterpret_cast<const uint4*>(gsrc + (long)(h1 * Wl + w0) * MD + pe));
                    const uint4 e4 = stage_convert_ovfl<uint16_t>(*reinterpret_cast<const uint4*>(gsrc + (long)(h1 * Wl + w1c) * MD + pe));
                    acc[jj][0] = pk_fma_h2_bh(k34[pt], e4.x, pk_fma_h2_bl(k34[pt], e3.x, pk_fma_h2_bh(k12[pt], e2.x, pk_fma_h2_bl(k12[pt], e1.x, acc[jj][0]))));
                    acc[jj][1] = pk_fma_h2_bh(k34[pt], e4.y, pk_fma_h2_bl(k34[pt], e3.y, pk_fma_h2_bh(k12[pt], e2.y, pk_fma_h2_bl(k12[pt], e1.y, acc[jj][1]))));
                    acc[jj][2] = pk_fma_h2_bh(k34[pt], e4.z, pk_fma_h2_bl(k34[pt], e3.z, pk_fma_h2_bh(k12[pt], e2.z, pk_fma_h2_bl(k12[pt], e1.z, acc[jj][2]))));
                    acc[jj][3] = pk_fma_h2_bh(k34[pt], e4.w, pk_fma_h2_bl(k34[pt], e3.w, pk_fma_h2_bh(k12[pt], e2.w, pk_fma_h2_bl(k12[pt], e1.w, acc[jj][3]))));
                }
            }
        }
        float res[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t t3 = quad_dpp_u<0x39>(acc[3][i]), t2 = quad_dpp_u<0x4E>(acc[2][i]), t1 = quad_dpp_u<0x93>(acc[1][i]);
            res[2 * i] = fma_mix_lo(1.f, t1, fma_mix_lo(1.f, t2, fma_mix_lo(1.f, t3, fma_mix_lo(1.f, acc[0][i], 0.f))));
            res[2 * i + 1] = fma_mix_hi(1.f, t1, fma_mix_hi(1.f, t2, fma_mix_hi(1.f, t3, fma_mix_hi(1.f, acc[0][i], 0.f))));
        }
        *reinterpret_cast<uint4*>(out + bq * MD + m * 32 + p * 8) = ET<uint16_t>::pack(res);
        cur = nxt; rf = rf_n; bq = bqn;
    }
}

// ---- fp32 query phase (round 4): one lane per (query, head, LEVEL), fp32 windows, packed-fp32 accumulation ----------------------
// The first fp32 form (8 lanes per (query, head), each lane 4 channels of ALL 16 points) repeated the 16-point geometry and the
// 16-logit softmax in every lane: ~7000 VALU lane-instructions per (query, head), 1.17 ms per encoder call at B = 32 -- the largest
// kernel of the split-fp32 engine (7.0 of 35 ms per step).  This is the level-per-lane decomposition of the 16-bit forms on fp32
// windows: a lane does the geometry of ITS level's 4 points, reads whole 128-byte pixel rows (8 x ds_read_b128, pieces in the
// lane-rotated order (jj + 2 level) & 7 so the quad's reads spread over the banks), accumulates the 32 channels of its level with
// v_pk_fma_f32 (two channels per instruction, nothing to unpack), and the quad is combined by the same DPP reduce-scatter (lane i ends
// up with channels 8i..8i+7 = pieces 2i, 2i+1).  Geometry as in msda_fused_quad / the third 16-bit form: coordinate clamped to
// [-1, size], ONE unsigned compare per axis and corner, no long-lived lane masks.  Differences from the reference's evaluation order:
// the attention weight is folded into the bilinear weights and the levels are summed pairwise in the quad -- fp32 rounding-order
// noise (~1e-7 relative), below the GEMMs' own.  ~460 VALU per lane-iteration; LDS-read-bound (2 KB per lane-iteration).
typedef float encf2_t __attribute__((ext_vector_type(2)));
template <typename OT, int NT>
__device__ __forceinline__ void enc_queries_f32_lvl(
    const unsigned char* smem, const int* tok, const float* __restrict__ vimg, const OT* __restrict__ ow, const float* __restrict__ ref,
    float* __restrict__ out, const EncLevels lv, const int (&wc0)[4], const int (&wc1)[4], int nq, int S, int M, int m, int b)
{
    const int tid = threadIdx.x, p = tid & 3;
    const int MD = M * 32;
    const int Hl = p == 0 ? lv.H[0] : p == 1 ? lv.H[1] : p == 2 ? lv.H[2] : lv.H[3];
    const int Wl = p == 0 ? lv.W[0] : p == 1 ? lv.W[1] : p == 2 ? lv.W[2] : lv.W[3];
    const int startl = p == 0 ? lv.start[0] : p == 1 ? lv.start[1] : p == 2 ? lv.start[2] : lv.start[3];
    const int loffl = p == 0 ? lv.loff[0] : p == 1 ? lv.loff[1] : p == 2 ? lv.loff[2] : lv.loff[3];
    const int wstride = p == 0 ? lv.wmax[0] : p == 1 ? lv.wmax[1] : p == 2 ? lv.wmax[2] : lv.wmax[3];
    const int wc0l = p == 0 ? wc0[0] : p == 1 ? wc0[1] : p == 2 ? wc0[2] : wc0[3];
    const int wc1l = p == 0 ? wc1[0] : p == 1 ? wc1[1] : p == 2 ? wc1[2] : wc1[3];
    const float fH = (float)Hl, fW = (float)Wl;
    const unsigned char* win = smem + (long)loffl * 128;
    const float* gsrc = vimg + (long)startl * MD;
    // Piece order of this lane: (jj + 2 level + query slot) & 7.  With 2 level alone the 16 lanes of one ds_read_b128 group used FOUR of
    // the eight 16-byte columns (SQ counters: 61 % of the LDS-active cycles were bank conflicts, LDS active 66 % of the CU time); the
    // query term spreads them over all eight -- two lanes per column instead of four.  The quad shares the slot, so the reduce-scatter
    // below is unchanged and only the two 16-byte output pieces of a lane rotate with it.
    const int qs = (tid >> 2) & 7;
    int rot[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) rot[jj] = ((jj + 2 * p + qs) & 7) * 16;

    auto token_of = [&](int it) -> long { return (long)b * S + tok[it >> 2]; };
    const int total = nq * 4;
    if (tid >= total) return;
    long bq = token_of(tid);
    RowRaw<OT> cur, nxt;
    float2 rf, rf_n;
    cur.load(ow + bq * (long)(M * 48), M, m, p);
    rf = *reinterpret_cast<const float2*>(ref + bq * 8 + 2 * p);

    for (int it = tid; it < total; it += NT) {
        const int itn = it + NT < total ? it + NT : it;          // next iteration's row one iteration ahead
        const long bqn = token_of(itn);
        nxt.load(ow + bqn * (long)(M * 48), M, m, p);
        rf_n = *reinterpret_cast<const float2*>(ref + bqn * 8 + 2 * p);

        float off[8], lg[4];
        cur.get(off, lg);
        float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
        mx = fmaxf(mx, quad_dpp<0xB1>(mx));
        mx = fmaxf(mx, quad_dpp<0x4E>(mx));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { lg[i] = __expf(lg[i] - mx); sum += lg[i]; }
        sum += quad_dpp<0xB1>(sum);
        sum += quad_dpp<0x4E>(sum);
        const float inv = 1.0f / sum;

        encf2_t acc[8][2];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) { acc[jj][0] = encf2_t{0.f, 0.f}; acc[jj][1] = encf2_t{0.f, 0.f}; }
#define DTLR_ACC_F32(D, K)                                                                         \
        {                                                                                          \
            const encf2_t k2_ = {(K), (K)};                                                        \
            _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                     \
                acc[jj][0] = __builtin_elementwise_fma(k2_, encf2_t{__uint_as_float(D[jj].x), __uint_as_float(D[jj].y)}, acc[jj][0]); \
                acc[jj][1] = __builtin_elementwise_fma(k2_, encf2_t{__uint_as_float(D[jj].z), __uint_as_float(D[jj].w)}, acc[jj][1]); \
            }                                                                                      \
        }
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            // the reference's own order, two roundings (ms_deform_attn.py:102-105: ref + off / (W, H)): the product with 1 / W, or an FMA the
            // compiler contracts it into, moves a sampling point by ~1e-5 px at far offsets -- 8e-6 in the output, above the oracle tolerance
            const float lx = __fadd_rn(rf.x, __fdiv_rn(off[2 * pt], fW));
            const float ly = __fadd_rn(rf.y, __fdiv_rn(off[2 * pt + 1], fH));
            // v_med3_f32: a NaN operand yields min3 of the others = -1 (zero weights), like fminf(fmaxf(x, -1), size)
            const float h_im = __builtin_amdgcn_fmed3f(ly * fH - 0.5f, -1.f, fH), w_im = __builtin_amdgcn_fmed3f(lx * fW - 0.5f, -1.f, fW);
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_low = (int)hf, w_low = (int)wf;
            const float lh = h_im - hf, lw = w_im - wf;
            const float a = lg[pt] * inv;
            const float wy0 = (unsigned)h_low < (unsigned)Hl ? 1.f - lh : 0.f, wy1 = (unsigned)(h_low + 1) < (unsigned)Hl ? lh : 0.f;
            const float wx0 = (unsigned)w_low < (unsigned)Wl ? (1.f - lw) * a : 0.f, wx1 = (unsigned)(w_low + 1) < (unsigned)Wl ? lw * a : 0.f;
            const float k1 = wy0 * wx0, k2 = wy0 * wx1, k3 = wy1 * wx0, k4 = wy1 * wx1;
            const int h0 = clamp0_i32(h_low, Hl - 1), h1 = clamp0_i32(h_low + 1, Hl - 1);
            const int w0 = clamp0_i32(w_low, Wl - 1), w1c = clamp0_i32(w_low + 1, Wl - 1);
            if ((w0 >= wc0l) && (w1c < wc1l)) {
                const unsigned char* r0 = win + (h0 * wstride) * 128;
                const unsigned char* r1 = win + (h1 * wstride) * 128;
                const int a0 = (w0 - wc0l) * 128, a1 = (w1c - wc0l) * 128;
                {
                    uint4 d1[8], d2[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        d1[jj] = *reinterpret_cast<const uint4*>(r0 + a0 + rot[jj]);
                        d2[jj] = *reinterpret_cast<const uint4*>(r0 + a1 + rot[jj]);
                    }
                    DTLR_ACC_F32(d1, k1) DTLR_ACC_F32(d2, k2)
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    uint4 d3[8], d4[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        d3[jj] = *reinterpret_cast<const uint4*>(r1 + a0 + rot[jj]);
                        d4[jj] = *reinterpret_cast<const uint4*>(r1 + a1 + rot[jj]);
                    }
                    DTLR_ACC_F32(d3, k3) DTLR_ACC_F32(d4, k4)
                }
            } else if (k1 != 0.f || k2 != 0.f || k3 != 0.f || k4 != 0.f) {      // inside the map, outside the staged window: global path
                const float* g00 = gsrc + (long)(h0 * Wl + w0) * MD;
                const float* g01 = gsrc + (long)(h0 * Wl + w1c) * MD;
                const float* g10 = gsrc + (long)(h1 * Wl + w0) * MD;
                const float* g11 = gsrc + (long)(h1 * Wl + w1c) * MD;
                {
                    uint4 d1[8], d2[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        d1[jj] = *reinterpret_cast<const uint4*>(g00 + (rot[jj] >> 2));
                        d2[jj] = *reinterpret_cast<const uint4*>(g01 + (rot[jj] >> 2));
                    }
                    DTLR_ACC_F32(d1, k1) DTLR_ACC_F32(d2, k2)
                }
                {
                    uint4 d3[8], d4[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        d3[jj] = *reinterpret_cast<const uint4*>(g10 + (rot[jj] >> 2));
                        d4[jj] = *reinterpret_cast<const uint4*>(g11 + (rot[jj] >> 2));
                    }
                    DTLR_ACC_F32(d3, k3) DTLR_ACC_F32(d4, k4)
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef DTLR_ACC_F32
        // quad reduce-scatter: lane p's acc[jj] is piece (jj + 2p + slot) & 7 of its level; lane i collects pieces 2i + slot (acc[0], acc[2]
        // of lane i-1, acc[4] of lane i-2, acc[6] of lane i-3) and 2i + 1 + slot (the odd ones)
        float res[8];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    res[4 * e + 2 * h + c] = (acc[e][h][c] + quad_dpp<0x39>(acc[6 + e][h][c])) + (quad_dpp<0x4E>(acc[4 + e][h][c]) + quad_dpp<0x93>(acc[2 + e][h][c]));
        float* dst = out + bq * MD + m * 32;                        // res[0..3] = piece (2 i + slot) & 7, res[4..7] = the next one
        *reinterpret_cast<float4*>(dst + ((2 * p + qs) & 7) * 4) = make_float4(res[0], res[1], res[2], res[3]);
        *reinterpret_cast<float4*>(dst + ((2 * p + 1 + qs) & 7) * 4) = make_float4(res[4], res[5], res[6], res[7]);
        cur = nxt; rf = rf_n; bq = bqn;
    }
}

// VAR: 16-bit values: 0 first form, 1 packed-fp16 form, 2 third form; fp32 values: 0 first form (8 lanes per (query, head)), 3 level-per-lane form
template <typename T, typename OT, int VAR = 0, int NT = 256>
__global__ __launch_bounds__(NT, VAR == 3 ? 2 : NT == 384 ? 3 : NT / 128) void msda_enc_lds_kernel(
    const T* __restrict__ value, const OT* __restrict__ ow, const float* __restrict__ ref, T* __restrict__ out,
    EncLevels lv, int S, int M, int TW0, int R, int tok_off)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int VEC = ET<T>::VEC, CP = ET<T>::CP;
    constexpr int PIX_BYTES = 32 * (int)sizeof(T);
    const int tid = threadIdx.x;
    const int t = blockIdx.x, m = blockIdx.y, b = blockIdx.z;
    const int MD = M * 32;
    const int W0 = lv.W[0];

    // ---- per-level geometry of this tile (uniform across the workgroup) ---------------------------
    int qc0[4], qn[4], wc0[4], wc1[4], qbase[5];
    qbase[0] = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        qc0[l] = cols_left_of(t, TW0, W0, lv.W[l]);
        const int qc1 = cols_left_of(t + 1, TW0, W0, lv.W[l]);
        qn[l] = qc1 - qc0[l];
        const int lo = (int)(((long)t * TW0 * lv.W[l]) / W0) - R;
        const int hi = (int)((((long)(t + 1) * TW0 * lv.W[l]) + W0 - 1) / W0) + R;
        wc0[l] = max(lo, 0);
        wc1[l] = min(min(hi, lv.W[l]), wc0[l] + lv.wmax[l]);
        qbase[l + 1] = qbase[l] + lv.H[l] * qn[l];
    }
    const int nq = qbase[4];
    if constexpr (VAR != 0) {
        // token table of this tile: tile-local query number -> token index inside the image (read by the query phase)
        int* tok = reinterpret_cast<int*>(smem + tok_off);
        for (int q = tid; q < nq; q += NT) {
            const int lq = q >= qbase[3] ? 3 : q >= qbase[2] ? 2 : q >= qbase[1] ? 1 : 0;
            const int r = q - (lq == 3 ? qbase[3] : lq == 2 ? qbase[2] : lq == 1 ? qbase[1] : 0);
            const int nc = lq == 3 ? qn[3] : lq == 2 ? qn[2] : lq == 1 ? qn[1] : qn[0];
            const int c0q = lq == 3 ? qc0[3] : lq == 2 ? qc0[2] : lq == 1 ? qc0[1] : qc0[0];
            const int Wq = lq == 3 ? lv.W[3] : lq == 2 ? lv.W[2] : lq == 1 ? lv.W[1] : lv.W[0];
            const int stq = lq == 3 ? lv.start[3] : lq == 2 ? lv.start[2] : lq == 1 ? lv.start[1] : lv.start[0];
            if constexpr (VAR == 2 || VAR == 3 || VAR == 4) {
                // r / nc without the ~60-instruction integer division: (r + 0.5) / nc is at least 0.5 / nc away from an integer, far
                // more than the fp32 error of the product for r < 2^15 and nc <= 2^8 (restated and swept in tests/test_host_logic.py)
                const int qi = (int)(((float)r + 0.5f) * __builtin_amdgcn_rcpf((float)nc));
                tok[q] = stq + qi * Wq + c0q + (r - qi * nc);
            } else
            tok[q] = stq + (r / nc) * Wq + c0q + r % nc;
        }
    }

    // ---- stage the four windows of head m: coalesced 16-byte chunks, CP chunks per pixel ------------
    const T* vimg = value + (long)b * S * MD + m * 32;
    if constexpr (VAR == 2 || VAR == 4) {
        // Variant 3 stages with ~45% of the default loop's VALU instructions (270 per four chunks there: two divisions by the runtime window
        // width per chunk, and the fp16 saturation guard as v_max + v_med3 per element): a lane walks its chunks as (row, chunk-in-row)
        // advanced by the constant step (NT / rowlen, NT % rowlen) -- ONE division per level per lane --, and the conversion saturates in
        // hardware: MODE.FP16_OVFL clamps an overflowing fp16 result to +-65504 (it also covers the packed accumulation of the query phase).
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const unsigned ww = (unsigned)(wc1[l] - wc0[l]), rowlen = ww * CP, nchunk = (unsigned)lv.H[l] * rowlen;
            const unsigned dq = (unsigned)NT / rowlen, dr = (unsigned)NT - dq * rowlen;
            const T* src = vimg + ((long)lv.start[l] + wc0[l]) * MD;
            unsigned char* dst = smem + (long)lv.loff[l] * PIX_BYTES;
            unsigned row = (unsigned)tid / rowlen, x = (unsigned)tid - row * rowlen;
            for (unsigned c0 = (unsigned)tid; c0 < nchunk; c0 += NT * 4) {
                uint4 d[4];
                unsigned dst_off[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool live = c0 + NT * u < nchunk;                    // tail lanes re-read chunk (0, 0): a valid address
                    const unsigned r_ = live ? row : 0u, x_ = live ? x : 0u;
                    const unsigned col = x_ / CP, part = x_ % CP;              // CP is a power of two
                    d[u] = *reinterpret_cast<const uint4*>(src + (long)(r_ * (unsigned)lv.W[l] + col) * MD + part * VEC);
                    dst_off[u] = (r_ * (unsigned)lv.wmax[l] + col) * PIX_BYTES + part * 16;
                    x += dr; row += dq;
                    if (x >= rowlen) { x -= rowlen; ++row; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (c0 + NT * u < nchunk) *reinterpret_cast<uint4*>(dst + dst_off[u]) = stage_convert_ovfl<T>(d[u]);
            }
        }
    } else
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const int ww = wc1[l] - wc0[l];
        const int nchunk = lv.H[l] * ww * CP;
        const T* src = vimg + (long)lv.start[l] * MD;
        unsigned char* dst = smem + (long)lv.loff[l] * PIX_BYTES;
        // 4 independent 16-byte loads in flight per lane before the first LDS store (a plain
        // load->store loop is one serialized L2/HBM round trip per chunk: ~19 per thread per workgroup)
        for (int c0 = tid; c0 < nchunk; c0 += NT * 4) {
            uint4 d[4];
            int dst_off[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = min(c0 + NT * u, nchunk - 1);                // clamp: tail lanes re-read a valid chunk
                const int part = c % CP, pc = c / CP;
                const int col = pc % ww, row = pc / ww;
                d[u] = *reinterpret_cast<const uint4*>(src + (long)(row * lv.W[l] + wc0[l] + col) * MD + part * VEC);
                dst_off[u] = (row * lv.wmax[l] + col) * PIX_BYTES + part * 16;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c0 + NT * u < nchunk) *reinterpret_cast<uint4*>(dst + dst_off[u]) = stage_convert<T>(d[u]);
        }
    }
    __syncthreads();

    if constexpr (sizeof(T) == 2) {
        if constexpr (VAR == 0) enc_queries_bf16<OT>(smem, vimg, ow, ref, out, lv, qc0, qn, wc0, wc1, qbase, nq, S, M, m, b);
        else if constexpr (VAR == 1) enc_queries_bf16_h<OT, NT>(smem, reinterpret_cast<const int*>(smem + tok_off), vimg, ow, ref, out, lv, qc0, qn, wc0, wc1, qbase, nq, S, M, m, b);
        else if constexpr (VAR == 4) enc_queries_bf16_h4<OT, NT>(smem, reinterpret_cast<const int*>(smem + tok_off), vimg, ow, ref, out, lv, wc0, wc1, nq, S, M, m, b);
        else enc_queries_bf16_h3<OT, NT>(smem, reinterpret_cast<const int*>(smem + tok_off), vimg, ow, ref, out, lv, wc0, wc1, nq, S, M, m, b);
        return;
    }
    if constexpr (VAR == 3) {
        static_assert(sizeof(T) == 4, "VAR 3 is the fp32 level-per-lane form");
        enc_queries_f32_lvl<OT, NT>(smem, reinterpret_cast<const int*>(smem + tok_off), vimg, ow, ref, out, lv, wc0, wc1, nq, S, M, m, b);
        return;
    }
    static_assert(sizeof(T) == 2 || VAR == 3 || NT == 256, "the first fp32 query phase strides by 256");
    // ---- queries: CP lanes per query (16 bytes = VEC channels each) ---------------------------------
    for (int it = tid; it < nq * CP; it += 256) {
        const int part = it % CP, q = it / CP;
        int lq = 0;
        if (q >= qbase[1]) lq = 1;
        if (q >= qbase[2]) lq = 2;
        if (q >= qbase[3]) lq = 3;
        int r, nc, c0q, Wq, stq;
        switch (lq) {       // scalar selects (arrays indexed by a lane-varying value would go to scratch)
        case 0: r = q - qbase[0]; nc = qn[0]; c0q = qc0[0]; Wq = lv.W[0]; stq = lv.start[0]; break;
        case 1: r = q - qbase[1]; nc = qn[1]; c0q = qc0[1]; Wq = lv.W[1]; stq = lv.start[1]; break;
        case 2: r = q - qbase[2]; nc = qn[2]; c0q = qc0[2]; Wq = lv.W[2]; stq = lv.start[2]; break;
        default: r = q - qbase[3]; nc = qn[3]; c0q = qc0[3]; Wq = lv.W[3]; stq = lv.start[3]; break;
        }
        const int qi = r / nc, qj = c0q + r % nc;
        const long bq = (long)b * S + stq + qi * Wq + qj;

        const OT* row = ow + bq * (long)(M * 48);
        float off[32], lg[16];
        ld16f<OT>(row + m * 32, *reinterpret_cast<float (*)[16]>(&off[0]));
        ld16f<OT>(row + m * 32 + 16, *reinterpret_cast<float (*)[16]>(&off[16]));
        ld16f<OT>(row + M * 32 + m * 16, lg);
        float rf[8];
        {
            const float4 a = reinterpret_cast<const float4*>(ref + bq * 8)[0], c = reinterpret_cast<const float4*>(ref + bq * 8)[1];
            rf[0] = a.x; rf[1] = a.y; rf[2] = a.z; rf[3] = a.w; rf[4] = c.x; rf[5] = c.y; rf[6] = c.z; rf[7] = c.w;
        }
        float mx = lg[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, lg[i]);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { lg[i] = expf(lg[i] - mx); sum += lg[i]; }
        const float inv = 1.0f / sum;

        float col[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) col[i] = 0.f;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int H = lv.H[l], W = lv.W[l];
            const unsigned char* win = smem + (long)lv.loff[l] * PIX_BYTES + part * 16;
            const T* gsrc = vimg + (long)lv.start[l] * MD + part * VEC;
            const int wstride = lv.wmax[l];
            // geometry of the 4 points of this level first (so that, on the common path where every point is
            // inside the map and inside the staged window, all 16 LDS reads are issued back to back)
            int o00[4], o01[4], o10[4], o11[4];               // LDS byte offsets of the 4 corners (clamped)
            float w1[4], w2[4], w3[4], w4[4];                 // bilinear weights x attention, zero for invalid corners
            bool fast = true;
            int gh0[4], gh1[4], gw0[4], gw1[4];
            bool ins[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float lx = rf[2 * l] + off[(l * 4 + p) * 2] / (float)W;
                const float ly = rf[2 * l + 1] + off[(l * 4 + p) * 2 + 1] / (float)H;
                const float h_im = ly * (float)H - 0.5f, w_im = lx * (float)W - 0.5f;
                const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
                const float hf = floorf(h_im), wf = floorf(w_im);
                const int h_low = (int)fminf(fmaxf(hf, -1.f), (float)H), w_low = (int)fminf(fmaxf(wf, -1.f), (float)W);
                const int h_high = h_low + 1, w_high = w_low + 1;
                const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                const bool top = inside && h_low >= 0, bot = inside && h_high <= H - 1, left = w_low >= 0, right = w_high <= W - 1;
                const int h0 = min(max(h_low, 0), H - 1), h1 = max(min(h_high, H - 1), 0);
                const int w0 = min(max(w_low, 0), W - 1), w1c = max(min(w_high, W - 1), 0);
                gh0[p] = h0; gh1[p] = h1; gw0[p] = w0; gw1[p] = w1c; ins[p] = inside;
                const bool res = (w0 >= wc0[l]) && (w1c < wc1[l]);
                fast = fast && (res || !inside);
                // clamp into the STAGED columns [0, ww): a point outside the map (weights all zero) must still read
                // initialised LDS -- uninitialised bits can be NaN/Inf and 0 * NaN = NaN
                const int ww_ = wc1[l] - wc0[l];
                const int a0 = min(max(w0 - wc0[l], 0), ww_ - 1), a1 = min(max(w1c - wc0[l], 0), ww_ - 1);
                o00[p] = (h0 * wstride + a0) * PIX_BYTES; o01[p] = (h0 * wstride + a1) * PIX_BYTES;
                o10[p] = (h1 * wstride + a0) * PIX_BYTES; o11[p] = (h1 * wstride + a1) * PIX_BYTES;
                // the reference multiplies val = w1 v1 + w2 v2 + w3 v3 + w4 v4 by the attention weight afterwards;
                // keep that order: the weights below are the pure bilinear ones, zeroed for corners outside the map
                w1[p] = (top && left) ? hh * hw : 0.f;  w2[p] = (top && right) ? hh * lw : 0.f;
                w3[p] = (bot && left) ? lh * hw : 0.f;  w4[p] = (bot && right) ? lh * lw : 0.f;
            }
            if (fast) {
                uint4 d[4][4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    d[p][0] = *reinterpret_cast<const uint4*>(win + o00[p]);
                    d[p][1] = *reinterpret_cast<const uint4*>(win + o01[p]);
                    d[p][2] = *reinterpret_cast<const uint4*>(win + o10[p]);
                    d[p][3] = *reinterpret_cast<const uint4*>(win + o11[p]);
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float v1[VEC], v2[VEC], v3[VEC], v4[VEC];
                    ET<T>::unpack(d[p][0], v1); ET<T>::unpack(d[p][1], v2); ET<T>::unpack(d[p][2], v3); ET<T>::unpack(d[p][3], v4);
                    const float a = lg[l * 4 + p] * inv;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) col[i] += (w1[p] * v1[i] + w2[p] * v2[i] + w3[p] * v3[i] + w4[p] * v4[i]) * a;
                }
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (!ins[p]) continue;
                    uint4 d1, d2, d3, d4;
                    if (gw0[p] >= wc0[l] && gw1[p] < wc1[l]) {
                        d1 = *reinterpret_cast<const uint4*>(win + o00[p]); d2 = *reinterpret_cast<const uint4*>(win + o01[p]);
                        d3 = *reinterpret_cast<const uint4*>(win + o10[p]); d4 = *reinterpret_cast<const uint4*>(win + o11[p]);
                    } else {                                    // outside the staged window: global path
                        d1 = *reinterpret_cast<const uint4*>(gsrc + (long)(gh0[p] * W + gw0[p]) * MD);
                        d2 = *reinterpret_cast<const uint4*>(gsrc + (long)(gh0[p] * W + gw1[p]) * MD);
                        d3 = *reinterpret_cast<const uint4*>(gsrc + (long)(gh1[p] * W + gw0[p]) * MD);
                        d4 = *reinterpret_cast<const uint4*>(gsrc + (long)(gh1[p] * W + gw1[p]) * MD);
                    }
                    float v1[VEC], v2[VEC], v3[VEC], v4[VEC];
                    ET<T>::unpack(d1, v1); ET<T>::unpack(d2, v2); ET<T>::unpack(d3, v3); ET<T>::unpack(d4, v4);
                    const float a = lg[l * 4 + p] * inv;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) col[i] += (w1[p] * v1[i] + w2[p] * v2[i] + w3[p] * v3[i] + w4[p] * v4[i]) * a;
                }
            }
        }
        *reinterpret_cast<uint4*>(out + bq * MD + m * 32 + part * VEC) = ET<T>::pack(col);
    }
}

struct EncPlan { EncLevels lv; int S, TW0, R, ntiles, tok_off; size_t lds; };

// ---- far-sample probe: how many of a launch's sampling points would take msda_enc_lds_kernel's global path? -----------------
// Same tile geometry and the same coordinate arithmetic as the query phase above, nothing else: a lane reads its (query, head, level)
// slice of the projection row and counts the points that fall inside the map but outside the tile's staged column window.
// counts[0] += such points, counts[1] += points inside the map.  One wave-instruction stalls on 16 dependent global loads per far
// point, so a few percent of them cost more than the gather kernel's flat time (tools/msda_sweep.py): the engine runs this probe
// now and then and picks the kernel per layer (DTLREngine._msda_mode).
template <typename OT>
__global__ __launch_bounds__(256) void msda_enc_far_count_kernel(const OT* __restrict__ ow, const float* __restrict__ ref, EncLevels lv,
                                                                  int S, int M, int TW0, int R, unsigned long long* __restrict__ counts)
{
    const int tid = threadIdx.x, p = tid & 3;
    const int t = blockIdx.x, m = blockIdx.y, b = blockIdx.z;
    const int W0 = lv.W[0];
    int qc0[4], qn[4], wc0[4], wc1[4], qbase[5];
    qbase[0] = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        qc0[l] = cols_left_of(t, TW0, W0, lv.W[l]);
        qn[l] = cols_left_of(t + 1, TW0, W0, lv.W[l]) - qc0[l];
        const int lo = (int)(((long)t * TW0 * lv.W[l]) / W0) - R;
        const int hi = (int)((((long)(t + 1) * TW0 * lv.W[l]) + W0 - 1) / W0) + R;
        wc0[l] = max(lo, 0);
        wc1[l] = min(min(hi, lv.W[l]), wc0[l] + lv.wmax[l]);
        qbase[l + 1] = qbase[l] + lv.H[l] * qn[l];
    }
    const int nq = qbase[4];
    const int Hl = lv.H[p], Wl = lv.W[p], wc0l = wc0[p], wc1l = wc1[p];
    const float fH = (float)Hl, fW = (float)Wl, invH = 1.0f / fH, invW = 1.0f / fW;
    unsigned far = 0, ins = 0;
    for (int it = tid; it < nq * 4; it += 256) {
        const int q = it >> 2;
        const int lq = q >= qbase[3] ? 3 : q >= qbase[2] ? 2 : q >= qbase[1] ? 1 : 0;
        const int r = q - qbase[lq];
        const long bq = (long)b * S + lv.start[lq] + (r / qn[lq]) * lv.W[lq] + qc0[lq] + r % qn[lq];
        RowRaw<OT> row;
        row.load(ow + bq * (long)(M * 48), M, m, p);
        const float2 rf = *reinterpret_cast<const float2*>(ref + bq * 8 + 2 * p);
        float off[8], lg[4];
        row.get(off, lg);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const float lx = rf.x + off[2 * pt] * invW, ly = rf.y + off[2 * pt + 1] * invH;
            const float h_im = ly * fH - 0.5f, w_im = lx * fW - 0.5f;
            const bool inside = h_im > -1.f && w_im > -1.f && h_im < fH && w_im < fW;
            const int w_low = (int)fminf(fmaxf(floorf(w_im), -1.f), fW);
            const int w0 = min(max(w_low, 0), Wl - 1), w1c = max(min(w_low + 1, Wl - 1), 0);
            const bool staged = (w0 >= wc0l) && (w1c < wc1l);
            ins += inside ? 1u : 0u;
            far += (inside && !staged) ? 1u : 0u;
        }
    }
    __shared__ unsigned red[2][4];
    far = (unsigned)wave_sum((float)far);                       // <= 64 x 4 x iterations: exact in fp32 for the tile sizes of the plan
    ins = (unsigned)wave_sum((float)ins);
    if ((tid & 63) == 0) { red[0][tid >> 6] = far; red[1][tid >> 6] = ins; }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(counts, (unsigned long long)(red[0][0] + red[0][1] + red[0][2] + red[0][3]));
        atomicAdd(counts + 1, (unsigned long long)(red[1][0] + red[1][1] + red[1][2] + red[1][3]));
    }
}

static bool make_plan(const int* hw, int elem, int R, EncPlan& pl) {
    int start = 0;
    for (int l = 0; l < 4; ++l) {
        pl.lv.H[l] = hw[2 * l]; pl.lv.W[l] = hw[2 * l + 1]; pl.lv.start[l] = start;
        if (pl.lv.H[l] <= 0 || pl.lv.W[l] <= 0) return false;
        start += hw[2 * l] * hw[2 * l + 1];
    }
    pl.S = start; pl.R = R;
    const int W0 = pl.lv.W[0];
    // widest level-0 tile whose windows fit: first try 2 workgroups per CU (<= 80 KB), then 1 (<= 160 KB)
    for (int pass = 0; pass < 2; ++pass) {
        const size_t cap = pass == 0 ? 80 * 1024 : 160 * 1024;
        for (int TW0 = 64; TW0 >= (pass == 0 ? 16 : 4); TW0 >>= 1) {
            long pix = 0;
            for (int l = 0; l < 4; ++l) {
                pl.lv.wmax[l] = (int)(((long)TW0 * pl.lv.W[l] + W0 - 1) / W0) + 2 * R + 1;
                if (pl.lv.wmax[l] > pl.lv.W[l]) pl.lv.wmax[l] = pl.lv.W[l];
                pl.lv.loff[l] = (int)pix;
                pix += (long)pl.lv.H[l] * pl.lv.wmax[l];
            }
            // + the token table of the tile (one int per query: at most ceil(TW0 W_l / W_0) + 1 columns of every row of every level)
            long nqmax = 0;
            for (int l = 0; l < 4; ++l) nqmax += (long)pl.lv.H[l] * ((((long)TW0 * pl.lv.W[l] + W0 - 1) / W0) + 1);
            const size_t win = ((size_t)pix * 32 * elem + 15) & ~(size_t)15;
            const size_t lds = win + (size_t)nqmax * 4;
            if (lds <= cap) {
                pl.TW0 = TW0; pl.lds = lds; pl.tok_off = (int)win; pl.ntiles = (W0 + TW0 - 1) / TW0;
                return true;
            }
        }
    }
    return false;
}

template <typename T, typename OT, int VAR = 0, int NT = 256>
static int launch_enc(const void* value, const void* ow, const float* ref, void* out, const EncPlan& pl, int N, int M, hipStream_t st) {
    static DevOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute((const void*)msda_enc_lds_kernel<T, OT, VAR, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
    hipLaunchKernelGGL((msda_enc_lds_kernel<T, OT, VAR, NT>), dim3(pl.ntiles, M, N), dim3(NT), pl.lds, st,
                       (const T*)value, (const OT*)ow, ref, (T*)out, pl.lv, pl.S, M, pl.TW0, pl.R, pl.tok_off);
    return check_launch();
}
// 16-bit query-phase form: 3 = the third form, 512 threads (the default since round 4: same-box A/B of the step 9.24 -> 9.12 ms, tests green on
// hardware); experiment builds only (env DTLR_MSDA_ENC_V, read once per process): 0 first form (fp32 accumulators, v_fma_mix), 1 packed-fp16 form
// with 256 threads, 2 the same with 512 threads (the round-2/3 default).  The product libraries have no run-time knob.
static int enc_variant() {
    static const int v = exp_env_int("DTLR_MSDA_ENC_V", 3);
    return (v >= 0 && v <= 4) ? v : 3;
}

}  // namespace dtlr

using namespace dtlr;

// 1 when the LDS window plan of dtlr_msda_encoder_forward fits these level shapes (full-height column windows + halo of all four
// levels within 160 KB), 0 when it does not (tall canvases: the caller then uses the gather kernel, dtlr_msda_fused_forward,
// which has no size limit), negative on bad arguments.
extern "C" int dtlr_msda_encoder_plan_ok(const int* level_hw, int dtype, int halo)
{
    if (!level_hw || halo < 0) return DTLR_EINVAL;
    if (dtype != DTLR_F32 && dtype != DTLR_H16) return DTLR_EDTYPE;
    EncPlan pl;
    return make_plan(level_hw, dtype == DTLR_F32 ? 4 : 2, halo, pl) ? 1 : 0;
}

// counts[0] += sampling points of this launch that dtlr_msda_encoder_forward would fetch through its global path (inside the map,
// outside the tile's staged column window), counts[1] += points inside the map.  counts: 2 x uint64 in device memory (accumulated,
// not reset).  Same arguments as dtlr_msda_encoder_forward where they are shared; dtype = the VALUE dtype (it sets the window plan).
extern "C" int dtlr_msda_encoder_far_samples(const void* ow, const float* ref, const int* level_hw, int N, int M, int halo,
                                             int dtype, int ow_dtype, unsigned long long* counts, void* stream)
{
    clear_stale_error();
    if (!ow || !ref || !level_hw || !counts) return DTLR_EINVAL;
    if (N <= 0 || M <= 0 || halo < 0) return DTLR_EINVAL;
    EncPlan pl;
    if (!make_plan(level_hw, dtype == DTLR_F32 ? 4 : 2, halo, pl)) return DTLR_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (ow_dtype == DTLR_F32)
        hipLaunchKernelGGL(msda_enc_far_count_kernel<float>, dim3(pl.ntiles, M, N), dim3(256), 0, st, (const float*)ow, ref, pl.lv, pl.S, M, pl.TW0, pl.R, counts);
    else if (ow_dtype == DTLR_H16)
        hipLaunchKernelGGL(msda_enc_far_count_kernel<uint16_t>, dim3(pl.ntiles, M, N), dim3(256), 0, st, (const uint16_t*)ow, ref, pl.lv, pl.S, M, pl.TW0, pl.R, counts);
    else return DTLR_EDTYPE;
    return check_launch();
}

extern "C" int dtlr_msda_encoder_forward(const void* value, const void* ow, const float* ref, const int* level_hw,
                                         int N, int M, int D, int L, int P, int halo,
                                         int dtype, int ow_dtype, void* out, void* stream)
{
    clear_stale_error();
    if (!value || !ow || !ref || !level_hw || !out) return DTLR_EINVAL;
    if (N <= 0 || M <= 0 || halo < 0) return DTLR_EINVAL;
    if (L != 4 || P != 4 || D != 32) return DTLR_ESHAPE;
    EncPlan pl;
    const int elem = dtype == DTLR_F32 ? 4 : 2;
    if (!make_plan(level_hw, elem, halo, pl)) return DTLR_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DTLR_F32 && ow_dtype == DTLR_F32) {
        static const bool first_form = exp_env_int("DTLR_MSDA_ENC_F32_V", 3) == 0;      // experiment builds: =0 the first fp32 form (A/B timing)
        if (first_form) return launch_enc<float, float>(value, ow, ref, out, pl, N, M, st);
        return launch_enc<float, float, 3, 512>(value, ow, ref, out, pl, N, M, st);
    }
    if (dtype == DTLR_H16 && ow_dtype == DTLR_F32) {
        if (enc_variant() == 0) return launch_enc<uint16_t, float>(value, ow, ref, out, pl, N, M, st);
        if (enc_variant() == 1) return launch_enc<uint16_t, float, 1, 256>(value, ow, ref, out, pl, N, M, st);
        if (enc_variant() == 2) return launch_enc<uint16_t, float, 1, 512>(value, ow, ref, out, pl, N, M, st);
#ifdef DTLR_EXPERIMENT
        if (enc_variant() == 4) return launch_enc<uint16_t, float, 4, 384>(value, ow, ref, out, pl, N, M, st);
#endif
        return launch_enc<uint16_t, float, 2, 512>(value, ow, ref, out, pl, N, M, st);
    }
    if (dtype == DTLR_H16 && ow_dtype == DTLR_H16) {
        if (enc_variant() == 0) return launch_enc<uint16_t, uint16_t>(value, ow, ref, out, pl, N, M, st);
        if (enc_variant() == 1) return launch_enc<uint16_t, uint16_t, 1, 256>(value, ow, ref, out, pl, N, M, st);
        if (enc_variant() == 2) return launch_enc<uint16_t, uint16_t, 1, 512>(value, ow, ref, out, pl, N, M, st);
#ifdef DTLR_EXPERIMENT
        if (enc_variant() == 4) return launch_enc<uint16_t, uint16_t, 4, 384>(value, ow, ref, out, pl, N, M, st);
#endif
        return launch_enc<uint16_t, uint16_t, 2, 512>(value, ow, ref, out, pl, N, M, st);
    }
    return DTLR_EDTYPE;
}
