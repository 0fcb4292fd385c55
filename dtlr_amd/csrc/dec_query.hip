// The query stage of a decoder layer in ONE launch (16-bit engines):
//
//     ref_in[q, l, :] = ref[q, :] * (vr[b,l,0], vr[b,l,1], vr[b,l,0], vr[b,l,1])                    deformable_transformer.py:684-690
//     sine[q, :]      = [emb(y) | emb(x) | emb(w) | emb(h)] of ref_in[q, 0, :]                       models/dino/utils.py:141-167
//     qpos            = ref_point_head(sine) = (relu(sine W0^T + b0)) W1^T + b1                      deformable_transformer.py:690-692
//     [q | k]         = (tgt + qpos) Wqk^T + bqk          v = tgt Wv^T + bv                          nn.MultiheadAttention in_proj, :904-907
//
// Round 2 ran this as six launches per layer (query_prep, two GEMMs of the MLP, the [q|k] GEMM with its A + A2 prologue, the v GEMM) --
// 113 us of a 376 us layer for 23 GFLOP, each launch a single partial wave of 225 tiles on 256 CUs with its own fill and drain, and
// 29 + 15 + 15 MB of intermediates through HBM.  Here a workgroup owns 128 queries and walks the whole chain with the activations in
// LDS; only ref_in, qpos (the cross-attention needs it again), [q|k] and v leave the chip.
//
// Structure (one workgroup = 128 queries, 8 waves, one workgroup per CU; 225 workgroups at B = 32):
//   * MFMA orientation of the other kernels: A-operand = 16 weight rows x 32 k, B-operand = 32 k x 16 queries, so a lane's four
//     accumulators of a tile are four CONSECUTIVE channels of one query (8-byte stores, no transposes between the GEMMs).
//   * wave w owns the output channels [w N/8, (w+1) N/8) of every GEMM (2 row tiles for N = 256; N = 512 as two passes of 2) for all 128 queries:
//     a weight fragment is read ONCE per workgroup, straight from global memory (L2) into registers -- a whole GEMM stage's
//     fragments at once, one stage ahead of their use (dq_fetch) -- and feeds 8 MFMAs; the activation fragments come from the LDS tile (row pitch = row bytes + 16: conflict-free ds_read_b128) and
//     are shared by the wave's row tiles.
//   * the 512-wide sine embedding never exists as a whole: it is produced a quarter (one coordinate, 128 channels) at a time into a
//     ping-pong buffer while the previous quarter is being multiplied (K = 512 as 4 x 128).
//   * LDS: region 0 = two sine quarters (2 x 34 KB), later the A = tgt + qpos tile; region 1 = the hidden tile H, later the tgt tile.
// Measured (MI355X, B = 32: 225 workgroups; back-to-back launches, builds with parts switched off): whole kernel 49 us (55 us inside
// the decoder loop) against 113 us for the six launches; without the MFMAs 39 us, without the weight fetch 44 us (it was 40 of 57 us
// before the weights were packed in fragment order and requested a stage ahead), without MFMAs, fetch, LDS reads and sine 28 us: the
// remainder is the launch itself (~8 us for any 225-workgroup kernel here), the dependent chain of reference-box loads, barriers and
// epilogues, and the 262 KB of output per workgroup (written as 16-byte pieces, 64 contiguous bytes per query: dq_store_pair).
// (Round 3 also fused the NEXT pair of operators the same way -- out_proj + residual + LayerNorm followed by the cross-attention's
// [offsets|logits] projection of (tgt + qpos), statistics exchanged between the waves through LDS: 40.7 us against 17.6 + 23.8 us for the
// two launches, i.e. nothing gained; with the second GEMM as three single-tile passes 54.9 us.  Removed.)
// Arithmetic identical to the unfused path (same MFMA, k ascending, same roundings: sine, H, qpos, tgt + qpos and the outputs are
// rounded to the 16-bit format exactly where the separate kernels stored them), so the results are bit-identical to it.
#include "dtlr_common.h"

namespace dtlr {

typedef __attribute__((ext_vector_type(8))) h16_hw_t dq_h16x8_t;
typedef __attribute__((ext_vector_type(4))) float dq_f32x4_t;

constexpr int DQ_ROWS = 128;                      // queries per workgroup (NTT = 8 query tiles of 16); the small-batch form: 32 (NTT = 2)
constexpr int DQ_PITCH_Q = 128 * 2 + 16;          // sine quarter tile: 128 channels per row
constexpr int DQ_PITCH = 256 * 2 + 16;            // 256-channel tiles (H, A, tgt)
constexpr int DQ_R0 = 2 * DQ_ROWS * DQ_PITCH_Q;   // 69632 B: two sine quarters | the A tile (67584 B)   (sizes of the 128-query form; the
constexpr int DQ_R1 = DQ_ROWS * DQ_PITCH;         // 67584 B: H | tgt                                     32-query form uses a quarter of each)
constexpr int DQ_RCP = DQ_R0 + DQ_R1;             // 128 floats: 1 / dim_t
constexpr int DQ_LDS = DQ_RCP + 128 * 4;
template <int NTT> struct DqCfg {
    static constexpr int ROWS = 16 * NTT, R0 = 2 * ROWS * DQ_PITCH_Q, R1 = ROWS * DQ_PITCH, RCP = R0 + R1, LDS = RCP + 128 * 4;
    static constexpr int TPT = 512 / ROWS;        // threads per query in the sine / reference-box stage: 4 or 16
    static constexpr int PPT = 64 / TPT;          // (sin, cos) pairs per thread and quarter: 16 or 4
};

__device__ __forceinline__ dq_f32x4_t dq_mma(const uint4& a, const uint4& b, dq_f32x4_t c) {
    return DTLR_MFMA_16x16x32_H16(__builtin_bit_cast(dq_h16x8_t, a), __builtin_bit_cast(dq_h16x8_t, b), c, 0, 0, 0);
}

// The weight fragments of one GEMM stage of a wave: 2 row tiles x KT / 32 k-steps.  Weights are handed over PACKED in fragment order
// (ops.dq_pack / dtlr_dq_pack_weights): unit u = output channels [32 u, 32 u + 32), image [unit][k-step][tile 2][lane 64][8 elements],
// lane (m, g) of tile t <- W[32 u + 16 t + m][32 ks + 8 g ..]: one load instruction of a wave is ONE contiguous KB (8 full cache lines;
// from the row-major weight it was 16 half lines).  ALL of a stage's fragments are requested at once, one stage AHEAD of their use
// (an L2 round trip is ~2000 cycles, a k-step of this kernel 256): the first version fetched one k-step ahead and waited 40 of its 67 us.
template <int KT>
__device__ __forceinline__ void dq_fetch(uint4 (&a)[KT / 32][2], const uint16_t* __restrict__ Wp, int unit, int ksteps_total, int ks0, int lane)
{
    const uint4* base = reinterpret_cast<const uint4*>(Wp) + ((long)unit * ksteps_total + ks0) * 128 + lane;
#pragma unroll
    for (int ks = 0; ks < KT / 32; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) a[ks][t] = base[(ks * 2 + t) * 64];
}

// acc[t][tt] += W-fragments . X[16 tt + n][0 .. KT)^T for the wave's NT row tiles and the 8 query tiles; xt: LDS tile (rows = queries),
// `pitch` bytes per row.  One activation fragment (ds_read_b128) feeds the NT MFMAs of its k-step.
template <int KT, int NT, int NTT>
__device__ __forceinline__ void dq_gemm(dq_f32x4_t (&acc)[NT][NTT], const uint4 (&a)[KT / 32][NT], const unsigned char* xt, int pitch, int lane)
{
    const unsigned char* xp = xt + (lane & 15) * pitch + 16 * (lane >> 4);
#pragma unroll
    for (int ks = 0; ks < KT / 32; ++ks) {
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const uint4 b = *reinterpret_cast<const uint4*>(xp + (16 * tt) * pitch + 64 * ks);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][tt] = dq_mma(a[ks][t], b, acc[t][tt]);
        }
    }
}

// Store the wave's two row tiles (32 channels) of one query tile as 16-byte pieces: the tiles are paired with v_permlane16_swap so that a
// lane holds 8 consecutive channels (even g: channels 8 (g/2).. of tile 0, odd g: the same 8 channels of tile 1) -- 64 contiguous bytes
// per query and instruction instead of two 32-byte pieces (the output of a workgroup is 262 KB).
__device__ __forceinline__ void dq_store_pair(uint16_t* __restrict__ row /* &out[q][ch0] */, const dq_f32x4_t& c0, const dq_f32x4_t& c1,
                                              const float4& b0, const float4& b1, int g, bool live)
{
    const uint32_t lo0 = pack_bf16x2(c0[0] + b0.x, c0[1] + b0.y), hi0 = pack_bf16x2(c0[2] + b0.z, c0[3] + b0.w);
    const uint32_t lo1 = pack_bf16x2(c1[0] + b1.x, c1[1] + b1.y), hi1 = pack_bf16x2(c1[2] + b1.z, c1[3] + b1.w);
    const auto s0 = __builtin_amdgcn_permlane16_swap(lo0, lo1, false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(hi0, hi1, false, false);
    if (live) *reinterpret_cast<uint4*>(row + (g & 1) * 16 + 8 * (g >> 1)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
}

template <int NT, int NTT>
__device__ __forceinline__ void dq_zero(dq_f32x4_t (&acc)[NT][NTT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) acc[t][tt] = dq_f32x4_t{0.f, 0.f, 0.f, 0.f};
}

// one sine quarter: coordinate value c[token] (already x 2 pi), 128 channels = 64 (sin, cos) pairs; thread -> (token = tid / TPT, PPT pairs)
template <int NTT>
__device__ __forceinline__ void dq_sine_quarter(unsigned char* dst, float c, const float* rcp, int tid)
{
    constexpr int TPT = DqCfg<NTT>::TPT, PPT = DqCfg<NTT>::PPT;
    const int tok = tid / TPT, p0 = (tid % TPT) * PPT;
    unsigned char* row = dst + tok * DQ_PITCH_Q + p0 * 4;
#pragma unroll
    for (int j = 0; j < PPT; j += 4) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = p0 + j + i;
            w[i] = pack_bf16x2(__sinf(c * rcp[2 * p]), __cosf(c * rcp[2 * p + 1]));      // the roundings of query_prep_kernel (misc.hip)
        }
        *reinterpret_cast<uint4*>(row + j * 4) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

template <int NTT>
__global__ __launch_bounds__(512) void dec_query_stage_kernel(
    const float* __restrict__ ref, const float* __restrict__ vr, const float* __restrict__ dim_t, const uint16_t* __restrict__ tgt,
    const uint16_t* __restrict__ W0, const float* __restrict__ b0, const uint16_t* __restrict__ W1, const float* __restrict__ b1,
    const uint16_t* __restrict__ Wqk, const float* __restrict__ bqk, const uint16_t* __restrict__ Wv, const float* __restrict__ bv,
    float* __restrict__ ref_in, uint16_t* __restrict__ qpos, uint16_t* __restrict__ qk, uint16_t* __restrict__ v,
    long Q, int nq, int L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Cfg = DqCfg<NTT>;
    constexpr int ROWS = Cfg::ROWS, TPT = Cfg::TPT;
    unsigned char* r0 = smem;
    unsigned char* r1 = smem + Cfg::R0;
    float* rcp = reinterpret_cast<float*>(smem + Cfg::RCP);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const long q0 = (long)blockIdx.x * ROWS;

    // ---- reference boxes of this workgroup's queries: ref_in for every level, the four sine arguments of level 0 ----------------
    if (tid < 128) rcp[tid] = __frcp_rn(dim_t[tid]);
    const long qs = min(q0 + (tid / TPT), Q - 1);                 // thread -> (query tid / TPT, level / pair group tid % TPT)
    const int bs = (int)(qs / nq);
    const float4 rr = *reinterpret_cast<const float4*>(ref + qs * 4);
    {
        const int lv = tid % TPT;
        if (lv < L && q0 + (tid / TPT) < Q) {
            const float vx = vr[(bs * L + lv) * 2], vy = vr[(bs * L + lv) * 2 + 1];
            *reinterpret_cast<float4*>(ref_in + (qs * L + lv) * 4) = make_float4(rr.x * vx, rr.y * vy, rr.z * vx, rr.w * vy);
        }
        for (int lv2 = TPT + (tid % TPT); lv2 < L; lv2 += TPT) {  // more levels than threads per query (not the reference's configuration)
            if (q0 + (tid / TPT) < Q) {
                const float vx = vr[(bs * L + lv2) * 2], vy = vr[(bs * L + lv2) * 2 + 1];
                *reinterpret_cast<float4*>(ref_in + (qs * L + lv2) * 4) = make_float4(rr.x * vx, rr.y * vy, rr.z * vx, rr.w * vy);
            }
        }
    }
    const float vx0 = vr[(bs * L) * 2], vy0 = vr[(bs * L) * 2 + 1];
    const float scale = 6.283185307179586f;
    const float cq[4] = {rr.y * vy0 * scale, rr.x * vx0 * scale, rr.z * vx0 * scale, rr.w * vy0 * scale};   // order y, x, w, h
    __syncthreads();                                              // rcp table

    // ---- H = relu(sine W0^T + b0): K = 512 as four quarters; quarter qd + 1 (its sine values AND its weight fragments) is produced /
    //      requested while quarter qd is multiplied ---------------------------------------------------------------------------------
    uint4 wq[2][4][2];                                            // two quarters of W0 fragments in flight
    uint4 w1f[8][2];
    dq_fetch<128>(wq[0], W0, wave, 16, 0, lane);
    dq_sine_quarter<NTT>(r0, cq[0], rcp, tid);
    __syncthreads();
    {
        dq_f32x4_t acc[2][NTT];
        dq_zero<2, NTT>(acc);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            if (qd + 1 < 4) {
                dq_fetch<128>(wq[(qd + 1) & 1], W0, wave, 16, 4 * (qd + 1), lane);
                dq_sine_quarter<NTT>(r0 + ((qd + 1) & 1) * (ROWS * DQ_PITCH_Q), cq[qd + 1], rcp, tid);
            } else {
                dq_fetch<256>(w1f, W1, wave, 8, 0, lane);      // the next stage's weights
            }
            dq_gemm<128, 2, NTT>(acc, wq[qd & 1], r0 + (qd & 1) * (ROWS * DQ_PITCH_Q), DQ_PITCH_Q, lane);
            __syncthreads();
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ch = wave * 32 + 16 * t + 4 * g;
            const float4 bb = *reinterpret_cast<const float4*>(b0 + ch);
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                const dq_f32x4_t c = acc[t][tt];
                *reinterpret_cast<uint2*>(r1 + (16 * tt + n) * DQ_PITCH + ch * 2) =
                    make_uint2(pack_bf16x2(fmaxf(c[0] + bb.x, 0.f), fmaxf(c[1] + bb.y, 0.f)), pack_bf16x2(fmaxf(c[2] + bb.z, 0.f), fmaxf(c[3] + bb.w, 0.f)));
            }
        }
    }
    __syncthreads();

    // ---- qpos = H W1^T + b1 ; A = tgt + qpos -> region 0, tgt -> region 1, qpos -> global --------------------------------------
    uint4 wf[8][2];                                               // the [q|k] projection's first half (row tiles 0, 1 of the wave's four)
    {
        dq_f32x4_t acc[2][NTT];
        dq_zero<2, NTT>(acc);
        dq_fetch<256>(wf, Wqk, 2 * wave, 8, 0, lane);
        // this lane's tgt values (the residual input of A = tgt + qpos and the v projection's operand): requested BEFORE the GEMM so that
        // their HBM latency hides behind it (loaded inside the epilogue loop they cost a dependent round trip per tile: 16 x ~2 us)
        uint2 tq[2][NTT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt)
                tq[t][tt] = *reinterpret_cast<const uint2*>(tgt + min(q0 + 16 * tt + n, Q - 1) * 256 + wave * 32 + 16 * t + 4 * g);
        dq_gemm<256, 2, NTT>(acc, w1f, r1, DQ_PITCH, lane);
        __syncthreads();                                          // every wave has finished reading H: region 1 may take the tgt tile
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ch = wave * 32 + 16 * t + 4 * g;
            const float4 bb = *reinterpret_cast<const float4*>(b1 + ch);
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                const long q = q0 + 16 * tt + n;
                const bool live = q < Q;
                const uint2 tw = tq[t][tt];
                const dq_f32x4_t c = acc[t][tt];
                const uint2 pw = make_uint2(pack_bf16x2(c[0] + bb.x, c[1] + bb.y), pack_bf16x2(c[2] + bb.z, c[3] + bb.w));
                if (live) *reinterpret_cast<uint2*>(qpos + q * 256 + ch) = pw;
                const uint2 aw = make_uint2(pack_bf16x2(h16_lo(tw.x) + h16_lo(pw.x), h16_hi(tw.x) + h16_hi(pw.x)),
                                            pack_bf16x2(h16_lo(tw.y) + h16_lo(pw.y), h16_hi(tw.y) + h16_hi(pw.y)));
                *reinterpret_cast<uint2*>(r0 + (16 * tt + n) * DQ_PITCH + ch * 2) = aw;
                *reinterpret_cast<uint2*>(r1 + (16 * tt + n) * DQ_PITCH + ch * 2) = tw;
            }
        }
    }
    __syncthreads();

    // ---- [q | k] = A Wqk^T + bqk (N = 512: the wave's 64 channels as two halves of two row tiles), then v = tgt Wv^T + bv; the next
    //      stage's weights are always in flight behind the current one -------------------------------------------------------------
#define DQ_OUT_STAGE(WF, XT, OUT, LDO, CH0, BIAS)                                                  \
    {                                                                                              \
        dq_f32x4_t acc[2][NTT];                                                                    \
        dq_zero<2, NTT>(acc);                                                                      \
        dq_gemm<256, 2, NTT>(acc, WF, XT, DQ_PITCH, lane);                                         \
        const float4 bb0 = *reinterpret_cast<const float4*>((BIAS) + (CH0) + 4 * g);               \
        const float4 bb1 = *reinterpret_cast<const float4*>((BIAS) + (CH0) + 16 + 4 * g);          \
        _Pragma("unroll") for (int tt = 0; tt < NTT; ++tt) {                                       \
            const long q = q0 + 16 * tt + n;                                                       \
            dq_store_pair((OUT) + min(q, Q - 1) * (LDO) + (CH0), acc[0][tt], acc[1][tt], bb0, bb1, g, q < Q); \
        }                                                                                          \
    }
    uint4 wg[8][2];
    dq_fetch<256>(wg, Wqk, 2 * wave + 1, 8, 0, lane);
    DQ_OUT_STAGE(wf, r0, qk, 512, wave * 64, bqk)
    dq_fetch<256>(wf, Wv, wave, 8, 0, lane);
    DQ_OUT_STAGE(wg, r0, qk, 512, wave * 64 + 32, bqk)
    DQ_OUT_STAGE(wf, r1, v, 256, wave * 32, bv)
#undef DQ_OUT_STAGE
}

}  // namespace dtlr

using namespace dtlr;

// [N, K] row-major 16-bit weight (host memory) -> the fragment-order image of dec_query_stage_kernel (host memory, same size):
// out[(((u * K/32 + ks) * 2 + t) * 64 + lane) * 8 + e] = W[32 u + 16 t + (lane & 15)][32 ks + 8 (lane >> 4) + e]
extern "C" int dtlr_dq_pack_weights(const unsigned short* w_host, unsigned short* out_host, int N, int K)
{
    if (!w_host || !out_host) return DTLR_EINVAL;
    if (N <= 0 || K <= 0 || (N & 31) || (K & 31)) return DTLR_ESHAPE;
    const int KS = K / 32;
    for (int u = 0; u < N / 32; ++u)
        for (int ks = 0; ks < KS; ++ks)
            for (int t = 0; t < 2; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e)
                        out_host[((((long)u * KS + ks) * 2 + t) * 64 + lane) * 8 + e] =
                            w_host[(long)(32 * u + 16 * t + (lane & 15)) * K + 32 * ks + 8 * (lane >> 4) + e];
    return DTLR_OK;
}

extern "C" int dtlr_dec_query_stage(const float* ref, const float* valid_ratios, const float* dim_t, const void* tgt,
                                    const void* W0, const float* b0, const void* W1, const float* b1,
                                    const void* Wqk, const float* bqk, const void* Wv, const float* bv,
                                    float* ref_in, void* qpos, void* qk, void* v, int B, int nq, int L, int dtype, void* stream)
{
    clear_stale_error();
    if (!ref || !valid_ratios || !dim_t || !tgt || !W0 || !b0 || !W1 || !b1 || !Wqk || !bqk || !Wv || !bv || !ref_in || !qpos || !qk || !v)
        return DTLR_EINVAL;
    if (B <= 0 || nq <= 0 || L <= 0) return DTLR_EINVAL;
    if (dtype != DTLR_H16) return DTLR_EDTYPE;
    const long Q = (long)B * nq;
    const long grid = (Q + DQ_ROWS - 1) / DQ_ROWS;
    if (grid > 0x7fffffffL) return DTLR_ESHAPE;
    if (grid <= 64) {
        // small batches (round 5; one line = 900 queries = 8 workgroups of 128): workgroups of 32 queries -- the four dependent GEMM stages,
        // the sine quarters and the 64 KB of output per workgroup shrink four-fold, the per-workgroup weight fetch from L2 does not grow
        static DevOnce attrs;
        if (attrs.first()) { (void)hipFuncSetAttribute((const void*)dec_query_stage_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, DqCfg<2>::LDS); (void)hipGetLastError(); }
        hipLaunchKernelGGL(dec_query_stage_kernel<2>, dim3((unsigned)((Q + 31) / 32)), dim3(512), DqCfg<2>::LDS, (hipStream_t)stream,
                           ref, valid_ratios, dim_t, (const uint16_t*)tgt, (const uint16_t*)W0, b0, (const uint16_t*)W1, b1,
                           (const uint16_t*)Wqk, bqk, (const uint16_t*)Wv, bv, ref_in, (uint16_t*)qpos, (uint16_t*)qk, (uint16_t*)v, Q, nq, L);
        return check_launch();
    }
    static DevOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute((const void*)dec_query_stage_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS); (void)hipGetLastError(); }
    hipLaunchKernelGGL(dec_query_stage_kernel<8>, dim3((unsigned)grid), dim3(512), DQ_LDS, (hipStream_t)stream,
                       ref, valid_ratios, dim_t, (const uint16_t*)tgt, (const uint16_t*)W0, b0, (const uint16_t*)W1, b1,
                       (const uint16_t*)Wqk, bqk, (const uint16_t*)Wv, bv, ref_in, (uint16_t*)qpos, (uint16_t*)qk, (uint16_t*)v, Q, nq, L);
    return check_launch();
}
