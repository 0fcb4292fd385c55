// Fused position-wise feed-forward block of a deformable-DETR layer, bf16:
//
//     Y = LayerNorm( X + relu(X W1^T + b1) W2^T + b2 )            X, Y: [M, 256]   W1: [d_ff, 256]   W2: [256, d_ff]
// (W2 is handed over packed chunk-major: W2p[c][o][k] = W2[o][32 c + k], shape [d_ff/32][256][32])
//
// == forward_ffn + norm2 of the encoder layer (models/dino/deformable_transformer.py:804-823:
// `src2 = linear2(dropout2(activation(linear1(src)))); src = norm2(src + dropout3(src2))`, dropout = identity at
// inference) and the ffn + norm3 of the decoder layer (:876-880).
//
// Why fuse: as two GEMMs the [M, d_ff] intermediate (713 MB at M = 174080, d_ff = 2048) is written to HBM and
// read back -- 1.4 GB of the block's 1.6 GB of traffic; measured 0.40 + 0.26 ms + 0.03 ms LayerNorm per encoder
// layer against an MFMA time of 0.15 ms.  Here the intermediate never leaves the CU: HBM sees X once and Y once.
//
// Structure (one workgroup = 128 tokens, 8 waves, one workgroup per CU):
//   * wave (tg, h): token group tg = 32 tokens (two 16-token MFMA column tiles), half h.
//   * X^T of the wave's tokens stays in registers for the whole kernel as MFMA B-fragments (64 VGPRs).
//   * the hidden dimension is streamed in chunks of 32 units.  A chunk's weights (32 rows of W1, 32 columns of W2,
//     32 KB) are DMA'd global -> LDS (global_load_lds_dwordx4, no VGPR staging) into a 4-stage ring, already in
//     MFMA-fragment order: every A-fragment is one linear 1 KB block (lane l reads base + 16 l: conflict-free
//     without swizzle), so the fragment layout is produced by the per-lane SOURCE addresses of the DMA.
//   * phase A: wave (tg, h) computes H^T[16 hidden of the chunk, 32 tokens] = W1c X^T (16 MFMA), adds b1, ReLU,
//     rounds to bf16 (the same rounding the unfused path applies when it stores linear1's output) and writes its
//     8 bytes per lane into the pair's H buffer -- laid out so that a lane's 16 bytes there ARE its B-fragment of
//     the second GEMM (the hidden units of a chunk are assigned to MFMA rows so that no transpose is needed).
//   * phase B: Y^T[128 channels (half h), 32 tokens] += W2c H^T (16 MFMA), fp32 accumulators for the whole kernel.
//   * ONE barrier per chunk: it publishes H(c) to the partner wave, retires chunk c+1's DMA (each wave waits for its
//     own pieces with a counted vmcnt first) and frees the ring stage the next DMA overwrites.
//   * epilogue: + b2 + X (the residual is already in registers: W2's rows are assigned to MFMA rows so that a lane's
//     accumulators are exactly the channels its X fragments hold), LayerNorm over the 256 channels (lane-local
//     sums, two cross-lane steps, one exchange with the partner wave through LDS), 16-byte stores.
#include "dtlr_common.h"
#include <stdlib.h>

namespace dtlr {

typedef __attribute__((ext_vector_type(8))) h16_hw_t ffn_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float ffn_f32x4_t;

constexpr int FFN_NS = 4;                                   // ring stages
constexpr int FFN_STAGE = 32768;                            // W1 chunk image (16 KB) | W2 chunk image (16 KB)
constexpr int FFN_H_OFF = FFN_NS * FFN_STAGE;               // H buffers: [tg 4][buf 2][tt 2][1024 B]
constexpr int FFN_LN_OFF = FFN_H_OFF + 4 * 2 * 2 * 1024;    // LayerNorm exchange: [tg 4][half 2][32 tokens] floats
constexpr int FFN_B1_OFF = FFN_LN_OFF + 4 * 256 * 4;         // (LayerNorm uses 1 KB of it, the MLP-head epilogue 4 KB)     // b1 staged once (a compiler-visible global load inside the chunk
                                                            // loop would be waited for with vmcnt(0) and drain the DMA queue)
constexpr int FFN_MAX_DFF = 2048;
constexpr int FFN_LDS = FFN_B1_OFF + FFN_MAX_DFF * 4;

// LDS-DMA: 64 lanes x 16 bytes, destination = wave-uniform LDS byte address + 16 * lane.  Invisible to hipcc's
// waitcnt bookkeeping: completion is counted by hand (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// same with a wave-uniform base in SGPRs and a 32-bit per-lane byte offset (one VGPR instead of a 64-bit address per source)
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ uint4 ffn_load16(const void* p) {
    uint4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}

template <int DBG = 0>
__device__ __forceinline__ ffn_f32x4_t ffn_mma(const uint4& a, const uint4& b, ffn_f32x4_t c) {
    if constexpr (DBG & 2) { asm volatile("" :: "v"(a.x), "v"(b.x)); return c; }
    return DTLR_MFMA_16x16x32_H16(__builtin_bit_cast(ffn_bf16x8_t, a), __builtin_bit_cast(ffn_bf16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ void unpack8(const uint4& t, float (&v)[8]) {
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = h16_lo(w[i]); v[2 * i + 1] = h16_hi(w[i]); }
}

// DBG (timing experiments only, env DTLR_FFN_DBG; results are garbage): 1 = no weight DMA after the prologue,
// 2 = no MFMA, 4 = no per-chunk barrier, 8 = no pinned interleave, 16 = no H store, 32 = no weight-fragment LDS reads after chunk 0
// (8 and up only in the instrumented build).  DBG = 0 is the product kernel; the switches are compile-time so it carries no branches.
// HEAD = true: the same two MFMA phases run a 3-layer box MLP (models/dino/dino.py MLP(256, 256, 4, 3)): no residual / LayerNorm;
// the epilogue applies ReLU to the second layer, multiplies by the 4 x 256 output layer (fp32, `gamma` = its weight, `beta` = its
// bias) and finishes with mode 0: sigmoid(delta + inverse_sigmoid(ref)) or mode 1: delta + ref -- `eps` carries the mode, Y is
// the fp32 [M, 4] result and `ref4` the fp32 [M, 4] reference boxes.  (3 launches of M = 28800 per evaluation -> 1.)
// PART = true (round 5, small M): the block runs NS-fold over the hidden dimension.  At M = 900 (ONE line: the reference's evaluation batch,
// evaluation.py:494-499) the block is 8 workgroups walking 64 chunks one after the other -- 60 us on 8 of 256 CUs, whatever the batch.
// Workgroup b = (tile b / NS, part b % NS) multiplies only the chunks [cb[part], cb[part + 1]) (the weight / bias pointers are advanced,
// the pipeline below runs unchanged on the shorter range) and stores its RAW fp32 accumulators to part `part` of a workspace
// (`Y` then points at it: [NS][M][256] floats); ffn_fused_finish_kernel adds the parts, b2 and the residual and normalises.
struct FfnParts { int ns, cb[9]; };
template <int DBG, bool HEAD = false, bool PART = false>
__global__ __launch_bounds__(512, 2) void ffn_fused_bf16_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ W1, const float* __restrict__ b1,
    const uint16_t* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, uint16_t* __restrict__ Y, int M, int d_ff, const float* __restrict__ ref4, FfnParts fpp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tg = wave >> 1, half = wave & 1;
    const int n = lane & 15, g = lane >> 4;
    const int part = PART ? (int)blockIdx.x % fpp.ns : 0;
    const int tile = PART ? (int)blockIdx.x / fpp.ns : (int)blockIdx.x;
    if constexpr (PART) {
        int c_lo = 0, c_hi = 0;
#pragma unroll
        for (int p_ = 0; p_ < 8; ++p_) if (p_ == part) { c_lo = fpp.cb[p_]; c_hi = fpp.cb[p_ + 1]; }
        W1 += (long)c_lo * 32 * 256;                            // rows 32 c_lo .. of [d_ff, 256]
        W2 += (long)c_lo * 256 * 32;                            // chunk-major [d_ff / 32][256][32]
        b1 += c_lo * 32;
        d_ff = (c_hi - c_lo) * 32;
    }
    const int nchunk = d_ff >> 5;
    const long tok0 = (long)tile * 128 + tg * 32;

    // ---- X^T fragments of this wave's 32 tokens: lane (n, g) holds X[tok][32 ks + 8 g .. +7] ----------------
    uint4 xf[8][2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const long tok = min(tok0 + tt * 16 + n, (long)M - 1);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xf[ks][tt] = ffn_load16(X + tok * 256 + ks * 32 + g * 8);
    }
    // The X loads are asm (uncounted) and waited for right here: as compiler-counted loads hipcc kept their waits (down
    // to vmcnt(0)) inside the chunk loop, where every iteration would then drain the DMA queue.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- DMA sources: wave w moves blocks 4w .. 4w+3 of a chunk image (blocks 0-15: W1, 16-31: W2) ----------
    // W1 block (j, ks): lane (m = n, g) <- W1[32 c + 8 (m>>2) + 4 j + (m&3)][32 ks + 8 g ..]   (hidden unit of MFMA row m, tile j)
    // W2 block (h, i) : lane (m = n, g) <- W2[128 h + 32 (i>>1) + 8 (m>>2) + 4 (i&1) + (m&3)][32 c + 8 g ..]  (chunk-major storage)
    const char* src[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int blk = wave * 4 + u;
        if (blk < 16) {
            const int j = blk >> 3, ks = blk & 7;
            src[u] = reinterpret_cast<const char*>(W1 + (long)(8 * (n >> 2) + 4 * j + (n & 3)) * 256 + ks * 32 + g * 8);
        } else {
            const int i = blk & 7, h = (blk >> 3) & 1;
            src[u] = reinterpret_cast<const char*>(W2 + (long)(128 * h + 32 * (i >> 1) + 8 * (n >> 2) + 4 * (i & 1) + (n & 3)) * 32 + g * 8);
        }
    }
    // bytes per chunk: 32 rows of W1 = one [256 x 32] panel of the chunk-major W2.  (W2 is passed chunk-major -- [d_ff/32][256][32]
    // -- because in the natural [256][d_ff] layout a chunk is 256 pieces of 64 B at a 4 KB stride: every CU of an XCD then reads
    // the same few L2 channels at the same time; measured 3300 cycles per chunk against 1000 of MFMA work.)
    const long cstride = 32L * 256 * 2;
    const unsigned my_blocks = lds_base + (unsigned)wave * 4096u;
#define FFN_ISSUE(C)                                                                               \
    {                                                                                              \
        const unsigned dst_ = my_blocks + (unsigned)((C) & (FFN_NS - 1)) * FFN_STAGE;              \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) glds16(src[u] + (long)(C) * cstride, dst_ + u * 1024u); \
    }

    ffn_f32x4_t yacc[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) yacc[i][tt] = ffn_f32x4_t{0.f, 0.f, 0.f, 0.f};

    {
        float* b1s = reinterpret_cast<float*>(smem + FFN_B1_OFF);
        for (int i = (int)threadIdx.x * 4; i < d_ff; i += 512 * 4) *reinterpret_cast<float4*>(b1s + i) = *reinterpret_cast<const float4*>(b1 + i);
    }
    // prologue: chunks 0, 1, 2 in flight; chunk 0 landed and visible
    FFN_ISSUE(0)
    if (nchunk > 1) FFN_ISSUE(1)
    if (nchunk > 2) FFN_ISSUE(2)
    if (nchunk > 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (nchunk > 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    unsigned char* hbuf = smem + FFN_H_OFF + tg * 4096;
    // Software pipeline of one wave (registers: wa = W1 fragments of the NEXT chunk, wb = W2 fragments of the CURRENT one):
    //     barrier(c) | read H(c) ; read wa <- W1(c+1) under phase B(c)'s 16 MFMAs ; read wb <- W2(c+1) under phase A(c+1)'s 16 MFMAs
    // Both weight images of chunk c+1 became visible at barrier(c), so neither read has to wait behind a barrier with the
    // matrix pipe idle (the first version fetched W2(c) AFTER barrier(c): all 8 waves then queued 80 ds_read_b128 on the LDS
    // port before the first MFMA of the chunk).  The interleave is pinned with sched_group_barrier: 1 LDS read per 2 MFMAs
    // is the rate at which the LDS port (8 waves x 1 KB per read) and the four matrix pipes stay equally busy.
    // phase A of chunk C: H^T[16 hidden (MFMA tile j = half), 32 tokens] = W1c X^T, + b1, ReLU, bf16 -> the pair's H buffer.
    // Four accumulation chains (even / odd k-steps x two token tiles): two chains leave the matrix pipe idle between
    // dependent MFMAs whenever the SIMD's other wave is parked at the barrier.
#define FFN_LOAD_WA(C)                                                                             \
    {                                                                                              \
        const unsigned char* w1f = smem + ((C) & (FFN_NS - 1)) * FFN_STAGE + half * 8192 + lane * 16; \
        if (!(DBG & 32) || (C) == 0) { _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) wa[ks] = *reinterpret_cast<const uint4*>(w1f + ks * 1024); } \
    }
#define FFN_LOAD_WB(C)                                                                             \
    {                                                                                              \
        const unsigned char* w2f = smem + ((C) & (FFN_NS - 1)) * FFN_STAGE + 16384 + half * 8192 + lane * 16; \
        if (!(DBG & 32) || (C) == 0) { _Pragma("unroll") for (int i = 0; i < 8; ++i) wb[i] = *reinterpret_cast<const uint4*>(w2f + i * 1024); } \
    }
#define FFN_LOAD_B1(C) const float4 bb = *reinterpret_cast<const float4*>(smem + FFN_B1_OFF + ((C) * 32 + g * 8 + half * 4) * 4);
#define FFN_MMA_A(C)                                                                               \
    {                                                                                              \
        ffn_f32x4_t he[2] = {ffn_f32x4_t{0.f, 0.f, 0.f, 0.f}, ffn_f32x4_t{0.f, 0.f, 0.f, 0.f}};    \
        ffn_f32x4_t ho[2] = {ffn_f32x4_t{0.f, 0.f, 0.f, 0.f}, ffn_f32x4_t{0.f, 0.f, 0.f, 0.f}};    \
        _Pragma("unroll") for (int ks = 0; ks < 8; ks += 2) {                                      \
            he[0] = ffn_mma<DBG>(wa[ks], xf[ks][0], he[0]);         he[1] = ffn_mma<DBG>(wa[ks], xf[ks][1], he[1]); \
            ho[0] = ffn_mma<DBG>(wa[ks + 1], xf[ks + 1][0], ho[0]); ho[1] = ffn_mma<DBG>(wa[ks + 1], xf[ks + 1][1], ho[1]); \
        }                                                                                          \
        /* lane (n, g) holds hidden units 32 C + 8 g + 4 half + r (r = 0..3) of token n: bytes [8 half, 8 half + 8) of */ \
        /* the lane's 16-byte B-fragment slot */                                                    \
        unsigned char* hw = hbuf + ((C) & 1) * 2048 + lane * 16 + half * 8;                        \
        _Pragma("unroll") for (int tt = 0; tt < 2; ++tt) {                                         \
            const float h0 = fmaxf(he[tt][0] + ho[tt][0] + bb.x, 0.f), h1 = fmaxf(he[tt][1] + ho[tt][1] + bb.y, 0.f); \
            const float h2 = fmaxf(he[tt][2] + ho[tt][2] + bb.z, 0.f), h3 = fmaxf(he[tt][3] + ho[tt][3] + bb.w, 0.f); \
            if (!(DBG & 16)) *reinterpret_cast<uint2*>(hw + tt * 1024) = make_uint2(pack_bf16x2(h0, h1), pack_bf16x2(h2, h3)); \
            else asm volatile("" :: "v"(h0), "v"(h1), "v"(h2), "v"(h3));                           \
        }                                                                                          \
    }
#define FFN_SGB(MASK, N) __builtin_amdgcn_sched_group_barrier(MASK, N, 0);

    // one chunk step.  Phase B runs ONE STEP LATE: after barrier(c) the wave multiplies chunk c-1 (H(c-1) and W2(c-1) are already
    // in registers), so the matrix pipe restarts the moment the barrier opens, while the LDS reads that had to wait for this
    // barrier -- W1(c+1) for phase A(c+1), H(c) and W2(c) for the next step's phase B -- land underneath those 16 MFMAs.
    // (Measured with the ablation switches: the MFMAs alone cost 0.19 ms and the barrier / LDS-latency / VALU skeleton 0.13 ms,
    // and in the first version -- read H(c), wait, phase B(c) -- the two simply added up.)
#define FFN_STEP(C, WITH_B, WITH_A)                                                                \
    {                                                                                              \
        /* the one barrier of the chunk: publishes H(C) to the partner wave; my pieces of chunk C+1 (read by phase */ \
        /* A(C+1)) have landed -- chunk C+2's four may stay in flight; my H writes are done */     \
        if (!(DBG & 1) && (C) + 2 < nchunk) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                           \
        if (!(DBG & 4)) __builtin_amdgcn_s_barrier();                                              \
        if (!(DBG & 1) && (C) + 3 < nchunk) FFN_ISSUE((C) + 3)   /* into the stage of chunk C-1: its last readers (W2(C-1), step C-1) are done */ \
        FFN_LOAD_B1((C) + 1)                                      /* ahead of the pinned region: it would otherwise be scheduled last, in front of the H store */ \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (WITH_A) FFN_LOAD_WA((C) + 1)                                                           \
        const uint4 hn0 = *reinterpret_cast<const uint4*>(hbuf + ((C) & 1) * 2048 + lane * 16);    \
        const uint4 hn1 = *reinterpret_cast<const uint4*>(hbuf + ((C) & 1) * 2048 + 1024 + lane * 16); \
        if (WITH_B) {                                                                              \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                        \
                yacc[i][0] = ffn_mma<DBG>(wb[i], hb0, yacc[i][0]);                                 \
                yacc[i][1] = ffn_mma<DBG>(wb[i], hb1, yacc[i][1]);                                 \
            }                                                                                      \
        }                                                                                          \
        FFN_LOAD_WB(C)                                                                             \
        if (WITH_A) FFN_MMA_A((C) + 1)                                                             \
        hb0 = hn0; hb1 = hn1;                                                                      \
        if (WITH_A && WITH_B && !(DBG & (8 | 32))) {                                               \
            FFN_SGB(0x100, 1)                                                                      \
            _Pragma("unroll") for (int r_ = 0; r_ < 9; ++r_) { FFN_SGB(0x008, 2) FFN_SGB(0x100, 1) }   /* phase B(c-1) + 2 of A | W1(c+1) x 7, H(c) x 2 */ \
            _Pragma("unroll") for (int r_ = 0; r_ < 7; ++r_) { FFN_SGB(0x008, 2) FFN_SGB(0x100, 1) }   /* phase A(c+1) | W2(c) x 7 */ \
            FFN_SGB(0x100, 1)                                                                      \
        }                                                                                          \
    }

    uint4 wa[8], wb[8];
    uint4 hb0 = make_uint4(0u, 0u, 0u, 0u), hb1 = hb0;
    {
        FFN_LOAD_B1(0)
        FFN_LOAD_WA(0)
        FFN_MMA_A(0)
    }
    FFN_STEP(0, false, true)
    for (int c = 1; c + 1 < nchunk; ++c) FFN_STEP(c, true, true)
    FFN_STEP(nchunk - 1, true, false)
#pragma unroll
    for (int i = 0; i < 8; ++i) {                               // phase B of the last chunk
        yacc[i][0] = ffn_mma<DBG>(wb[i], hb0, yacc[i][0]);
        yacc[i][1] = ffn_mma<DBG>(wb[i], hb1, yacc[i][1]);
    }
#undef FFN_SGB
#undef FFN_MMA_A
#undef FFN_LOAD_B1
#undef FFN_LOAD_WA
#undef FFN_LOAD_WB
#undef FFN_STEP
#undef FFN_ISSUE

    if constexpr (HEAD) {
        // y = relu(acc + b2) (fp32); out[o] = sum_ch y[ch] W3[o][ch] + b3[o]: lane-local over its 32 channels, then the four g
        // lanes of a token, then the two waves of the pair through LDS
        float d[2][4];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int o = 0; o < 4; ++o) d[tt][o] = 0.f;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const int ch = 128 * half + 32 * kq + 8 * g;
            const float4 ba = *reinterpret_cast<const float4*>(b2 + ch), bc = *reinterpret_cast<const float4*>(b2 + ch + 4);
            const float bias[8] = {ba.x, ba.y, ba.z, ba.w, bc.x, bc.y, bc.z, bc.w};
            float w3[4][8];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const float4 wa = *reinterpret_cast<const float4*>(gamma + o * 256 + ch), wc = *reinterpret_cast<const float4*>(gamma + o * 256 + ch + 4);
                w3[o][0] = wa.x; w3[o][1] = wa.y; w3[o][2] = wa.z; w3[o][3] = wa.w; w3[o][4] = wc.x; w3[o][5] = wc.y; w3[o][6] = wc.z; w3[o][7] = wc.w;
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = fmaxf((e < 4 ? yacc[2 * kq][tt][e] : yacc[2 * kq + 1][tt][e - 4]) + bias[e], 0.f);
#pragma unroll
                    for (int o = 0; o < 4; ++o) d[tt][o] += y * w3[o][e];
                }
        }
        float* lnx = reinterpret_cast<float*>(smem + FFN_LN_OFF) + tg * 256;      // [half][32 tokens][4]
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                d[tt][o] += __shfl_xor(d[tt][o], 16, 64);
                d[tt][o] += __shfl_xor(d[tt][o], 32, 64);
                if (g == 0) lnx[(half * 32 + tt * 16 + n) * 4 + o] = d[tt][o];
            }
        __syncthreads();
        // wave half h finishes token tile tt = h: lane (n, g) -> output component o = g of token n
        {
            const int tt = half, o = g;
            const long tok = tok0 + tt * 16 + n;
            if (tok < M) {
                const float delta = lnx[(tt * 16 + n) * 4 + o] + lnx[(32 + tt * 16 + n) * 4 + o] + beta[o];
                const float r = ref4[tok * 4 + o];
                float yv;
                if (eps == 0.f) {                               // mode 0: iterative refinement
                    const float xx = fminf(fmaxf(r, 0.f), 1.f);
                    const float x1 = fmaxf(xx, 1e-3f), x2 = fmaxf(1.f - xx, 1e-3f);
                    yv = 1.f / (1.f + expf(-(delta + logf(x1 / x2))));
                } else yv = delta + r;                          // mode 1: + proposals (unsigmoided)
                reinterpret_cast<float*>(Y)[tok * 4 + o] = yv;
            }
        }
        return;
    }
    if constexpr (PART) {
        // raw partial sums: workspace [NS][M][256] fp32; lane (n, g), kq: channels 128 half + 32 kq + 8 g + e (e < 4: yacc[2 kq], e >= 4: yacc[2 kq + 1])
        float* P = reinterpret_cast<float*>(Y) + (long)part * M * 256;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const long tok = tok0 + tt * 16 + n;
            if (tok < M) {
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    float* dst = P + tok * 256 + 128 * half + 32 * kq + 8 * g;
                    *reinterpret_cast<float4*>(dst) = make_float4(yacc[2 * kq][tt][0], yacc[2 * kq][tt][1], yacc[2 * kq][tt][2], yacc[2 * kq][tt][3]);
                    *reinterpret_cast<float4*>(dst + 4) = make_float4(yacc[2 * kq + 1][tt][0], yacc[2 * kq + 1][tt][1], yacc[2 * kq + 1][tt][2], yacc[2 * kq + 1][tt][3]);
                }
            }
        }
        return;
    }
    // ---- epilogue: + b2 + residual, LayerNorm over 256 channels, store -----------------------------------------
    // lane (n, g), ks' = 0..3: channels ch = 128 half + 32 ks' + 8 g + e, e = 0..7: e < 4 from yacc[2 ks'][tt][e],
    // e >= 4 from yacc[2 ks' + 1][tt][e - 4]; the residual is X fragment ks = 4 half + ks'.
    float v[2][4][8];
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        const int ch = 128 * half + 32 * kq + 8 * g;
        const float4 ba = *reinterpret_cast<const float4*>(b2 + ch), bc = *reinterpret_cast<const float4*>(b2 + ch + 4);
        const float bias[8] = {ba.x, ba.y, ba.z, ba.w, bc.x, bc.y, bc.z, bc.w};
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float xr[8];
            unpack8(half ? xf[4 + kq][tt] : xf[kq][tt], xr);      // (a runtime index into xf would move it to scratch)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y = e < 4 ? yacc[2 * kq][tt][e] : yacc[2 * kq + 1][tt][e - 4];
                v[tt][kq][e] = y + bias[e] + xr[e];
                s[tt] += v[tt][kq][e];
            }
        }
    }
    float* lnx = reinterpret_cast<float*>(smem + FFN_LN_OFF) + tg * 64;     // [half][32 tokens]
    float mean[2], rstd[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        s[tt] += __shfl_xor(s[tt], 16, 64);
        s[tt] += __shfl_xor(s[tt], 32, 64);
        if (g == 0) lnx[half * 32 + tt * 16 + n] = s[tt];
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) mean[tt] = (s[tt] + lnx[(half ^ 1) * 32 + tt * 16 + n]) * (1.0f / 256.0f);
    __syncthreads();
    float q[2] = {0.f, 0.f};
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[tt][kq][e] - mean[tt]; q[tt] += d * d; }
        q[tt] += __shfl_xor(q[tt], 16, 64);
        q[tt] += __shfl_xor(q[tt], 32, 64);
        if (g == 0) lnx[half * 32 + tt * 16 + n] = q[tt];
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) rstd[tt] = rsqrtf((q[tt] + lnx[(half ^ 1) * 32 + tt * 16 + n]) * (1.0f / 256.0f) + eps);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        const int ch = 128 * half + 32 * kq + 8 * g;
        const float4 ga = *reinterpret_cast<const float4*>(gamma + ch), gc = *reinterpret_cast<const float4*>(gamma + ch + 4);
        const float4 ea = *reinterpret_cast<const float4*>(beta + ch), ec = *reinterpret_cast<const float4*>(beta + ch + 4);
        const float gm[8] = {ga.x, ga.y, ga.z, ga.w, gc.x, gc.y, gc.z, gc.w};
        const float bt[8] = {ea.x, ea.y, ea.z, ea.w, ec.x, ec.y, ec.z, ec.w};
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const long tok = tok0 + tt * 16 + n;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[tt][kq][e] - mean[tt]) * rstd[tt] * gm[e] + bt[e];
            if (tok < M)
                *reinterpret_cast<uint4*>(Y + tok * 256 + ch) =
                    make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
        }
    }
}


// rows of the hidden-split form: Y[row] = LayerNorm(sum of the NS partial rows + b2 + X[row]) in the 16-bit format; one wave per row, lane l
// owns channels 4 l .. 4 l + 3; two-pass statistics (the fused epilogue's arithmetic on another reduction tree).
__global__ __launch_bounds__(256) void ffn_fused_finish_kernel(const float* __restrict__ P, int ns, long M, const uint16_t* __restrict__ X,
                                                               const float* __restrict__ b2, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, uint16_t* __restrict__ Y)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float4 v = *reinterpret_cast<const float4*>(P + row * 256 + 4 * lane);
    for (int p = 1; p < ns; ++p) {
        const float4 t = *reinterpret_cast<const float4*>(P + ((long)p * M + row) * 256 + 4 * lane);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const float4 bb = *reinterpret_cast<const float4*>(b2 + 4 * lane);
    const uint2 xw = *reinterpret_cast<const uint2*>(X + row * 256 + 4 * lane);
    v.x += bb.x + h16_lo(xw.x); v.y += bb.y + h16_hi(xw.x); v.z += bb.z + h16_lo(xw.y); v.w += bb.w + h16_hi(xw.y);
    const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    const float rstd = rsqrtf(wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / 256.0f) + eps);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * lane), be = *reinterpret_cast<const float4*>(beta + 4 * lane);
    *reinterpret_cast<uint2*>(Y + row * 256 + 4 * lane) =
        make_uint2(pack_bf16x2(dx * rstd * ga.x + be.x, dy * rstd * ga.y + be.y), pack_bf16x2(dz * rstd * ga.z + be.z, dw * rstd * ga.w + be.w));
}

// ---------------------------------------------------------------------------------------------------------------
// Fused FFN, second structure (large M: the encoder call): ONE wave per SIMD, 64 tokens per wave, 256 per workgroup.
//   * a wave computes ALL 32 hidden units of a chunk for its 64 tokens (phase A: 2 hidden tiles x 4 token tiles x 8 k-steps
//     = 64 MFMA on 16 weight fragments) -- with the hidden-unit -> MFMA-row assignment of the first structure the two
//     accumulator tiles of a token tile ARE the B-fragment of the second GEMM, so H never leaves the wave's registers:
//     no LDS exchange, no pair barrier;
//   * phase B: all 256 output channels for the 64 tokens (16 channel tiles x 4 token tiles = 64 MFMA on 16 fragments),
//     256 fp32 accumulator registers per lane (the wave owns 512 registers: one wave per SIMD);
//   * every weight fragment read from LDS feeds 4 MFMAs (2 in the first structure) and a weight byte DMA'd from L2 serves
//     256 tokens (128): both shared resources that the ablation of the first structure showed at ~60% are halved per token;
//   * phase B trails by one chunk, so the H epilogue (bias, ReLU, bf16 rounding: ~80 VALU) of chunk c overlaps the 64 MFMAs
//     of phase B(c-1);
//   * W1 and W2 stream through separate 4-stage rings (W1(c) is consumed in iteration c, W2(c) in iteration c+1), three
//     iterations of DMA lead, one barrier per chunk (DMA visibility + stage reuse only);
//   * LayerNorm statistics: a token's 256 channels live in the 4 lanes (n, g = 0..3) of ONE wave: two shuffles, no LDS.
#ifdef DTLR_GEMM_TRACE
// cycle-counter timeline of the first 8 workgroups' wave 0 (trace build only): 1 iteration top, 2 DMA landed (vmcnt), 3 barrier
// passed, 4 phase A issued, 5 iteration end; 8 kernel start, 9 main loop done, 10 kernel end
__device__ unsigned long long g_ffn_tl[8 * 1024];
#define F2_TL_INIT unsigned long long* tl_ = (threadIdx.x == 0 && blockIdx.x < 8) ? g_ffn_tl + blockIdx.x * 1024 : nullptr; int tl_n_ = 0;
#define F2_TL(CODE) { if (tl_ && tl_n_ < 1024) tl_[tl_n_++] = ((unsigned long long)__builtin_readcyclecounter() << 8) | (CODE); }
#else
#define F2_TL_INIT
#define F2_TL(CODE)
#endif
constexpr int F2_NS = 4;
constexpr int F2_RING = 16384;                              // one W1 (or W2) chunk image
constexpr int F2_W2_OFF = F2_NS * F2_RING;
constexpr int F2_B1_OFF = 2 * F2_NS * F2_RING;
constexpr int F2_PRM_OFF = F2_B1_OFF + FFN_MAX_DFF * 4;    // b2 | gamma | beta: the epilogue reads them from LDS (as global loads, with every
                                                            // VGPR occupied, hipcc issued them a few at a time: serialised L2 round trips)
constexpr int F2_LDS = F2_PRM_OFF + 3 * 256 * 4;

template <int TT>
__global__ __launch_bounds__(256, 1) void ffn2_bf16_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ W1, const float* __restrict__ b1,
    const uint16_t* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, uint16_t* __restrict__ Y, int M, int d_ff)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = lane & 15, g = lane >> 4;
    const int nchunk = d_ff >> 5;
    const long tok0 = (long)blockIdx.x * (64 * TT) + wave * (16 * TT);
    F2_TL_INIT
    F2_TL(8)

    // X^T fragments of the wave's 64 tokens: lane (n, g) holds X[tok0 + 16 tt + n][32 ks + 8 g .. +7]
    uint4 xf[8][TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const long tok = min(tok0 + tt * 16 + n, (long)M - 1);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xf[ks][tt] = ffn_load16(X + tok * 256 + ks * 32 + g * 8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // DMA sources: wave w moves blocks 4w .. 4w+3 of each 16-block image.
    // W1 block (j, ks) = 8 j + ks : lane (m, g) <- W1[32 c + 8 (m>>2) + 4 j + (m&3)][32 ks + 8 g ..]
    // W2 block i                  : lane (m, g) <- W2p[c][32 (i>>1) + 8 (m>>2) + 4 (i&1) + (m&3)][8 g ..]
    // = a wave-uniform base (SGPRs) + a per-lane byte offset (one VGPR per image).
    const unsigned v1 = (unsigned)(((8 * (n >> 2) + (n & 3)) * 256 + 8 * g) * 2);
    const unsigned v2 = (unsigned)(((8 * (n >> 2) + (n & 3)) * 32 + 8 * g) * 2);
    const long cstride = 32L * 256 * 2;
    const char* W1b = reinterpret_cast<const char*>(W1);
    const char* W2b = reinterpret_cast<const char*>(W2);
    const unsigned my1 = lds_base + (unsigned)wave * 4096u, my2 = my1 + F2_W2_OFF;
#define F2_ISSUE1(C) { const unsigned d_ = my1 + (unsigned)((C) & (F2_NS - 1)) * F2_RING;           \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                            \
            const int blk_ = wave * 4 + u;                                                         \
            glds16s(W1b + (long)(C) * cstride + ((blk_ >> 3) * 4 * 256 + (blk_ & 7) * 32) * 2, v1, d_ + u * 1024u); } }
#define F2_ISSUE2(C) { const unsigned d_ = my2 + (unsigned)((C) & (F2_NS - 1)) * F2_RING;           \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                            \
            const int blk_ = wave * 4 + u;                                                         \
            glds16s(W2b + (long)(C) * cstride + ((32 * (blk_ >> 1) + 4 * (blk_ & 1)) * 32) * 2, v2, d_ + u * 1024u); } }
    // one 1 KB piece at a time: inside the main loop the pieces are spread between MFMA groups (a burst of 8 right behind the barrier
    // stalls the issuing wave for ~150 cycles per piece -- with one wave per SIMD that is matrix-pipe idle time)
#define F2_PIECE1(C, U) { const int blk_ = wave * 4 + (U);                                          \
        glds16s(W1b + (long)(C) * cstride + ((blk_ >> 3) * 4 * 256 + (blk_ & 7) * 32) * 2, v1,     \
                my1 + (unsigned)((C) & (F2_NS - 1)) * F2_RING + (U) * 1024u); }
#define F2_PIECE2(C, U) { const int blk_ = wave * 4 + (U);                                          \
        glds16s(W2b + (long)(C) * cstride + ((32 * (blk_ >> 1) + 4 * (blk_ & 1)) * 32) * 2, v2,    \
                my2 + (unsigned)((C) & (F2_NS - 1)) * F2_RING + (U) * 1024u); }
    {
        float* b1s = reinterpret_cast<float*>(smem + F2_B1_OFF);
        for (int i = (int)threadIdx.x * 4; i < d_ff; i += 256 * 4) *reinterpret_cast<float4*>(b1s + i) = *reinterpret_cast<const float4*>(b1 + i);
        float* prm = reinterpret_cast<float*>(smem + F2_PRM_OFF);
        prm[threadIdx.x] = b2[threadIdx.x];
        prm[256 + threadIdx.x] = gamma[threadIdx.x];
        prm[512 + threadIdx.x] = beta[threadIdx.x];
    }
    // issue order = the steady state's (iteration c issues W1(c+3), W2(c+2)): W1(0) | W1(1) W2(0) | W1(2) W2(1)
    F2_ISSUE1(0)
    if (nchunk > 1) F2_ISSUE1(1)
    F2_ISSUE2(0)
    if (nchunk > 2) F2_ISSUE1(2)
    if (nchunk > 1) F2_ISSUE2(1)

    ffn_f32x4_t yacc[16][TT];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) yacc[i][tt] = ffn_f32x4_t{0.f, 0.f, 0.f, 0.f};
    uint4 hb[TT];                                                   // H^T B-fragments of the chunk phase B works on
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) hb[tt] = make_uint4(0u, 0u, 0u, 0u);
    // ONE 16-fragment register buffer serves both GEMMs: as phase A consumes W1 fragment q the slot is refilled with W2 fragment q
    // of the chunk phase B works on, and as phase B consumes that, with W1 fragment q of the next chunk -- every LDS read has a
    // whole phase (16 TT MFMAs) to land, and nothing is read right behind a barrier.  slot q: W1 block (j = q & 1, ks = q >> 1) /
    // W2 block i = q.
    uint4 w[16];
#define F2_W1F(C, Q) (*reinterpret_cast<const uint4*>(smem + ((C) & (F2_NS - 1)) * F2_RING + (((Q) & 1) * 8 + ((Q) >> 1)) * 1024 + lane * 16))
#define F2_W2F(C, Q) (*reinterpret_cast<const uint4*>(smem + F2_W2_OFF + ((C) & (F2_NS - 1)) * F2_RING + (Q) * 1024 + lane * 16))
    // W1(0) has to be in registers before the first iteration (the only exposed fragment reads are these and the last W2's)
    if (nchunk > 1) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) w[q] = F2_W1F(0, q);

    // iteration C (after barrier C: W1(C+1) and W2(C) are visible; w = W1(C) is already in registers):
    //     phase A(C)  : 16 x TT MFMA; slot q <- W2(C-1) fragment q as soon as its MFMAs are issued
    //     phase B(C-1): 16 x TT MFMA; slot q <- W1(C+1) fragment q; the H epilogue of chunk C (VALU) runs underneath
#define F2_STEP(C, WITH_B, WITH_NEXT, DMA1, DMA2)                                                  \
    {                                                                                              \
        F2_TL(1)                                                                                   \
        if (DMA2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   /* the previous iteration's 8 pieces may stay in flight */ \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                           \
        F2_TL(2)                                                                                   \
        __builtin_amdgcn_s_barrier();                                                              \
        F2_TL(3)                                                                                   \
        const float4 bl = *reinterpret_cast<const float4*>(smem + F2_B1_OFF + ((C) * 32 + g * 8) * 4);      \
        const float4 bh = *reinterpret_cast<const float4*>(smem + F2_B1_OFF + ((C) * 32 + g * 8 + 4) * 4);  \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        ffn_f32x4_t he[2][TT];                                                                     \
        /* hand-placed stream: one group = TT MFMAs on fragment slot q + the slot's refill (+ a DMA piece every 4th group, + a */ \
        /* sixth of the H epilogue in phase B); sched_barrier(0) between groups keeps hipcc from re-clustering them */ \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                           \
            _Pragma("unroll") for (int tt = 0; tt < TT; ++tt)                                      \
                he[q & 1][tt] = ffn_mma<0>(w[q], xf[q >> 1][tt], q < 2 ? ffn_f32x4_t{0.f, 0.f, 0.f, 0.f} : he[q & 1][tt]); \
            if (WITH_B) w[q] = F2_W2F((C) - 1, q);                                                 \
            else if (WITH_NEXT) w[q] = F2_W1F((C) + 1, q);                                         \
            if ((q & 3) == 3 && (DMA1)) F2_PIECE1((C) + 3, q >> 2)                                 \
            if (!(WITH_B) && (q & 3) == 1 && (DMA2)) F2_PIECE2((C) + 2, q >> 2)                    \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
        /* H epilogue unit (j, tt): lane (n, g) holds hidden units 32 C + 8 g + 4 j + r of token n = k-slots 8 g + 4 j + r of the B-fragment */ \
        uint32_t hp[TT][4];                                                                        \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                           \
            if (WITH_B) {                                                                          \
                _Pragma("unroll") for (int tt = 0; tt < TT; ++tt) yacc[q][tt] = ffn_mma<0>(w[q], hb[tt], yacc[q][tt]); \
                if (WITH_NEXT) w[q] = F2_W1F((C) + 1, q);                                          \
                if ((q & 3) == 3 && (DMA2)) F2_PIECE2((C) + 2, q >> 2)                             \
            }                                                                                      \
            if (q < 2 * TT) {                                                                      \
                const int j_ = q & 1, t_ = q >> 1;                                                 \
                const float4 bb_ = j_ ? bh : bl;                                                   \
                hp[t_][2 * j_] = pack_bf16x2(fmaxf(he[j_][t_][0] + bb_.x, 0.f), fmaxf(he[j_][t_][1] + bb_.y, 0.f));     \
                hp[t_][2 * j_ + 1] = pack_bf16x2(fmaxf(he[j_][t_][2] + bb_.z, 0.f), fmaxf(he[j_][t_][3] + bb_.w, 0.f)); \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
        _Pragma("unroll") for (int tt = 0; tt < TT; ++tt) hb[tt] = make_uint4(hp[tt][0], hp[tt][1], hp[tt][2], hp[tt][3]); \
        F2_TL(5)                                                                                   \
    }

    // nchunk >= 4 (launch check).  The DMA flags are compile-time so that the steady-state iteration is ONE basic block (a runtime
    // `c + 3 < nchunk` around each piece split it into nine and un-did the MFMA / LDS / VALU interleave).
    F2_STEP(0, false, true, true, true)
    for (int c = 1; c + 3 < nchunk; ++c) F2_STEP(c, true, true, true, true)
    F2_STEP(nchunk - 3, true, true, false, true)
    F2_STEP(nchunk - 2, true, true, false, false)
    F2_STEP(nchunk - 1, true, false, false, false)
#undef F2_STEP
#undef F2_ISSUE1
#undef F2_ISSUE2
#undef F2_PIECE1
#undef F2_PIECE2
    F2_TL(9)
    // phase B of the last chunk (its W2 image was published by the last barrier)
#pragma unroll
    for (int q = 0; q < 16; ++q) w[q] = F2_W2F(nchunk - 1, q);
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) yacc[q][tt] = ffn_mma<0>(w[q], hb[tt], yacc[q][tt]);
#undef F2_W1F
#undef F2_W2F

    // ---- epilogue: + b2 + residual (X fragment ks holds exactly the channels of accumulator tiles 2 ks, 2 ks + 1), LayerNorm, store ----
    const float* prm2_ = reinterpret_cast<const float*>(smem + F2_PRM_OFF);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const long tok = tok0 + tt * 16 + n;
        float v[8][8];
        float sum = 0.f;
#pragma unroll
        for (int kq = 0; kq < 8; ++kq) {
            const int ch = 32 * kq + 8 * g;
            const float4 ba = *reinterpret_cast<const float4*>(prm2_ + ch), bc = *reinterpret_cast<const float4*>(prm2_ + ch + 4);
            const float bias[8] = {ba.x, ba.y, ba.z, ba.w, bc.x, bc.y, bc.z, bc.w};
            float xr[8];
            unpack8(xf[kq][tt], xr);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y = e < 4 ? yacc[2 * kq][tt][e] : yacc[2 * kq + 1][tt][e - 4];
                v[kq][e] = y + bias[e] + xr[e];
                sum += v[kq][e];
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / 256.0f);
        float q = 0.f;
#pragma unroll
        for (int kq = 0; kq < 8; ++kq)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[kq][e] - mean; q += d * d; }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q * (1.0f / 256.0f) + eps);
#pragma unroll
        for (int kq = 0; kq < 8; ++kq) {
            const int ch = 32 * kq + 8 * g;
            const float4 ga = *reinterpret_cast<const float4*>(prm2_ + 256 + ch), gc = *reinterpret_cast<const float4*>(prm2_ + 256 + ch + 4);
            const float4 ea = *reinterpret_cast<const float4*>(prm2_ + 512 + ch), ec = *reinterpret_cast<const float4*>(prm2_ + 512 + ch + 4);
            const float gm[8] = {ga.x, ga.y, ga.z, ga.w, gc.x, gc.y, gc.z, gc.w};
            const float bt[8] = {ea.x, ea.y, ea.z, ea.w, ec.x, ec.y, ec.z, ec.w};
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[kq][e] - mean) * rstd * gm[e] + bt[e];
            if (tok < M)
                *reinterpret_cast<uint4*>(Y + tok * 256 + ch) =
                    make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
        }
    }
    F2_TL(10)
}

// ---------------------------------------------------------------------------------------------------------------
// Output projection + residual + LayerNorm of an attention block, bf16:   Y = LayerNorm(R + A W^T + b),  all [M, 256]
// == `src = norm1(src + dropout1(self_attn(...)))` where the last op of self_attn is `output_proj`
// (models/dino/deformable_transformer.py:810-815; ops/modules/ms_deform_attn.py:124), and the decoder's
// `tgt = norm2(tgt + self_attn.out_proj(...))` / `tgt = norm1(tgt + cross_attn.output_proj(...))` (:847-870).
// As GEMM + LayerNorm the projected rows make an HBM round trip (178 MB per encoder call); here they stay in the
// accumulators.  Same building blocks as the fused FFN above: A^T of 32 tokens per wave in registers as B-fragments, the
// whole 256x256 weight (128 KB) DMA'd into LDS in fragment order in four k-groups with counted waits, W's rows assigned
// to MFMA rows so that a lane's accumulators are 8-channel runs (16-byte residual loads and stores), LayerNorm with one
// exchange between the two waves of a token group.
// Workgroup = 64 tokens, 4 waves (token group tg, channel half h); the weight streams through a 2-stage ring of k-groups
// (2 x 32 KB) so that TWO workgroups fit a CU and one's load phase overlaps the other's MFMA phase.  (First version: 128
// tokens, 8 waves, the whole 128 KB weight resident -> one workgroup per CU, its phases strictly serial: 107 us per encoder
// call against 60 us of HBM time.)
constexpr int PLN_LN_OFF = 64 * 1024;
constexpr int PLN_LDS = PLN_LN_OFF + 2 * 2 * 32 * 4;

// SPLIT = true (two-stage query selection, deformable_transformer.py:320-345: `output_memory = enc_output_norm(enc_output(memory
// masked))`): no residual; `R` is instead a per-token keep mask (uint8, 0 = the token's features are zeroed BEFORE the
// projection, models/dino/utils.py:60-62), and the fp32 LayerNorm result is written as THREE bf16 images per token,
// Y[tok] = [hi | lo | hi] with hi = bf16(y), lo = bf16(y - hi): multiplied with a class-head weight laid out as
// [W_hi | W_hi | W_lo] the bf16 matrix cores deliver y W^T to ~2^-16 relative -- the selection scores no longer need the
// fp32 MFMA path (0.27 ms for the 166 x 256 head over 174080 tokens) nor an fp32 copy of output_memory.
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void proj_ln_bf16_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, const float* __restrict__ bias,
    const uint16_t* __restrict__ R, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    uint16_t* __restrict__ Y, int M)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tg = wave >> 1, half = wave & 1;
    const int n = lane & 15, g = lane >> 4;
    const long tok0 = (long)blockIdx.x * 64 + tg * 32;

    uint4 af[8][2], rr[4][2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const long tok = min(tok0 + tt * 16 + n, (long)M - 1);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) af[ks][tt] = ffn_load16(A + tok * 256 + ks * 32 + g * 8);
        if constexpr (!SPLIT) {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) rr[kq][tt] = ffn_load16(R + tok * 256 + 128 * half + 32 * kq + 8 * g);
        } else {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) rr[kq][tt] = make_uint4(0u, 0u, 0u, 0u);
            if (R && reinterpret_cast<const unsigned char*>(R)[tok] == 0) {        // masked token: features zeroed before the projection
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) af[ks][tt] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    // weight DMA: k-group j = k-steps {2j, 2j+1} = 32 blocks of 1 KB, 8 per wave: block (i = 4 w + (u>>1), kk = u&1) of the
    // group goes to ring stage j&1 at ((i*2 + kk) KB).  W is pre-packed in image order (dtlr_proj_pack_weights): block
    // (i, ks) is the contiguous KB number i*8 + ks -- every DMA instruction reads 1 KB of contiguous memory.
#define PLN_ISSUE(J)                                                                               \
    {                                                                                              \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                            \
            const int i = 4 * wave + (u >> 1), kk = u & 1;                                         \
            glds16(W + ((long)(i * 8 + 2 * (J) + kk) * 64 + lane) * 8,                             \
                   lds_base + (unsigned)(((J) & 1) * 32768 + (i * 2 + kk) * 1024));               \
        }                                                                                          \
    }
    PLN_ISSUE(0)
    ffn_f32x4_t yacc[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) yacc[i][tt] = ffn_f32x4_t{0.f, 0.f, 0.f, 0.f};
#define PLN_GROUP(J)                                                                               \
    {                                                                                              \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* A, R and my pieces of group J landed */ \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        __builtin_amdgcn_s_barrier();                      /* ... everyone's; and everyone is done reading the other stage */ \
        if ((J) + 1 < 4) PLN_ISSUE((J) + 1)                                                        \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                         \
            const int ks = 2 * (J) + kk;                                                           \
            uint4 wf[8];                                                                           \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                                          \
                wf[i] = *reinterpret_cast<const uint4*>(smem + ((J) & 1) * 32768 + ((8 * half + i) * 2 + kk) * 1024 + lane * 16); \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                        \
                yacc[i][0] = ffn_mma<0>(wf[i], af[ks][0], yacc[i][0]);                             \
                yacc[i][1] = ffn_mma<0>(wf[i], af[ks][1], yacc[i][1]);                             \
            }                                                                                      \
        }                                                                                          \
    }
    PLN_GROUP(0) PLN_GROUP(1) PLN_GROUP(2) PLN_GROUP(3)
#undef PLN_GROUP
#undef PLN_ISSUE

    // ---- epilogue: + bias + residual, LayerNorm over 256 channels, store (as in the fused FFN) -------------------------
    float v[2][4][8];
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        const int ch = 128 * half + 32 * kq + 8 * g;
        const float4 ba = *reinterpret_cast<const float4*>(bias + ch), bc = *reinterpret_cast<const float4*>(bias + ch + 4);
        const float bs[8] = {ba.x, ba.y, ba.z, ba.w, bc.x, bc.y, bc.z, bc.w};
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float xr[8];
            unpack8(rr[kq][tt], xr);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y = e < 4 ? yacc[2 * kq][tt][e] : yacc[2 * kq + 1][tt][e - 4];
                v[tt][kq][e] = y + bs[e] + xr[e];
                s[tt] += v[tt][kq][e];
            }
        }
    }
    float* lnx = reinterpret_cast<float*>(smem + PLN_LN_OFF) + tg * 64;
    float mean[2], rstd[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        s[tt] += __shfl_xor(s[tt], 16, 64);
        s[tt] += __shfl_xor(s[tt], 32, 64);
        if (g == 0) lnx[half * 32 + tt * 16 + n] = s[tt];
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) mean[tt] = (s[tt] + lnx[(half ^ 1) * 32 + tt * 16 + n]) * (1.0f / 256.0f);
    __syncthreads();
    float q[2] = {0.f, 0.f};
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[tt][kq][e] - mean[tt]; q[tt] += d * d; }
        q[tt] += __shfl_xor(q[tt], 16, 64);
        q[tt] += __shfl_xor(q[tt], 32, 64);
        if (g == 0) lnx[half * 32 + tt * 16 + n] = q[tt];
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) rstd[tt] = rsqrtf((q[tt] + lnx[(half ^ 1) * 32 + tt * 16 + n]) * (1.0f / 256.0f) + eps);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        const int ch = 128 * half + 32 * kq + 8 * g;
        const float4 ga = *reinterpret_cast<const float4*>(gamma + ch), gc = *reinterpret_cast<const float4*>(gamma + ch + 4);
        const float4 ea = *reinterpret_cast<const float4*>(beta + ch), ec = *reinterpret_cast<const float4*>(beta + ch + 4);
        const float gm[8] = {ga.x, ga.y, ga.z, ga.w, gc.x, gc.y, gc.z, gc.w};
        const float bt[8] = {ea.x, ea.y, ea.z, ea.w, ec.x, ec.y, ec.z, ec.w};
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const long tok = tok0 + tt * 16 + n;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[tt][kq][e] - mean[tt]) * rstd[tt] * gm[e] + bt[e];
            const uint4 hi = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
            if constexpr (!SPLIT) {
                if (tok < M) *reinterpret_cast<uint4*>(Y + tok * 256 + ch) = hi;
            } else {
                float hf[8], l[8];
                unpack8(hi, hf);
#pragma unroll
                for (int e = 0; e < 8; ++e) l[e] = o[e] - hf[e];
                const uint4 lo = make_uint4(pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3]), pack_bf16x2(l[4], l[5]), pack_bf16x2(l[6], l[7]));
                if (tok < M) {
                    *reinterpret_cast<uint4*>(Y + tok * 768 + ch) = hi;
                    *reinterpret_cast<uint4*>(Y + tok * 768 + 256 + ch) = lo;
                    *reinterpret_cast<uint4*>(Y + tok * 768 + 512 + ch) = hi;
                }
            }
        }
    }
}

}  // namespace dtlr

using namespace dtlr;

extern "C" int dtlr_ffn_fused_bf16(const void* X, const void* W1, const float* b1, const void* W2, const float* b2,
                                   const float* gamma, const float* beta, float eps, void* Y,
                                   int M, int d_model, int d_ff, void* stream)
{
    clear_stale_error();
    if (!X || !W1 || !b1 || !W2 || !b2 || !gamma || !beta || !Y) return DTLR_EINVAL;
    if (M <= 0 || d_ff <= 0) return DTLR_EINVAL;
    if (d_model != 256 || (d_ff & 31) || d_ff < 64 || d_ff > FFN_MAX_DFF) return DTLR_ESHAPE;     // >= 2 chunks: phase B trails by one
    static const int dbg = exp_env_int("DTLR_FFN_DBG", 0);          // experiment builds only: ablated variants (results are garbage)
    // Second structure (one wave per SIMD, H in registers; see ffn2_bf16_kernel) whenever it has its >= 4 chunks:
    //   * whole rounds of 256 workgroups x 192 tokens (TT = 3), then the remainder as 128-token workgroups (TT = 2) when those
    //     fit one round -- at M = 174080 that is 3 full rounds + 208 workgroups of 2/3 the length instead of a fourth round
    //     that would be 54% full;
    //   * M <= 32768 (the decoder call, M = 28800: a single partial round) stays on the first structure, which is as fast there.
    // experiment builds: DTLR_FFN_V = 1 forces the first structure, 2 the second with TT = 3 only (measurements / tests).
    static const int ver = exp_env_int("DTLR_FFN_V", 0);               // experiment builds only
    if (dbg == 0 && d_ff >= 128 && ver != 1 && (ver == 2 || M > 256 * 128)) {          // one partial round or less: the first structure is as fast
        static DevOnce attr2;
        if (attr2.first()) {
            (void)hipFuncSetAttribute((const void*)ffn2_bf16_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, F2_LDS);
            (void)hipFuncSetAttribute((const void*)ffn2_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, F2_LDS);
            (void)hipGetLastError();
           
        }
        const uint16_t* Xp = (const uint16_t*)X;
        uint16_t* Yp = (uint16_t*)Y;
        const int NCU = 256;
        long done = 0;
        if (ver == 2) {
            hipLaunchKernelGGL(ffn2_bf16_kernel<3>, dim3((unsigned)((M + 191) / 192)), dim3(256), F2_LDS, (hipStream_t)stream,
                               Xp, (const uint16_t*)W1, b1, (const uint16_t*)W2, b2, gamma, beta, eps, Yp, M, d_ff);
            return check_launch();
        }
        const long full = ((long)M / 192 / NCU) * NCU;                    // workgroups in whole rounds
        if (full > 0) {
            hipLaunchKernelGGL(ffn2_bf16_kernel<3>, dim3((unsigned)full), dim3(256), F2_LDS, (hipStream_t)stream,
                               Xp, (const uint16_t*)W1, b1, (const uint16_t*)W2, b2, gamma, beta, eps, Yp, (int)(full * 192), d_ff);
            done = full * 192;
        }
        const long rem = M - done;
        if (rem > 0) {
            const uint16_t* Xr = Xp + done * 256;
            uint16_t* Yr = Yp + done * 256;
            if (rem <= (long)NCU * 128)
                hipLaunchKernelGGL(ffn2_bf16_kernel<2>, dim3((unsigned)((rem + 127) / 128)), dim3(256), F2_LDS, (hipStream_t)stream,
                                   Xr, (const uint16_t*)W1, b1, (const uint16_t*)W2, b2, gamma, beta, eps, Yr, (int)rem, d_ff);
            else
                hipLaunchKernelGGL(ffn2_bf16_kernel<3>, dim3((unsigned)((rem + 191) / 192)), dim3(256), F2_LDS, (hipStream_t)stream,
                                   Xr, (const uint16_t*)W1, b1, (const uint16_t*)W2, b2, gamma, beta, eps, Yr, (int)rem, d_ff);
        }
        return check_launch();
    }
    const unsigned grid = (unsigned)((M + 127) / 128);
    // Hidden split (round 5): when the 128-token tiles fill less than half of the chip -- one to a few lines: the latency case -- run
    // every tile NS-fold over the hidden dimension (>= 8 chunks per part) into the stream's workspace and finish with one small kernel.
    if (dbg == 0 && grid <= 96 && d_ff >= 512) {
        const int nchunk = d_ff >> 5;
        int ns = (int)(256 / grid);
        if (ns > 8) ns = 8;
        if (ns > nchunk / 8) ns = nchunk / 8;
        float* ws = ns >= 2 ? stream_workspace((size_t)ns * M * 256 * sizeof(float), (hipStream_t)stream) : nullptr;
        if (ws) {
            FfnParts fp{};
            fp.ns = ns;
            for (int p = 0; p <= ns; ++p) fp.cb[p] = (int)((long)nchunk * p / ns);
            static DevOnce attrp;
            if (attrp.first()) { (void)hipFuncSetAttribute((const void*)ffn_fused_bf16_kernel<0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FFN_LDS); (void)hipGetLastError(); }
            hipLaunchKernelGGL((ffn_fused_bf16_kernel<0, false, true>), dim3(grid * (unsigned)ns), dim3(512), FFN_LDS, (hipStream_t)stream,
                               (const uint16_t*)X, (const uint16_t*)W1, b1, (const uint16_t*)W2, b2, gamma, beta, eps, reinterpret_cast<uint16_t*>(ws), M, d_ff,
                               (const float*)nullptr, fp);
            hipLaunchKernelGGL(ffn_fused_finish_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                               (const float*)ws, ns, (long)M, (const uint16_t*)X, b2, gamma, beta, eps, (uint16_t*)Y);
            return check_launch();
        }
    }
#define FFN_LAUNCH(D)                                                                              \
    {                                                                                              \
        static DevOnce attr;                                                                  \
        if (attr.first()) { (void)hipFuncSetAttribute((const void*)ffn_fused_bf16_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, FFN_LDS); (void)hipGetLastError(); } \
        hipLaunchKernelGGL(ffn_fused_bf16_kernel<D>, dim3(grid), dim3(512), FFN_LDS, (hipStream_t)stream, \
                           (const uint16_t*)X, (const uint16_t*)W1, b1, (const uint16_t*)W2, b2, gamma, beta, eps, (uint16_t*)Y, M, d_ff, (const float*)nullptr, FfnParts{}); \
    }
    switch (dbg) {
    case 1: FFN_LAUNCH(1) break;
    case 2: FFN_LAUNCH(2) break;
    case 3: FFN_LAUNCH(3) break;
    case 4: FFN_LAUNCH(4) break;
    case 6: FFN_LAUNCH(6) break;
    case 7: FFN_LAUNCH(7) break;
#ifdef DTLR_GEMM_ABLATION
    case 8: FFN_LAUNCH(8) break;
    case 16: FFN_LAUNCH(16) break;
    case 32: FFN_LAUNCH(32) break;
    case 33: FFN_LAUNCH(33) break;
    case 34: FFN_LAUNCH(34) break;
    case 48: FFN_LAUNCH(48) break;
    case 49: FFN_LAUNCH(49) break;
    case 51: FFN_LAUNCH(51) break;
    case 55: FFN_LAUNCH(55) break;
#endif
    default: FFN_LAUNCH(0) break;
    }
#undef FFN_LAUNCH
    return check_launch();
}

extern "C" int dtlr_box_mlp_refine_bf16(const void* X, const void* W1, const float* b1, const void* W2p, const float* b2,
                                        const float* W3, const float* b3, const float* ref, float* out, int M, int mode, void* stream)
{
    clear_stale_error();
    if (!X || !W1 || !b1 || !W2p || !b2 || !W3 || !b3 || !ref || !out) return DTLR_EINVAL;
    if (M <= 0 || (mode != 0 && mode != 1)) return DTLR_EINVAL;
    static DevOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute((const void*)ffn_fused_bf16_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FFN_LDS); (void)hipGetLastError(); }
    hipLaunchKernelGGL((ffn_fused_bf16_kernel<0, true>), dim3((unsigned)((M + 127) / 128)), dim3(512), FFN_LDS, (hipStream_t)stream,
                       (const uint16_t*)X, (const uint16_t*)W1, b1, (const uint16_t*)W2p, b2, W3, b3, (float)mode, (uint16_t*)out, M, 256, ref, FfnParts{});
    return check_launch();
}

// W [256, 256] row-major bf16 (host) -> the fragment-major image the kernel streams: block (i, ks) = 64 lanes x 8 elements,
// lane (m = lane & 15, g = lane >> 4) <- W[sigma(i, m)][32 ks + 8 g ..], sigma(i, m) = 128 (i>>3) + 32 ((i&7)>>1) + 8 (m>>2) + 4 (i&1) + (m&3)
extern "C" int dtlr_proj_pack_weights(const unsigned short* w_host, unsigned short* wp_host)
{
    if (!w_host || !wp_host) return DTLR_EINVAL;
    for (int i = 0; i < 16; ++i)
        for (int ks = 0; ks < 8; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = lane & 15, g = lane >> 4;
                const int row = 128 * (i >> 3) + 32 * ((i & 7) >> 1) + 8 * (m >> 2) + 4 * (i & 1) + (m & 3);
                for (int e = 0; e < 8; ++e) wp_host[((i * 8 + ks) * 64 + lane) * 8 + e] = w_host[row * 256 + ks * 32 + g * 8 + e];
            }
    return DTLR_OK;
}

extern "C" int dtlr_proj_ln_bf16(const void* A, const void* W, const float* bias, const void* R,
                                 const float* gamma, const float* beta, float eps, void* Y, int M, int d_model, void* stream)
{
    clear_stale_error();
    if (!A || !W || !bias || !R || !gamma || !beta || !Y) return DTLR_EINVAL;
    if (M <= 0) return DTLR_EINVAL;
    if (d_model != 256) return DTLR_ESHAPE;
    static DevOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute((const void*)proj_ln_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PLN_LDS); (void)hipGetLastError(); }
    hipLaunchKernelGGL(proj_ln_bf16_kernel<false>, dim3((unsigned)((M + 63) / 64)), dim3(256), PLN_LDS, (hipStream_t)stream,
                       (const uint16_t*)A, (const uint16_t*)W, bias, (const uint16_t*)R, gamma, beta, eps, (uint16_t*)Y, M);
    return check_launch();
}

extern "C" int dtlr_proj_ln_split_bf16(const void* A, const void* W, const float* bias, const unsigned char* keep,
                                       const float* gamma, const float* beta, float eps, void* Y3, int M, int d_model, void* stream)
{
    clear_stale_error();
    if (!A || !W || !bias || !gamma || !beta || !Y3) return DTLR_EINVAL;
    if (M <= 0) return DTLR_EINVAL;
    if (d_model != 256) return DTLR_ESHAPE;
    static DevOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute((const void*)proj_ln_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PLN_LDS); (void)hipGetLastError(); }
    hipLaunchKernelGGL(proj_ln_bf16_kernel<true>, dim3((unsigned)((M + 63) / 64)), dim3(256), PLN_LDS, (hipStream_t)stream,
                       (const uint16_t*)A, (const uint16_t*)W, bias, reinterpret_cast<const uint16_t*>(keep), gamma, beta, eps, (uint16_t*)Y3, M);
    return check_launch();
}

#ifdef DTLR_GEMM_TRACE
extern "C" int dtlr_debug_ffn_trace(unsigned long long* out, int clear_only)
{
    if (hipDeviceSynchronize() != hipSuccess) return DTLR_ELAUNCH;
    const size_t bytes = sizeof(unsigned long long) * 8 * 1024;
    if (!clear_only && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ffn_tl), bytes) != hipSuccess) return DTLR_ELAUNCH;
    void* dptr = nullptr;
    if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(g_ffn_tl)) != hipSuccess) return DTLR_ELAUNCH;
    if (hipMemset(dptr, 0, bytes) != hipSuccess) return DTLR_ELAUNCH;
    return DTLR_OK;
}
#endif
