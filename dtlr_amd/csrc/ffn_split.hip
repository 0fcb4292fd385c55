// Fused position-wise feed-forward block of the SPLIT-fp32 engine (round 4; dtype DTLR_F32S):
//
//     Y = LayerNorm( X + relu(X W1^T + b1) W2^T + b2 )        X, Y: [M, 256] fp32   W1: [d_ff, 256]   W2: [256, d_ff]
// == forward_ffn + norm2 / norm3 (models/dino/deformable_transformer.py:804-823, 876-880), every product as three fp16 MFMAs on
// hi + lo halves (gemm.hip, GT<f32s_t>), fp32 accumulation, fp32 residual and LayerNorm.
//
// Why: as two tiled split GEMMs + LayerNorm the encoder FFN was 7.7 of the split engine's 27.9 ms per step (linear1 0.70 ms, linear2 0.59 ms per
// layer at B = 32): both are bound by L2 -> LDS operand delivery at 128 x 128 tiles (fp32-sized operands: 5.6 GB per GEMM at ~9 TB/s), and the
// [M, d_ff] fp32 intermediate (1.43 GB) is written and read back.  Here the dataflow of ffn32.hip (ffn3_bf16_kernel):
//   * a workgroup = 4 waves (one per SIMD) owns 128 tokens, a wave 32 of them.  The wave keeps X^T of its tokens as B-fragments of the
//     32x32x16 MFMA in registers, SPLIT once: xh[s], xl[s] (k-step s = 16 k; lane (j = token, hh): k = 16 s + 8 hh .. + 7) -- 128 VGPRs --
//     and the Y^T accumulators of all 256 channels (8 tiles x 16 registers = 128, AGPRs);
//   * the hidden dimension is walked in chunks of 32 units.  Phase A: H^T[32, 32 tokens] = W1c X^T, 16 k-steps x 3 products (W_hi x_lo,
//     W_lo x_hi, W_hi x_hi) = 48 MFMAs.  + b1, ReLU in fp32, then H is split IN REGISTERS: with the hidden units ordered inside W2's
//     k-steps as in ffn32.hip the accumulator registers 8 s .. 8 s + 7 ARE the lane's B-fragment of k-step s, so hh[s] / hl[s] are two
//     conversions of registers the lane already holds.  Phase B: Y^T[256, 32 tokens] += W2c H^T, 8 channel tiles x 2 k-steps x 3 = 48 MFMAs;
//   * the four weight images of a chunk (W1_hi, W1_lo, W2_hi, W2_lo: 16 fragments of 1 KB each, fragment order, packed by
//     ops.ffn_split_pack) are one linear 64 KB block per chunk, DMA'd global -> LDS (global_load_lds_dwordx4, 16 pieces per wave and chunk) into
//     two two-stage rings (W1 / W2); phase B trails by one chunk, so W1(c + 2) and W2(c + 1) stream in while W1(c + 1) and W2(c) are
//     multiplied; one barrier per chunk.  A weight byte serves 128 tokens: 1360 workgroups x 4 MB = 5.4 GB of L2 -> LDS traffic per
//     encoder call against 11.2 GB for the two tiled GEMMs, and no intermediate in HBM;
//   * epilogue: + b2 + X (the residual re-read in fp32: it must not be rounded), LayerNorm with two-pass statistics (one exchange with
//     lane ^ 32 each), fp32 stores of 16 bytes per lane.
// 96 MFMAs of 32 cycles per chunk and wave = 3072 matrix cycles against 64 fragment reads (64 KB per wave, 256 KB per CU: 2048 LDS
// cycles) and 64 KB of DMA writes: the matrix pipe is the bound, the LDS port follows at ~80%.
#include "dtlr_common.h"

namespace dtlr {

typedef __attribute__((ext_vector_type(8))) _Float16 fs_f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 fs_f16x2_t;
typedef __attribute__((ext_vector_type(16))) float fs_f32x16_t;

// experiment hooks (tools/experiments/ffn_split_bench.py builds this file alone with -DFS_STANDALONE and these): fragment look-ahead in groups,
// and timing ablations whose results are garbage (1: no weight DMA in the chunk loop, 2: no wait / barrier per chunk, 4: no fragment reads)
#ifndef FS_LA
#define FS_LA 2
#endif
#ifndef FS_DBG
#define FS_DBG 0
#endif
#ifdef FS_STANDALONE
thread_local int g_last_hip_error = 0;
float* stream_workspace(size_t bytes, hipStream_t) {           // the standalone experiment build has no gemm.hip: one growing buffer
    static float* p = nullptr; static size_t n = 0;
    if (bytes > n) { if (p) (void)hipFree(p); p = nullptr; n = 0; if (hipMalloc((void**)&p, bytes) != hipSuccess) return nullptr; n = bytes; }
    return p;
}
#endif
constexpr int FS_CHUNK = 65536;                              // W1_hi | W1_lo | W2_hi | W2_lo of one 32-unit chunk: 4 x 16 fragments of 1 KB
constexpr int FS_IMG = 32768;                                // the W1 (or W2) half of a chunk image: hi fragments, then lo fragments
constexpr int FS_W2_OFF = 2 * FS_IMG;                        // LDS: W1 ring (2 stages), then W2 ring (2 stages)
constexpr int FS_B1_OFF = 4 * FS_IMG;
constexpr int FS_MAX_DFF = 2048;
constexpr int FS_PAD = 2;                                    // zero chunks behind the image (streamed, multiplied into nothing)
constexpr int FS_PRM_OFF = FS_B1_OFF + (FS_MAX_DFF + 32 * (FS_PAD + 1)) * 4;       // b2 | gamma | beta (3 x 256 floats)
constexpr int FS_LDS = FS_PRM_OFF + 3 * 256 * 4;

// LDS-DMA: wave-uniform global base in SGPRs + per-lane byte offset; OFF (< 4096) applies to both the global address and the LDS destination
template <int OFF> __device__ __forceinline__ void fs_glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst), "n"(OFF) : "memory");
}
// (the immediate must be a compile-time constant: in the unrolled group loops `k` folds to one)
__device__ __forceinline__ void fs_piece(int k, const void* sbase, unsigned voff, unsigned lds_dst) {
    if (k == 0) fs_glds16<0>(sbase, voff, lds_dst);
    else if (k == 1) fs_glds16<1024>(sbase, voff, lds_dst);
    else if (k == 2) fs_glds16<2048>(sbase, voff, lds_dst);
    else fs_glds16<3072>(sbase, voff, lds_dst);
}
__device__ __forceinline__ fs_f32x16_t fs_mma(const uint4& a, const uint4& b, fs_f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(fs_f16x8_t, a), __builtin_bit_cast(fs_f16x8_t, b), c, 0, 0, 0);
}
// 2 fp32 -> packed fp16 hi pair and lo pair (lo = fp16(x - hi): the difference is exact, both conversions round to nearest even)
__device__ __forceinline__ void fs_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    const fs_f16x2_t a = __builtin_convertvector(f32x2_hw_t{x0, x1}, fs_f16x2_t);
    const fs_f16x2_t b = __builtin_convertvector(f32x2_hw_t{x0 - (float)a[0], x1 - (float)a[1]}, fs_f16x2_t);
    hi = __builtin_bit_cast(uint32_t, a);
    lo = __builtin_bit_cast(uint32_t, b);
}

// Schedule.  Iteration c (after the barrier that ends iteration c - 1: W1(c + 1) and W2(c) are visible, he_cur = phase A of chunk c):
//     groups  0..15: phase A of chunk c + 1 into he_nxt (W1 ring stage (c + 1) & 1), one DMA piece per group -- W1(c + 2) into the stage
//                    phase A of chunk c left in iteration c - 1, W2(c + 1) into the stage phase B of chunk c - 1 left --, and the H
//                    epilogue of chunk c (+ b1, ReLU, split: he_cur -> hs) as four VALU slices in the shadow of the MFMAs;
//     groups 16..31: phase B of chunk c (W2 ring stage c & 1) with hs;
//     then s_waitcnt vmcnt(0) (the pieces had >= 16 groups to land) and ONE barrier.
// A group = the two fragment reads of group g + 2 (hi, lo), then the three MFMAs of group g: every ds_read_b128 has two groups (192
// matrix cycles) to land; __builtin_amdgcn_sched_barrier(0) between groups keeps hipcc from re-serialising read -> wait -> MFMA (its
// own schedule of the straightforward loop used ONE fragment register pair).  The packed image carries FS_PAD zero chunks, and an odd
// chunk count is padded by one more (b1 = 0 there: relu(0) = 0 adds nothing), so the steady state has no conditionals.
// PART (round 5): the hidden-dimension split of the LAST, partial round.  1360 tiles on 256 CUs are 5.31 rounds of one workgroup per CU:
// the sixth round ran 80 workgroups for a full tile time (31% of the chip, 148 of the call's 889 us).  With PART the tail tiles are
// launched NS-fold: workgroup b = tile (tile0 + b / NS), part b % NS multiplies only the hidden chunks [cb[part], cb[part + 1]) (even
// counts; the image's look-ahead chunks behind a part are the next part's real chunks or the zero padding: streamed, phase-A'd into a
// discarded accumulator) and stores its RAW Y^T accumulators to part `part` of a workspace; ffn_split_finish_kernel adds the parts,
// b2 and the residual and normalises.  Summation order of the tail rows differs from the other rows' by that regrouping only.
struct FsParts { int ns, tile0, cb[5]; };
template <bool PART>
__global__ __launch_bounds__(256, 1) void ffn_split_kernel(
    const float* __restrict__ X, const unsigned char* __restrict__ Wp, const float* __restrict__ b1, const float* __restrict__ b2,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* __restrict__ Y, int M, int d_ff, FsParts fp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fs_smem[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)fs_smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31, hh = lane >> 5;
    const int part = PART ? (int)blockIdx.x % fp.ns : 0;
    const int tile = PART ? fp.tile0 + (int)blockIdx.x / fp.ns : (int)blockIdx.x;
    const int c_lo = PART ? (part == 0 ? fp.cb[0] : part == 1 ? fp.cb[1] : part == 2 ? fp.cb[2] : fp.cb[3]) : 0;
    const int c_hi = PART ? (part == 0 ? fp.cb[1] : part == 1 ? fp.cb[2] : part == 2 ? fp.cb[3] : fp.cb[4]) : 0;
    const int nchunk = d_ff >> 5;
    const int nc2 = PART ? (c_hi - c_lo) : ((nchunk + 1) & ~1);        // chunks multiplied (an odd count runs one zero chunk)
    if (PART) { Wp += (long)c_lo * FS_CHUNK; }
    const int b1_off = c_lo * 32;                                       // first hidden unit of this workgroup's chunk range
    const long tok0 = (long)tile * 128 + wave * 32;
    const long tok = min(tok0 + j, (long)M - 1);               // rows past M are clamped (computed, never stored)

    // ---- weight DMA: wave w moves bytes [8 w KB, 8 (w + 1) KB) of a 32 KB W1 / W2 image, 8 pieces of 1 KB ------------------------------
    const unsigned vlane = (unsigned)lane * 16u;
    const unsigned char* Wb = Wp + wave * 8192;
    const unsigned mine = lds_base + (unsigned)wave * 8192u;
    // piece P (0..7) of image IMG (0 = W1, 1 = W2) of chunk C into ring stage ST
#define FS_PIECE(IMG, C, ST, P)                                                                    \
    fs_piece((P) & 3, Wb + (long)(C) * FS_CHUNK + (IMG) * FS_IMG + ((P) >> 2) * 4096, vlane,       \
             mine + (unsigned)((IMG) * FS_W2_OFF + (ST) * FS_IMG + ((P) >> 2) * 4096));
#define FS_IMAGE(IMG, C, ST) { FS_PIECE(IMG, C, ST, 0) FS_PIECE(IMG, C, ST, 1) FS_PIECE(IMG, C, ST, 2) FS_PIECE(IMG, C, ST, 3) \
                               FS_PIECE(IMG, C, ST, 4) FS_PIECE(IMG, C, ST, 5) FS_PIECE(IMG, C, ST, 6) FS_PIECE(IMG, C, ST, 7) }
    FS_IMAGE(0, 0, 0)
    FS_IMAGE(1, 0, 0)
    FS_IMAGE(0, 1, 1)

    // ---- X^T B-fragments, split once: lane (j, hh) holds X[tok][16 s + 8 hh .. + 7] as xh[s] | xl[s] -----------------------------------
    uint4 xh[16], xl[16];
    {
        const float* xr = X + tok * 256 + hh * 8;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(xr + s * 16), c = *reinterpret_cast<const float4*>(xr + s * 16 + 4);
            fs_split2(a.x, a.y, xh[s].x, xl[s].x); fs_split2(a.z, a.w, xh[s].y, xl[s].y);
            fs_split2(c.x, c.y, xh[s].z, xl[s].z); fs_split2(c.z, c.w, xh[s].w, xl[s].w);
        }
    }
    {   // b1 table (zero behind d_ff: the padding chunks) and the epilogue parameters
        float* b1s = reinterpret_cast<float*>(fs_smem + FS_B1_OFF);
        for (int i = (int)threadIdx.x * 4; i < (nc2 + FS_PAD) * 32; i += 256 * 4)
            *reinterpret_cast<float4*>(b1s + i) = b1_off + i < d_ff ? *reinterpret_cast<const float4*>(b1 + b1_off + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        float* prm = reinterpret_cast<float*>(fs_smem + FS_PRM_OFF);
        prm[threadIdx.x] = b2[threadIdx.x];
        prm[256 + threadIdx.x] = gamma[threadIdx.x];
        prm[512 + threadIdx.x] = beta[threadIdx.x];
    }
    fs_f32x16_t yacc[8], he0, he1, zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) { zero16[r] = 0.f; he0[r] = 0.f; he1[r] = 0.f; }
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) yacc[ct] = zero16;
    uint4 hsh[2], hsl[2];                                      // H^T B-fragments (hi, lo) of the chunk phase B multiplies, per k-step
    const unsigned char* lbase = fs_smem + lane * 16;
#define FS_F1(ST, G, PART) (*reinterpret_cast<const uint4*>(lbase + (ST) * FS_IMG + (PART) * 16384 + (G) * 1024))
#define FS_F2(ST, F, PART) (*reinterpret_cast<const uint4*>(lbase + FS_W2_OFF + (ST) * FS_IMG + (PART) * 16384 + (F) * 1024))

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {   // ---- prologue: phase A of chunk 0 (W1 ring stage 0) ------------------------------------------------------------------------
        constexpr int NB = FS_LA + 1;
        uint4 fh[NB], fl[NB];
#pragma unroll
        for (int g = 0; g < FS_LA; ++g) { fh[g] = FS_F1(0, g, 0); fl[g] = FS_F1(0, g, 1); }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (g + FS_LA < 16) { fh[(g + FS_LA) % NB] = FS_F1(0, g + FS_LA, 0); fl[(g + FS_LA) % NB] = FS_F1(0, g + FS_LA, 1); }
            he0 = fs_mma(fh[g % NB], xl[g], g == 0 ? zero16 : he0);
            he0 = fs_mma(fl[g % NB], xh[g], he0);
            he0 = fs_mma(fh[g % NB], xh[g], he0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_barrier();                              // every wave has left W1 stage 0: iteration 0 may overwrite it

#define FS_ITER(C, HC, HN)                                                                         \
    {                                                                                              \
        const int st1_ = ((C) + 1) & 1, st2_ = (C) & 1;        /* stages of W1(C + 1) and W2(C); the DMA targets are the OTHER stages */ \
        const float* b1c_ = reinterpret_cast<const float*>(fs_smem + FS_B1_OFF) + (C) * 32 + 4 * hh; \
        float4 bq_[4];                                                                             \
        constexpr int NB = FS_LA + 1;                                                              \
        uint4 fh[NB], fl[NB];                                                                      \
        _Pragma("unroll") for (int g = 0; g < FS_LA; ++g) {                                        \
            if (!(FS_DBG & 4)) { fh[g] = FS_F1(st1_, g, 0); fl[g] = FS_F1(st1_, g, 1); }           \
            else { fh[g] = xh[g]; fl[g] = xl[g]; }                                                 \
        }                                                                                          \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) bq_[q] = *reinterpret_cast<const float4*>(b1c_ + 8 * q); \
        _Pragma("unroll") for (int g = 0; g < 32; ++g) {                                           \
            if (FS_DBG & 4) { if (g + FS_LA < 32) { fh[(g + FS_LA) % NB] = xh[g & 15]; fl[(g + FS_LA) % NB] = xl[g & 15]; } } \
            else if (g + FS_LA < 16) { fh[(g + FS_LA) % NB] = FS_F1(st1_, g + FS_LA, 0); fl[(g + FS_LA) % NB] = FS_F1(st1_, g + FS_LA, 1); } \
            else if (g + FS_LA < 32) { fh[(g + FS_LA) % NB] = FS_F2(st2_, g + FS_LA - 16, 0); fl[(g + FS_LA) % NB] = FS_F2(st2_, g + FS_LA - 16, 1); } \
            if (g < 16) {                                                                          \
                HN = fs_mma(fh[g % NB], xl[g], g == 0 ? zero16 : HN);                              \
                HN = fs_mma(fl[g % NB], xh[g], HN);                                                \
                HN = fs_mma(fh[g % NB], xh[g], HN);                                                \
                if (!(FS_DBG & 1)) { if (g < 8) FS_PIECE(0, (C) + 2, st2_, g) else FS_PIECE(1, (C) + 1, st1_, g - 8) } \
                if (g >= 2 && g < 10 && (g & 1) == 0) {        /* H epilogue slice q: registers 4 q .. 4 q + 3 of HC */ \
                    const int q = (g - 2) >> 1;                                                    \
                    const float v0 = fmaxf(HC[4 * q] + bq_[q].x, 0.f), v1 = fmaxf(HC[4 * q + 1] + bq_[q].y, 0.f); \
                    const float v2 = fmaxf(HC[4 * q + 2] + bq_[q].z, 0.f), v3 = fmaxf(HC[4 * q + 3] + bq_[q].w, 0.f); \
                    uint32_t h0, l0, h1, l1;                                                       \
                    fs_split2(v0, v1, h0, l0); fs_split2(v2, v3, h1, l1);                          \
                    if (q == 0) { hsh[0].x = h0; hsh[0].y = h1; hsl[0].x = l0; hsl[0].y = l1; }    \
                    else if (q == 1) { hsh[0].z = h0; hsh[0].w = h1; hsl[0].z = l0; hsl[0].w = l1; } \
                    else if (q == 2) { hsh[1].x = h0; hsh[1].y = h1; hsl[1].x = l0; hsl[1].y = l1; } \
                    else { hsh[1].z = h0; hsh[1].w = h1; hsl[1].z = l0; hsl[1].w = l1; }           \
                }                                                                                  \
            } else {                                                                               \
                const int ct = (g - 16) & 7, s = (g - 16) >> 3;                                    \
                yacc[ct] = fs_mma(fh[g % NB], hsl[s], yacc[ct]);                                   \
                yacc[ct] = fs_mma(fl[g % NB], hsh[s], yacc[ct]);                                   \
                yacc[ct] = fs_mma(fh[g % NB], hsh[s], yacc[ct]);                                   \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
        if (!(FS_DBG & 2)) {                                                                       \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                            \
            __builtin_amdgcn_s_barrier();                                                          \
        }                                                                                          \
    }
    for (int c = 0; c < nc2; c += 2) {
        FS_ITER(c, he0, he1)
        FS_ITER(c + 1, he1, he0)
    }
#undef FS_ITER
#undef FS_F1
#undef FS_F2
#undef FS_IMAGE
#undef FS_PIECE

    if constexpr (PART) {
        // raw partial sums of this chunk range: workspace [ns][tail rows][256], same lane -> channel map as the final store below
        if (tok0 + j < M) {
            const long trow = (tok0 + j) - (long)fp.tile0 * 128;
            const long tail_rows = (long)M - (long)fp.tile0 * 128;
            float* prow = Y + ((long)part * tail_rows + trow) * 256 + 4 * hh;
#pragma unroll
            for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(prow + 32 * ct + 8 * q) = make_float4(yacc[ct][4 * q], yacc[ct][4 * q + 1], yacc[ct][4 * q + 2], yacc[ct][4 * q + 3]);
        }
        return;
    }
    // ---- epilogue: + b2 + X (fp32 residual), LayerNorm, store.  Lane (j, hh), tile ct, register r: channel 32 ct + 8 (r >> 2) + 4 hh + (r & 3) ----
    const float* prm_ = reinterpret_cast<const float*>(fs_smem + FS_PRM_OFF);
    const float* xrow = X + tok * 256 + 4 * hh;
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 xx = *reinterpret_cast<const float4*>(xrow + 32 * ct + 8 * q);
            const float4 bb = *reinterpret_cast<const float4*>(prm_ + 32 * ct + 8 * q + 4 * hh);
            yacc[ct][4 * q] += bb.x + xx.x; yacc[ct][4 * q + 1] += bb.y + xx.y; yacc[ct][4 * q + 2] += bb.z + xx.z; yacc[ct][4 * q + 3] += bb.w + xx.w;
            sum += (yacc[ct][4 * q] + yacc[ct][4 * q + 1]) + (yacc[ct][4 * q + 2] + yacc[ct][4 * q + 3]);
        }
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / 256.0f);
    float sq = 0.f;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = yacc[ct][r] - mean; sq += d * d; }
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = rsqrtf(sq * (1.0f / 256.0f) + eps);
    if (tok0 + j < M) {
        float* yrow = Y + (tok0 + j) * 256 + 4 * hh;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = 32 * ct + 8 * q + 4 * hh;
                const float4 ga = *reinterpret_cast<const float4*>(prm_ + 256 + ch), be = *reinterpret_cast<const float4*>(prm_ + 512 + ch);
                *reinterpret_cast<float4*>(yrow + 32 * ct + 8 * q) =
                    make_float4((yacc[ct][4 * q] - mean) * rstd * ga.x + be.x, (yacc[ct][4 * q + 1] - mean) * rstd * ga.y + be.y,
                                (yacc[ct][4 * q + 2] - mean) * rstd * ga.z + be.z, (yacc[ct][4 * q + 3] - mean) * rstd * ga.w + be.w);
            }
    }
}

// tail rows: Y[row] = LayerNorm(sum of the NS partial rows + b2 + X[row]); one wave per row, lane l owns channels 4 l .. 4 l + 3.
// Two-pass statistics like the main epilogue.
__global__ __launch_bounds__(256) void ffn_split_finish_kernel(const float* __restrict__ P, int ns, long tail_rows, const float* __restrict__ X,
                                                               const float* __restrict__ b2, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, float* __restrict__ Y)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= tail_rows) return;
    float4 v = *reinterpret_cast<const float4*>(P + row * 256 + 4 * lane);
    for (int p = 1; p < ns; ++p) {
        const float4 t = *reinterpret_cast<const float4*>(P + ((long)p * tail_rows + row) * 256 + 4 * lane);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const float4 bb = *reinterpret_cast<const float4*>(b2 + 4 * lane), xx = *reinterpret_cast<const float4*>(X + row * 256 + 4 * lane);
    v.x += bb.x + xx.x; v.y += bb.y + xx.y; v.z += bb.z + xx.z; v.w += bb.w + xx.w;
    const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    const float rstd = rsqrtf(wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / 256.0f) + eps);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * lane), be = *reinterpret_cast<const float4*>(beta + 4 * lane);
    *reinterpret_cast<float4*>(Y + row * 256 + 4 * lane) = make_float4(dx * rstd * ga.x + be.x, dy * rstd * ga.y + be.y, dz * rstd * ga.z + be.z, dw * rstd * ga.w + be.w);
}

}  // namespace dtlr

using namespace dtlr;

extern "C" int dtlr_ffn_split_pad_chunks(void) { return FS_PAD; }

// X, Y [M, 256] fp32; Wp = the chunk-major split image of (W1, W2) written by ops.ffn_split_pack: even(d_ff / 32) + dtlr_ffn_split_pad_chunks()
// blocks of 64 KB (W1_hi | W1_lo | W2_hi | W2_lo, 16 fragments of 1 KB each; fragment layouts as dtlr_ffn32_pack_weights; zero blocks behind
// the last real chunk); b1 [d_ff], b2 / gamma / beta [256] fp32.
extern "C" int dtlr_ffn_split(const void* X, const void* Wp, const float* b1, const float* b2, const float* gamma, const float* beta,
                              float eps, void* Y, long M, int d_ff, void* stream)
{
    clear_stale_error();
    if (!X || !Wp || !b1 || !b2 || !gamma || !beta || !Y) return DTLR_EINVAL;
    if (M <= 0 || M > 0x7fffffffL) return DTLR_EINVAL;
    if (d_ff < 32 || d_ff > FS_MAX_DFF || (d_ff & 31)) return DTLR_ESHAPE;
    static DevOnce once;
    if (once.first()) {
        (void)hipFuncSetAttribute((const void*)ffn_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FS_LDS);
        (void)hipFuncSetAttribute((const void*)ffn_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FS_LDS);
        (void)hipGetLastError();
    }
    hipStream_t st = (hipStream_t)stream;
    const long ntiles = (M + 127) / 128;
    // whole rounds of one workgroup per CU on the plain kernel; a last round that fills at most half of the chip is split over the hidden
    // dimension (see ffn_split_kernel<true>): NS = 2..4 parts of an even number of chunks each, as many as fit one round
    const int NCU = 256;
    const long full = (ntiles / NCU) * NCU, rem = ntiles - full;
    const int nc2 = ((d_ff >> 5) + 1) & ~1;
    int ns = (rem > 0 && full > 0) ? (int)(NCU / rem) : 1;
    if (ns > 4) ns = 4;
    if (ns > nc2 / 8) ns = nc2 / 8;                              // at least 8 chunks per part
    float* ws = nullptr;
    const long tail_rows = M - full * 128;
    if (ns >= 2) ws = stream_workspace((size_t)ns * tail_rows * 256 * sizeof(float), st);
    if (ns < 2 || !ws) {
        hipLaunchKernelGGL(ffn_split_kernel<false>, dim3((unsigned)ntiles), dim3(256), FS_LDS, st,
                           (const float*)X, (const unsigned char*)Wp, b1, b2, gamma, beta, eps, (float*)Y, (int)M, d_ff, FsParts{});
        return check_launch();
    }
    FsParts fp{};
    fp.ns = ns; fp.tile0 = (int)full;
    for (int p = 0; p <= ns; ++p) fp.cb[p] = (int)(((long)nc2 / 2 * p / ns) * 2);      // even boundaries, cb[0] = 0, cb[ns] = nc2
    hipLaunchKernelGGL(ffn_split_kernel<false>, dim3((unsigned)full), dim3(256), FS_LDS, st,
                       (const float*)X, (const unsigned char*)Wp, b1, b2, gamma, beta, eps, (float*)Y, (int)(full * 128), d_ff, FsParts{});
    hipLaunchKernelGGL(ffn_split_kernel<true>, dim3((unsigned)(rem * ns)), dim3(256), FS_LDS, st,
                       (const float*)X, (const unsigned char*)Wp, b1, b2, gamma, beta, eps, ws, (int)M, d_ff, fp);
    hipLaunchKernelGGL(ffn_split_finish_kernel, dim3((unsigned)((tail_rows + 3) / 4)), dim3(256), 0, st,
                       (const float*)ws, ns, tail_rows, (const float*)X + full * 128 * 256, b2, gamma, beta, eps, (float*)Y + full * 128 * 256);
    return check_launch();
}
