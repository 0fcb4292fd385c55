// Token-stationary class head for LARGE charsets (round 5; 16-bit engines): the three products of a split head
//
//     Y[M, N] = A W_hi^T + B W_hi^T + A W_lo^T + bias          A = X[:, a_off : a_off + 256], B = X[:, b_off : b_off + 256]   (nprod = 3)
//     Y[M, N] = A W_hi^T + A W_lo^T + bias                                                                                      (nprod = 2)
//
// and either the ROW MAXIMUM of Y (mode 0: the two-stage selection score, `topk(max over classes of enc_out_class_embed(output_memory))`,
// models/dino/deformable_transformer.py:341-345) or Y itself in fp32 (mode 1: `class_embed` on the decoder states, models/dino/dino.py:349-352,
// and the two-stage `interm_outputs` logits, dino.py:382-385).  A / B are the hi / lo 16-bit halves of an fp32 row (proj_ln_split's
// [hi | lo | hi] image: nprod 3) or the 16-bit decoder state itself (nprod 2); W_hi = 16-bit(W), W_lo = 16-bit(W - W_hi).
//
// Why: on the Chinese model (N = 7356) the tiled GEMM runs these heads at 0.20 of the MFMA peak -- `rowmax M217600 N7360 K768` alone is
// 4.9 of the 17.0 ms step (profiles/r05_bench_chinese_v1.json).  A 128 x 128 tile re-reads its 128 token rows for every one of the 58
// channel tiles: 98,600 tiles x 393 KB = 38.8 GB of L2 -> LDS traffic per launch, ~4 ms at the ~10 TB/s the L2s deliver; the row-max
// epilogue adds 58 atomics per row.  Here the TOKENS are stationary, as in the fused FFN kernels (ffn32.hip, ffn_split.hip: this is
// ffn_split's phase A on 16-bit operands):
//   * a workgroup = 8 waves (two per SIMD) owns 256 tokens, a wave 32 of them: X^T of its tokens as B-fragments of the 32x32x16 MFMA in
//     registers (xa[s], xb[s]: lane (j = token, hh): k = 16 s + 8 hh .. + 7 -- 128 VGPRs), loaded once;
//   * the classes are walked in chunks of 32: H^T[32 classes, 32 tokens] = Wc X^T as 16 k-steps x nprod MFMAs into ONE accumulator
//     (16 registers); + bias; mode 0: running maximum in a register, mode 1: the tile leaves through a per-wave LDS transpose as full
//     128-byte fp32 rows (four 16-byte stores per lane);
//   * the weight image of a chunk (W_hi fragments | W_lo fragments: 2 x 16 KB, fragment order, packed by ops.head_ts_pack) is DMA'd
//     global -> LDS (global_load_lds_dwordx4, 4 pieces per wave and chunk) into a two-stage ring, chunk c + 1 while chunk c is multiplied;
//     one barrier per chunk.  A weight byte serves 256 tokens: 850 workgroups x 7.5 MB = 6.4 GB of L2 -> LDS traffic per launch.
// 48 MFMAs of 32 cycles per chunk and wave, two waves per SIMD: 3072 matrix cycles per chunk against 64 KB of fragment reads per SIMD pair.
// Arithmetic: fp32 accumulation; per class the k order is (W_hi b, W_lo a, W_hi a) per 16-k step, k ascending -- the same three terms the
// tiled GEMM sums as [hi | lo | hi] . [W_hi | W_hi | W_lo], in another order: results agree to fp32 rounding, not bit for bit.
#include "dtlr_common.h"

namespace dtlr {

typedef __attribute__((ext_vector_type(8))) h16_hw_t ht_h16x8_t;
typedef __attribute__((ext_vector_type(16))) float ht_f32x16_t;

constexpr int HT_CHUNK = 32768;                              // W_hi | W_lo of one 32-class chunk: 2 x 16 fragments of 1 KB
constexpr int HT_PART = 16384;
constexpr int HT_PAD = 2;                                    // zero chunks behind the image (streamed by the look-ahead DMA, never multiplied)
constexpr int HT_LA = 2;                                     // fragment look-ahead in k-steps
constexpr int HT_TOK = 256;                                  // tokens per workgroup of the 8-wave form (the 4-wave form: 128)
constexpr int HT_YP = 36;                                    // row pitch (floats) of the per-wave output scratch of mode 1
constexpr float HT_NEG = -3.0e38f;                           // bias of a padded class: never the maximum, never stored

template <int OFF> __device__ __forceinline__ void ht_glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst), "n"(OFF) : "memory");
}
__device__ __forceinline__ ht_f32x16_t ht_mma(const uint4& a, const uint4& b, ht_f32x16_t c) {
    return DTLR_MFMA_32x32x16_H16(__builtin_bit_cast(ht_h16x8_t, a), __builtin_bit_cast(ht_h16x8_t, b), c, 0, 0, 0);
}

// MODE 0: out = rowmax [M] fp32.  MODE 1: out = Y [M, N] fp32 (N % 4 == 0).  NPROD 2 | 3.  NW = 8 waves (256 tokens per workgroup) or 4
// (128 tokens: for M too small to give every CU an 8-wave workgroup -- the decoder's 28,800 rows are 113 of those on 256 CUs, 225 of these).
template <int MODE, int NPROD, int NW>
__global__ __launch_bounds__(64 * NW, 1) void head_ts_kernel(
    const uint16_t* __restrict__ X, int ldx, int a_off, int b_off, const unsigned char* __restrict__ Wp, const float* __restrict__ bias,
    float* __restrict__ out, int M, int N, int nchunk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ht_smem[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ht_smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31, hh = lane >> 5;
    constexpr int TOKW = 32 * NW;                              // tokens per workgroup
    const long tok0 = (long)blockIdx.x * TOKW + wave * 32;
    const long tok = min(tok0 + j, (long)M - 1);               // rows past M are clamped: computed like row M - 1 (mode 1 stores the same values twice)

    // ---- weight DMA: wave w moves bytes [w, w + 1) x (32 / NW) KB of a 32 KB chunk image, in pieces of 1 KB ------------------------
    const unsigned vlane = (unsigned)lane * 16u;
    constexpr int WSHARE = HT_CHUNK / NW;                      // 4 KB (8 waves) or 8 KB (4 waves)
    const unsigned char* Wb = Wp + wave * WSHARE;
    const unsigned mine = lds_base + (unsigned)wave * (unsigned)WSHARE;
#define HT_IMAGE(C, ST)                                                                            \
    {                                                                                              \
        const unsigned char* s_ = Wb + (long)(C) * HT_CHUNK;                                       \
        const unsigned d_ = mine + (unsigned)(ST) * HT_CHUNK;                                      \
        ht_glds16<0>(s_, vlane, d_); ht_glds16<1024>(s_, vlane, d_); ht_glds16<2048>(s_, vlane, d_); ht_glds16<3072>(s_, vlane, d_); \
        if constexpr (NW == 4) {                                                                   \
            ht_glds16<0>(s_ + 4096, vlane, d_ + 4096u); ht_glds16<1024>(s_ + 4096, vlane, d_ + 4096u);                           \
            ht_glds16<2048>(s_ + 4096, vlane, d_ + 4096u); ht_glds16<3072>(s_ + 4096, vlane, d_ + 4096u);                        \
        }                                                                                          \
    }
    HT_IMAGE(0, 0)

    // ---- X^T B-fragments: lane (j, hh) holds A[tok][16 s + 8 hh .. + 7] as xa[s] (and B[...] as xb[s]) -----------------------------
    uint4 xa[16], xb[16];
    {
        const uint16_t* xr = X + tok * (long)ldx + hh * 8;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            xa[s] = *reinterpret_cast<const uint4*>(xr + a_off + s * 16);
            if constexpr (NPROD == 3) xb[s] = *reinterpret_cast<const uint4*>(xr + b_off + s * 16);
            else xb[s] = make_uint4(0u, 0u, 0u, 0u);               // (never read: the compiler drops it)
        }
    }
    // bias table: 32 (nchunk + HT_PAD) floats behind the ring (padded classes carry HT_NEG from the packer; the look-ahead chunks too)
    float* bs = reinterpret_cast<float*>(ht_smem + 2 * HT_CHUNK);
    for (int i = (int)threadIdx.x * 4; i < nchunk * 32; i += 64 * NW * 4) *reinterpret_cast<float4*>(bs + i) = *reinterpret_cast<const float4*>(bias + i);
    // Pin X before the loop: left alone the compiler waits for these loads at their first use inside the loop, i.e. vmcnt(0) behind a DMA issue.
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        asm volatile("" : "+v"(xa[s].x), "+v"(xa[s].y), "+v"(xa[s].z), "+v"(xa[s].w));
        if constexpr (NPROD == 3) asm volatile("" : "+v"(xb[s].x), "+v"(xb[s].y), "+v"(xb[s].z), "+v"(xb[s].w));
    }
    ht_f32x16_t zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    float rmax = HT_NEG;
    const unsigned char* lbase = ht_smem + lane * 16;
#define HT_F(ST, G, PART) (*reinterpret_cast<const uint4*>(lbase + (ST) * HT_CHUNK + (PART) * HT_PART + (G) * 1024))

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // chunk 0 and the bias table are visible

    for (int c = 0; c < nchunk; ++c) {
        const int st = c & 1;
        // chunk c + 1 -> the other stage (every wave left it at the barrier that ended iteration c - 1; the image has HT_PAD chunks behind nchunk)
        HT_IMAGE(c + 1, st ^ 1)
        constexpr int NB = HT_LA + 1;
        uint4 fh[NB], fl[NB];
#pragma unroll
        for (int g = 0; g < HT_LA; ++g) { fh[g] = HT_F(st, g, 0); fl[g] = HT_F(st, g, 1); }
        const float* bc = bs + c * 32 + 4 * hh;
        float4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(bc + 8 * q);
        ht_f32x16_t acc = zero16;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (g + HT_LA < 16) { fh[(g + HT_LA) % NB] = HT_F(st, g + HT_LA, 0); fl[(g + HT_LA) % NB] = HT_F(st, g + HT_LA, 1); }
            if constexpr (NPROD == 3) acc = ht_mma(fh[g % NB], xb[g], acc);      // W_hi . b
            acc = ht_mma(fl[g % NB], xa[g], acc);                      // W_lo . a
            acc = ht_mma(fh[g % NB], xa[g], acc);                      // W_hi . a
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- chunk epilogue.  Lane (j, hh), register r: class 32 c + 8 (r >> 2) + 4 hh + (r & 3) of token j -------------------------
        if constexpr (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rmax = fmaxf(rmax, fmaxf(fmaxf(acc[4 * q] + bq[q].x, acc[4 * q + 1] + bq[q].y), fmaxf(acc[4 * q + 2] + bq[q].z, acc[4 * q + 3] + bq[q].w)));
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // my pieces of chunk c + 1 have landed
        } else {
            // The accumulator layout gives a lane 4 x 16 bytes of one token's 128-byte class row, interleaved with its partner lane's: stored
            // directly, an instruction wrote 32-byte runs 29 KB apart (1.7 TB/s on the 848 MB logit matrix of the Chinese model).  The
            // 32 x 32 tile goes through a per-wave LDS scratch (row pitch 36 floats: conflict-free 16-byte writes, 2-way reads) and leaves as
            // FULL 128-byte rows: lane l stores piece l & 7 of token 8 it + (l >> 3).  Wave-local: the LDS serves a wave's requests in order.
            float* ys = reinterpret_cast<float*>(ht_smem + 2 * HT_CHUNK + nchunk * 128) + wave * (32 * HT_YP);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(ys + j * HT_YP + 8 * q + 4 * hh) =
                    make_float4(acc[4 * q] + bq[q].x, acc[4 * q + 1] + bq[q].y, acc[4 * q + 2] + bq[q].z, acc[4 * q + 3] + bq[q].w);
            __builtin_amdgcn_wave_barrier();
            const int pc = lane & 7, tr = lane >> 3;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float4 v = *reinterpret_cast<const float4*>(ys + (8 * it + tr) * HT_YP + 4 * pc);
                const long trow = min((long)blockIdx.x * TOKW + wave * 32 + 8 * it + tr, (long)M - 1);      // clamped rows rewrite row M - 1 with its own values
                // the class group 32 c + 4 pc .. + 3 is all real or all padding (N % 4 == 0); piece 0 is always real: every wave issues all four stores
                if (c * 32 + 4 * pc < N) *reinterpret_cast<float4*>(out + trow * (long)N + c * 32 + 4 * pc) = v;
            }
            __builtin_amdgcn_wave_barrier();
            // the DMA pieces of chunk c + 1 are OLDER than this chunk's four stores: the in-order counter at <= 4 means they have landed
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                          // chunk c + 1 visible to everyone; everyone has left stage st
    }
#undef HT_F
#undef HT_IMAGE
    if constexpr (MODE == 0) {
        rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
        if (hh == 0 && tok0 + j < M) out[tok0 + j] = rmax;
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last look-ahead DMA must have landed before the workgroup gives its LDS back
    }
}

}  // namespace dtlr

using namespace dtlr;

extern "C" int dtlr_head_ts_pad_chunks(void) { return HT_PAD; }

// X [M, ldx] 16-bit rows (ldx % 8 == 0; the 256-wide operands start at columns a_off, b_off, both % 8 == 0); Wp = ops.head_ts_pack image:
// nchunk + dtlr_head_ts_pad_chunks() blocks of 32 KB (W_hi fragments | W_lo fragments, 16 of 1 KB each: lane l of k-step s <-
// W[32 c + (l & 31)][16 s + 8 (l >> 5) ..]), nchunk = ceil(N / 32); bias: 32 nchunk floats, classes >= N at -3e38; mode 0: out [M] fp32 row
// maxima, mode 1: out [M, N] fp32 (N % 4 == 0); nprod 3: A Whi + B Whi + A Wlo, nprod 2: A Whi + A Wlo.
extern "C" int dtlr_head_ts(const void* X, int ldx, int a_off, int b_off, const void* Wp, const float* bias, int N, int nprod, int mode,
                            void* out, long M, void* stream)
{
    clear_stale_error();
    if (!X || !Wp || !bias || !out) return DTLR_EINVAL;
    if (M <= 0 || M > 0x7fffffffL || N <= 0) return DTLR_EINVAL;
    if ((ldx & 7) || (a_off & 7) || (b_off & 7) || a_off < 0 || b_off < 0 || a_off + 256 > ldx || (nprod == 3 && b_off + 256 > ldx)) return DTLR_ESHAPE;
    if ((nprod != 2 && nprod != 3) || (mode != 0 && mode != 1) || (mode == 1 && (N & 3))) return DTLR_ESHAPE;
    const int nchunk = (N + 31) / 32;
    const size_t lds = 2 * (size_t)HT_CHUNK + (size_t)nchunk * 32 * sizeof(float) + (mode == 1 ? 8 * 32 * HT_YP * sizeof(float) : 0);
    if (lds > 160 * 1024) return DTLR_ESHAPE;                   // N <= 24576 (mode 0) / 15360 (mode 1)
    hipStream_t st = (hipStream_t)stream;
    // 8-wave workgroups of 256 tokens when they give at least ~3/4 of the 256 CUs one each; 4-wave workgroups of 128 tokens below that
    const bool small = (M + HT_TOK - 1) / HT_TOK < 192;
#define HT_LAUNCH(MODE_, NP_, NW_)                                                                 \
    {                                                                                              \
        static DevOnce once;                                                                       \
        if (once.first()) { (void)hipFuncSetAttribute((const void*)head_ts_kernel<MODE_, NP_, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError(); } \
        hipLaunchKernelGGL((head_ts_kernel<MODE_, NP_, NW_>), dim3((unsigned)((M + 32 * NW_ - 1) / (32 * NW_))), dim3(64 * NW_), lds, st,         \
                           (const uint16_t*)X, ldx, a_off, b_off, (const unsigned char*)Wp, bias, (float*)out, (int)M, N, nchunk); \
    }
#define HT_PICK(MODE_, NP_) { if (small) HT_LAUNCH(MODE_, NP_, 4) else HT_LAUNCH(MODE_, NP_, 8) }
    if (mode == 0) { if (nprod == 3) HT_PICK(0, 3) else HT_PICK(0, 2) }
    else { if (nprod == 3) HT_PICK(1, 3) else HT_PICK(1, 2) }
#undef HT_PICK
#undef HT_LAUNCH
    return check_launch();
}
