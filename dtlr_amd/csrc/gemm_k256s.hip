// Weight-resident streaming projection for the SPLIT-fp32 engine, K = 256, N = 256 (round 4; dtype DTLR_F32S):
//
//     MODE 0:  C[M, 256] = A[M, 256] . W^T + bias,  rows with row_mask != 0 written as zeros      (encoder value_proj: ms_deform_attn.py:94-96)
//     MODE 1:  C[M, 256] = LayerNorm( R + A . W^T + bias )                                         (output_proj + residual + norm1:
//                                                                                                   deformable_transformer.py:810-815)
//     MODE 2:  C[M, 256] = LayerNorm( bias + (row_mask ? 0 : A . W^T) )                            (two-stage front end: enc_output(memory with the
//                                                                                                   invalid rows zeroed) + enc_output_norm,
//                                                                                                   deformable_transformer.py:320-330, utils.py:58-62)
// every product as three fp16 MFMAs on hi + lo halves (gemm.hip, GT<f32s_t>), fp32 accumulation, fp32 residual and statistics.
//
// Why: through the tiled split GEMM these 13 projections per step are HBM streams held at 3.2 TB/s (111 us each at B = 32: eight K slabs, then
// an epilogue during which the loader waves stall), and MODE 1's LayerNorm is a second pass over the same rows (83 us).  This is gemm_k256.hip's
// structure for fp32 rows:
//   * the WEIGHT is resident: wave w of 8 keeps its 32 output channels x 256 k as MFMA A-fragments, hi AND lo (32 fragments = 128 VGPRs), loaded
//     once per workgroup from the packed image (dtlr_k256s_pack_weights: fragment order, so a wave-load is one contiguous KB);
//   * TOKENS stream: tiles of 64 tokens x 1 KB are DMA'd global -> LDS (global_load_lds_dwordx4, one row per instruction) into a raw
//     buffer; the eight waves then split the tile ONCE between them (8 rows each: one ds_read_b128 of 4 fp32, 10 VALU, two ds_write_b64) into
//     a fragment-ready fp16 image -- per token 512 B of hi halves then 512 B of lo halves, the 16-byte chunks of each part XOR-swizzled by
//     (token & 15) so that the 16 tokens of a B-fragment read hit 16 distinct bank groups -- and every wave reads its B-fragments (hi, lo:
//     two ds_read_b128) from that image.  (The first form split in registers from the raw tile: every wave converted ALL 64 tokens, 1024 VALU
//     per wave and tile, and the kernel ran no faster than the tiled GEMM.)  The next tile's DMA is issued as soon as the raw buffer has
//     been converted and flies under this tile's MFMAs and stores;
//   * two barriers per tile; persistent, one workgroup per CU; the accumulators ARE the output slice (C^T layout: a lane holds 4 consecutive
//     channels of one token): 16-byte fp32 stores;
//   * MODE 1: the residual is read in the accumulator layout (16 bytes per lane and tile), the row statistics are exchanged between the eight
//     waves through LDS (two passes: mean, then centred squares -- the fp32 LayerNorm kernels' own arithmetic), one extra barrier pair per tile.
// HBM-bound by construction: 128 KB (MODE 0) / 192 KB (MODE 1) of traffic per 64-token tile against 192 MFMAs and ~700 VALU per wave.
#include "dtlr_common.h"

namespace dtlr {

typedef __attribute__((ext_vector_type(8))) _Float16 ks_f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 ks_f16x2_t;
typedef __attribute__((ext_vector_type(4))) float ks_f32x4_t;

constexpr int KS_TOK = 64;                        // tokens per tile
constexpr int KS_STAGE = KS_TOK * 1024;           // 64 KB: 64 fp32 rows of 256 (raw), and the same size for their hi | lo fp16 image
constexpr int KS_IMG_OFF = KS_STAGE;
constexpr int KS_STAT_OFF = 2 * KS_STAGE;         // statistics exchange: [64 tokens][8 waves] floats
constexpr int KS_PAR_OFF = KS_STAT_OFF + KS_TOK * 8 * 4;   // bias | gamma | beta, [3][256] floats (read in the epilogue: no VGPRs across the MFMAs)
constexpr int KS_LDS = KS_PAR_OFF + 3 * 256 * 4;

__device__ __forceinline__ void ks_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ ks_f32x4_t ks_mma(const uint4& a, const uint4& b, ks_f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ks_f16x8_t, a), __builtin_bit_cast(ks_f16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ void ks_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    const ks_f16x2_t a = __builtin_convertvector(f32x2_hw_t{x0, x1}, ks_f16x2_t);
    const ks_f16x2_t b = __builtin_convertvector(f32x2_hw_t{x0 - (float)a[0], x1 - (float)a[1]}, ks_f16x2_t);
    hi = __builtin_bit_cast(uint32_t, a);
    lo = __builtin_bit_cast(uint32_t, b);
}

// Wp: [2 parts (hi, lo)][8 waves][2 row tiles][8 k-steps][64 lanes][8 halves]: lane (m = l & 15, g = l >> 4) <- W[32 wave + 16 rt + m][32 ks + 8 g + e]
// MASK (compile time): row_mask is read.  (As a run-time `if (row_mask)` around the flag loads and around their pin, hipcc's wait insertion
// assumed a path with the loads issued and the pin skipped, and put vmcnt(3..0) in front of the stores: every tile drained its successor's DMA.)
template <int MODE, bool MASK>
__global__ __launch_bounds__(512, 1) void gemm_k256s_kernel(
    const float* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias, const float* __restrict__ R,
    const uint8_t* __restrict__ row_mask, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    float* __restrict__ C, int M, int tiles_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ks_smem[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ks_smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = lane & 15, g = lane >> 4;
    const int ntiles = (M + KS_TOK - 1) / KS_TOK;
    const int t_begin = (int)blockIdx.x * tiles_per_wg;
    const int t_end = min(t_begin + tiles_per_wg, ntiles);
    if (t_begin >= t_end) return;

    // ---- DMA of token tile t into the raw buffer: 64 rows of 1 KB, this wave issues (and later converts) rows 8 wave .. 8 wave + 7
    auto issue = [&](int t) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = 8 * wave + u;
            const long tok = min((long)t * KS_TOK + row, (long)M - 1);
            ks_glds16(A + tok * 256 + lane * 4, lds_base + (unsigned)(row * 1024));
        }
    };
    issue(t_begin);

    // ---- the resident operand: this wave's 32 channels x 256 k, hi and lo fragments
    uint4 wh[2][8], wl[2][8];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const long f = ((long)(wave * 2 + rt) * 8 + ks) * 64 + lane;
            wh[rt][ks] = *reinterpret_cast<const uint4*>(Wp + f * 8);
            wl[rt][ks] = *reinterpret_cast<const uint4*>(Wp + (8L * 2 * 8 * 64 + f) * 8);
        }
    float* par = reinterpret_cast<float*>(ks_smem + KS_PAR_OFF);
    if (threadIdx.x < 256) {
        par[threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
        if constexpr (MODE != 0) { par[256 + threadIdx.x] = gamma[threadIdx.x]; par[512 + threadIdx.x] = beta[threadIdx.x]; }
    }
    // Pin the resident operands BEFORE the loop: left alone, the compiler waits for these loads lazily at their first use inside the loop
    // body -- a vmcnt(0) executed every iteration right after the next tile's DMA was issued (the first build: no overlap at all).
    // An empty asm that "modifies" each register forces the waits here.
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            asm volatile("" : "+v"(wh[rt][ks].x), "+v"(wh[rt][ks].y), "+v"(wh[rt][ks].z), "+v"(wh[rt][ks].w));
            asm volatile("" : "+v"(wl[rt][ks].x), "+v"(wl[rt][ks].y), "+v"(wl[rt][ks].z), "+v"(wl[rt][ks].w));
        }
    }
    float* stat = reinterpret_cast<float*>(ks_smem + KS_STAT_OFF);

    int mk[4] = {0, 0, 0, 0};
    for (int t = t_begin; t < t_end; ++t) {
        // my rows of tile t have landed.  The counter is in order and the only requests younger than that DMA group are the previous tile's
        // 8 stores per lane, which stay in flight: vmcnt(8), not a write acknowledgement per tile (first tile: nothing younger, vmcnt(0)).
        // (A tail tile issues fewer stores, but it is the last of its workgroup: nothing waits after it.)
        if (t == t_begin) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else              asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        if constexpr (MASK) {                                            // padded batch: this tile's flags, pinned BEFORE the DMA issue (a
#pragma unroll                                                           // wait after it would drain the DMA; this one drains the stores)
            for (int tt = 0; tt < 4; ++tt) mk[tt] = row_mask[min((long)t * KS_TOK + 16 * tt + n, (long)M - 1)];
        }
        // ---- my 8 raw rows into registers: lane L holds k = 4 L .. 4 L + 3 of each; the raw rows are then free, and the NEXT tile's DMA is
        //      issued at once (it flies under this tile's split, MFMAs and stores; the first form issued it after the second barrier)
        float4 raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) raw[u] = *reinterpret_cast<const float4*>(ks_smem + (8 * wave + u) * 1024 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (MASK) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) asm volatile("" : "+v"(mk[tt]));
        }
        if (t + 1 < t_end) issue(t + 1);
        uint2 sh[8], sl[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { ks_split2(raw[u].x, raw[u].y, sh[u].x, sl[u].x); ks_split2(raw[u].z, raw[u].w, sh[u].y, sl[u].y); }
        __builtin_amdgcn_s_barrier();                                    // every wave has finished reading the previous image
        // ---- 8 bytes of hi at chunk (L >> 1) ^ (row & 15), half L & 1; lo 512 B further
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = 8 * wave + u;
            unsigned char* dst = ks_smem + KS_IMG_OFF + row * 1024 + (((lane >> 1) ^ (row & 15)) * 16) + (lane & 1) * 8;
            *reinterpret_cast<uint2*>(dst) = sh[u];
            *reinterpret_cast<uint2*>(dst + 512) = sl[u];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                    // the image is complete
        float4 rr[4][2];
        if constexpr (MODE == 1) {                                       // this tile's residual rows, in the accumulator layout: requested here,
#pragma unroll                                                           // consumed in the epilogue (the MFMA phase hides them)
            for (int tt = 0; tt < 4; ++tt) {
                const long tok = min((long)t * KS_TOK + 16 * tt + n, (long)M - 1);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) rr[tt][rt] = *reinterpret_cast<const float4*>(R + tok * 256 + wave * 32 + rt * 16 + 4 * g);
            }
        }
        const unsigned char* tile = ks_smem + KS_IMG_OFF;
        ks_f32x4_t acc[2][4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[rt][tt] = ks_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int row = 16 * tt + n;                                 // row & 15 == n
            const unsigned char* rp = tile + row * 1024;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const uint4 xh = *reinterpret_cast<const uint4*>(rp + (((4 * ks + g) ^ n) * 16));
                const uint4 xl = *reinterpret_cast<const uint4*>(rp + 512 + (((4 * ks + g) ^ n) * 16));
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    acc[rt][tt] = ks_mma(wh[rt][ks], xl, acc[rt][tt]);
                    acc[rt][tt] = ks_mma(wl[rt][ks], xh, acc[rt][tt]);
                    acc[rt][tt] = ks_mma(wh[rt][ks], xh, acc[rt][tt]);
                }
            }
        }
        // ---- epilogue: lane (g, n), (rt, tt): channels 32 wave + 16 rt + 4 g .. + 3 of token 64 t + 16 tt + n ----
        float4 bv[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) bv[rt] = *reinterpret_cast<const float4*>(par + wave * 32 + rt * 16 + 4 * g);
        if constexpr (MODE == 0) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const long tok = (long)t * KS_TOK + 16 * tt + n;
                if (tok < M) {
                    const bool masked = MASK && mk[tt] != 0;
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        float4 v = make_float4(acc[rt][tt][0] + bv[rt].x, acc[rt][tt][1] + bv[rt].y, acc[rt][tt][2] + bv[rt].z, acc[rt][tt][3] + bv[rt].w);
                        if (masked) v = make_float4(0.f, 0.f, 0.f, 0.f);
                        *reinterpret_cast<float4*>(C + tok * 256 + wave * 32 + rt * 16 + 4 * g) = v;
                    }
                }
            }
        } else {
            float v[4][2][4];
            float ps[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                ps[tt] = 0.f;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (MODE == 1) r = rr[tt][rt];
                    if constexpr (MODE == 2 && MASK) { if (mk[tt] != 0) acc[rt][tt] = ks_f32x4_t{0.f, 0.f, 0.f, 0.f}; }
                    v[tt][rt][0] = acc[rt][tt][0] + bv[rt].x + r.x; v[tt][rt][1] = acc[rt][tt][1] + bv[rt].y + r.y;
                    v[tt][rt][2] = acc[rt][tt][2] + bv[rt].z + r.z; v[tt][rt][3] = acc[rt][tt][3] + bv[rt].w + r.w;
                    ps[tt] += (v[tt][rt][0] + v[tt][rt][1]) + (v[tt][rt][2] + v[tt][rt][3]);
                }
                ps[tt] += __shfl_xor(ps[tt], 16, 64);
                ps[tt] += __shfl_xor(ps[tt], 32, 64);                    // this wave's 32 channels of token (tt, n)
                if (g == 0) stat[(16 * tt + n) * 8 + wave] = ps[tt];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            float mean[4], pq[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const float4 s0 = *reinterpret_cast<const float4*>(stat + (16 * tt + n) * 8), s1 = *reinterpret_cast<const float4*>(stat + (16 * tt + n) * 8 + 4);
                mean[tt] = (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) * (1.0f / 256.0f);
                pq[tt] = 0.f;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[tt][rt][e] - mean[tt]; pq[tt] += d * d; }
                pq[tt] += __shfl_xor(pq[tt], 16, 64);
                pq[tt] += __shfl_xor(pq[tt], 32, 64);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();                                // every wave has read the sums: the table may be overwritten
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                if (g == 0) stat[(16 * tt + n) * 8 + wave] = pq[tt];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const float4 s0 = *reinterpret_cast<const float4*>(stat + (16 * tt + n) * 8), s1 = *reinterpret_cast<const float4*>(stat + (16 * tt + n) * 8 + 4);
                const float var = (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) * (1.0f / 256.0f);
                const float rstd = rsqrtf(var + eps);
                const long tok = (long)t * KS_TOK + 16 * tt + n;
                if (tok < M) {
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        const float4 gvr = *reinterpret_cast<const float4*>(par + 256 + wave * 32 + rt * 16 + 4 * g);
                        const float4 evr = *reinterpret_cast<const float4*>(par + 512 + wave * 32 + rt * 16 + 4 * g);
                        *reinterpret_cast<float4*>(C + tok * 256 + wave * 32 + rt * 16 + 4 * g) =
                            make_float4((v[tt][rt][0] - mean[tt]) * rstd * gvr.x + evr.x, (v[tt][rt][1] - mean[tt]) * rstd * gvr.y + evr.y,
                                        (v[tt][rt][2] - mean[tt]) * rstd * gvr.z + evr.z, (v[tt][rt][3] - mean[tt]) * rstd * gvr.w + evr.w);
                    }
                }
            }
        }
    }
}

// ---- the same projection with several OUTPUT SLICES per launch (round 6) -------------------------------------------------------------------
//     slice y (blockIdx.y):  C_y[M, n_valid_y] = act( A[M, 256] . W_y^T + bias_y + R_y ),   rows with row_mask != 0 written as zeros
// Every slice multiplies the SAME token tiles by its own resident [256, 256] weight image (rows >= n_valid zero: the waves that own them
// only help with the DMA and the split) and writes its own column range / buffer (row stride ldc).  Use: the split-fp32 engine's decoder,
// value_proj(memory) of all six cross-attention layers (ms_deform_attn.py:94-96) as six slices of one [M, 1536] buffer (474-534 us against
// 565 through the tiled GEMM).  res_rows > 0: R_y is ONE [res_rows, ldr] matrix shared by the M / res_rows images (row m pairs with row
// m % res_rows) and tiles are walked position-major; res_rows == 0: R_y (if any) is [M, ldr].
// What was measured while building it (tools/experiments/k256s_multi_bench.py, B = 32, M = 174080; boxes differ by +-10 %):
//   * the slices of one launch do NOT share their token tiles through the XCD's L2 (n plain slices cost n x one slice: 112 / 196 / 266 / 484 us
//     for 1 / 2 / 3 / 6): with 128 KB of stores per tile step and CU a 64 KB tile is evicted before its peer slices read it.  The encoder form
//     (value_proj | offsets | logits of `src` as three slices with the position term as a row-broadcast residual, ms_deform_attn.py:94-98)
//     therefore takes 302-314 us against 89 + 192 for the two launches it would replace: built, tested, OFF in the engine;
//   * the launch time follows 0.10 us per MB read + 0.53 us per MB written (a second 178 MB stream to read costs +18 us, halving the
//     written bytes -46 us) although plain elementwise kernels on the same box read AND write at 6 TB/s: the kernel is bound by its own
//     tile loop, not by HBM.  Ablation of this kernel (run-time switches, same call): 94.6 us complete; without the MFMAs 83.0; without
//     the global stores 77.7; without either 36.4; additionally without the in-loop DMA 29.9; the bare loop (barriers, raw reads) 24.0 --
//     i.e. DMA + split ~12 us, MFMAs ~42 us (192 MFMAs of 17 cycles per SIMD and 32-token tile), stores ~46 us, skeleton ~24 us, of
//     which only the MFMA and store phases overlap.  Neither the tile -> workgroup order (XCD bands / plain bands / strided: 111.7 / 114.3 /
//     112.9 us), nor this software-pipelined loop (two DMA tiles in flight, split of tile t + 1 under the MFMAs of tile t: 111.9 us), nor
//     whole-row 1 KB stores through an LDS transpose (119.5 us) moved it beyond box noise.
struct KsSlice { const uint16_t* Wp; const float* bias; const float* R; float* C; const uint8_t* row_mask; int ldc, ldr, n_valid, relu; };
constexpr int KS_MAX_SLICES = 8;
struct KsMulti { KsSlice s[KS_MAX_SLICES]; };

// Software-pipelined tile loop (round 6).  gemm_k256s_kernel above walks 64-token tiles through raw buffer -> split -> barrier -> image ->
// barrier -> MFMAs -> stores with every wave in the same phase at the same time: 10.5 us per tile and CU whatever the traffic (a launch
// with a second [M, 256] stream to read took 130 us against 112: the kernel was never bandwidth-bound, its phases just do not overlap).
// Here a tile is 32 tokens, the raw buffer AND the image are double-buffered (4 x 32 KB), and an iteration is
//     read my 4 raw rows of tile t + 1 -> registers | request tile t's residual rows | DMA tile t + 3 into the raw slot just read |
//     split tile t + 1 into image slot (t + 1) & 1 | MFMAs on image slot t & 1 | epilogue + stores of tile t | ONE barrier
// so a wave's split of the next tile overlaps the other waves' MFMAs of this one, two tiles of DMA are always in flight, and the three
// MFMAs of a product go to the four accumulators in turn (a dependent MFMA is three others away).
constexpr int KP_TOK = 32;
constexpr int KP_STAGE = KP_TOK * 1024;            // 32 KB: 32 fp32 rows of 256 (raw) / their hi | lo fp16 image
constexpr int KP_IMG_OFF = 2 * KP_STAGE;
constexpr int KP_PAR_OFF = 4 * KP_STAGE;
constexpr int KP_LDS = KP_PAR_OFF + 256 * 4;

template <int N> __device__ __forceinline__ void kp_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

template <bool MASK>
__global__ __launch_bounds__(512, 1) void gemm_k256s_multi_kernel(const float* __restrict__ A, KsMulti P, int M, int tiles_per_wg, int res_rows, int n_img)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ks_smem[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ks_smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = lane & 15, g = lane >> 4;
    const KsSlice& S = P.s[blockIdx.y];
    const uint16_t* __restrict__ Wp = S.Wp;
    const float* __restrict__ R = S.R;
    float* __restrict__ C = S.C;
    const uint8_t* __restrict__ row_mask = S.row_mask;
    const int ldc = S.ldc, ldr = S.ldr, relu = S.relu;
    const bool active = 32 * wave < S.n_valid;                       // wave-uniform: this wave's 32 channels exist
    const bool has_r = active && R != nullptr;
    const int ntiles = (M + KP_TOK - 1) / KP_TOK;
    const int t_begin = (int)blockIdx.x * tiles_per_wg;
    const int t_end = min(t_begin + tiles_per_wg, ntiles);
    if (t_begin >= t_end) return;
    // res_rows > 0: position-major walk -- tile t = position tile t / n_img of image t % n_img (res_rows % 32 == 0)
    auto row0 = [&](int t) -> long {
        if (n_img > 0) { const int pt = t / n_img; return (long)(t - pt * n_img) * res_rows + (long)pt * KP_TOK; }
        return (long)t * KP_TOK;
    };
    // DMA of tile t into raw slot t & 1: 32 rows of 1 KB, this wave issues (and later splits) rows 4 wave .. 4 wave + 3
    auto issue = [&](int t) {
        const long r0 = row0(t);
        const unsigned dst = lds_base + (unsigned)((t & 1) * KP_STAGE);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 4 * wave + u;
            const long tok = min(r0 + row, (long)M - 1);
            ks_glds16(A + tok * 256 + lane * 4, dst + (unsigned)(row * 1024));
        }
    };
    // split my 4 raw rows (registers) into image slot `slot`: 8 bytes of hi at chunk (L >> 1) ^ (row & 15), half L & 1; lo 512 B further
    auto split_store = [&](const float4 (&raw)[4], int slot) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint2 sh, sl;
            ks_split2(raw[u].x, raw[u].y, sh.x, sl.x);
            ks_split2(raw[u].z, raw[u].w, sh.y, sl.y);
            const int row = 4 * wave + u;
            unsigned char* dst = ks_smem + KP_IMG_OFF + slot * KP_STAGE + row * 1024 + (((lane >> 1) ^ (row & 15)) * 16) + (lane & 1) * 8;
            *reinterpret_cast<uint2*>(dst) = sh;
            *reinterpret_cast<uint2*>(dst + 512) = sl;
        }
    };
    auto read_raw = [&](float4 (&raw)[4], int slot) {
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const float4*>(ks_smem + slot * KP_STAGE + (4 * wave + u) * 1024 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    issue(t_begin);
    if (t_begin + 1 < t_end) issue(t_begin + 1);

    uint4 wh[2][8], wl[2][8];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const long f = ((long)(wave * 2 + rt) * 8 + ks) * 64 + lane;
            wh[rt][ks] = active ? *reinterpret_cast<const uint4*>(Wp + f * 8) : make_uint4(0u, 0u, 0u, 0u);
            wl[rt][ks] = active ? *reinterpret_cast<const uint4*>(Wp + (8L * 2 * 8 * 64 + f) * 8) : make_uint4(0u, 0u, 0u, 0u);
        }
    float* par = reinterpret_cast<float*>(ks_smem + KP_PAR_OFF);
    if (threadIdx.x < 256) par[threadIdx.x] = (S.bias && (int)threadIdx.x < S.n_valid) ? S.bias[threadIdx.x] : 0.f;
    kp_wait<0>();                                                    // both prologue tiles and the resident operand have landed
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            asm volatile("" : "+v"(wh[rt][ks].x), "+v"(wh[rt][ks].y), "+v"(wh[rt][ks].z), "+v"(wh[rt][ks].w));
            asm volatile("" : "+v"(wl[rt][ks].x), "+v"(wl[rt][ks].y), "+v"(wl[rt][ks].z), "+v"(wl[rt][ks].w));
        }
    }
    {   // tile t_begin: raw slot 0 -> image slot of its parity; its raw slot then takes tile t_begin + 2
        float4 raw[4];
        read_raw(raw, t_begin & 1);
        if (t_begin + 2 < t_end) issue(t_begin + 2);
        split_store(raw, t_begin & 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    int mk[2] = {0, 0};
    for (int t = t_begin; t < t_end; ++t) {
        const long r0 = row0(t);
        const bool nxt = t + 1 < t_end;                              // a next tile exists: split it in this iteration
        const bool dma3 = t + 3 < t_end;                             // ... and tile t + 3 is requested into the raw slot it frees
        if constexpr (MASK) {                                        // this tile's flags (compiler-counted loads: pinned before anything I count)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) mk[tt] = row_mask[min(r0 + 16 * tt + n, (long)M - 1)];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) asm volatile("" : "+v"(mk[tt]));
        }
        float4 raw[4];
        if (nxt) {
            // my rows of tile t + 1 have landed.  In-order counter; requests younger than that DMA group (issued in iteration t - 2, or in the
            // prologue): stores(t - 2) [4], residual(t - 1) [4], DMA(t + 2) [4], stores(t - 1) [4] -- those that exist.  The first two
            // iterations follow the prologue's vmcnt(0) (the DMA they need was covered by it or by iteration 0's own wait below).
            if constexpr (!MASK) {
                const bool d2 = t + 2 < t_end;
                if (t - t_begin < 2) { if (t == t_begin) {} else kp_wait<0>(); }
                else if (has_r) { if (d2) kp_wait<16>(); else kp_wait<12>(); }
                else            { if (d2) kp_wait<12>(); else kp_wait<8>(); }
            }
            read_raw(raw, (t + 1) & 1);
        }
        uint4 rr[2][2];
        if (has_r) {
            const long rbase = n_img > 0 ? (long)(t / n_img) * KP_TOK : r0;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const long rrow = n_img > 0 ? rbase + 16 * tt + n : min(rbase + 16 * tt + n, (long)M - 1);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rr[tt][rt]) : "v"(R + rrow * ldr + wave * 32 + rt * 16 + 4 * g) : "memory");
            }
        }
        if (dma3) issue(t + 3);
        if (nxt) split_store(raw, (t + 1) & 1);

        ks_f32x4_t acc[2][2];                                        // the accumulators, then (bias / residual / ReLU / mask applied in place) the output values
        if (active) {
            const unsigned char* tile = ks_smem + KP_IMG_OFF + (t & 1) * KP_STAGE;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) acc[rt][tt] = ks_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                uint4 xh[2], xl[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const unsigned char* rp = tile + (16 * tt + n) * 1024 + (((4 * ks + g) ^ n) * 16);
                    xh[tt] = *reinterpret_cast<const uint4*>(rp);
                    xl[tt] = *reinterpret_cast<const uint4*>(rp + 512);
                }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) acc[rt][tt] = ks_mma(wh[rt][ks], xl[tt], acc[rt][tt]);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) acc[rt][tt] = ks_mma(wl[rt][ks], xh[tt], acc[rt][tt]);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) acc[rt][tt] = ks_mma(wh[rt][ks], xh[tt], acc[rt][tt]);
            }
            float4 bv[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) bv[rt] = *reinterpret_cast<const float4*>(par + wave * 32 + rt * 16 + 4 * g);
            if (has_r) {
                // the residual rows have landed: the only requests younger than them are tile t + 3's 4 DMA rows (if issued)
                if (dma3) kp_wait<4>(); else kp_wait<0>();
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) asm volatile("" : "+v"(rr[tt][rt].x), "+v"(rr[tt][rt].y), "+v"(rr[tt][rt].z), "+v"(rr[tt][rt].w));
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const bool masked = MASK && mk[tt] != 0;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    ks_f32x4_t v = acc[rt][tt] + ks_f32x4_t{bv[rt].x, bv[rt].y, bv[rt].z, bv[rt].w};
                    if (has_r) v += ks_f32x4_t{__uint_as_float(rr[tt][rt].x), __uint_as_float(rr[tt][rt].y), __uint_as_float(rr[tt][rt].z), __uint_as_float(rr[tt][rt].w)};
                    if (relu) v = ks_f32x4_t{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                    if (masked) v = ks_f32x4_t{0.f, 0.f, 0.f, 0.f};
                    acc[rt][tt] = v;
                }
            }
        }
        // ---- the output tile leaves through LDS so that every store instruction writes ONE whole row (1 KB contiguous) ------------------
        // As 16-byte stores straight from the accumulator layout an instruction wrote 64 bytes of each of 16 rows, and the launch time
        // followed 0.10 us per MB read + 0.53 us per MB WRITTEN (tools/experiments/k256s_multi_bench.py: 112 us plain, 130 us with a second
        // 178 MB stream to read, 66 us for a 128-channel slice): the scattered 64-byte writes, not the reads, set the time.
        __builtin_amdgcn_s_barrier();                                // every wave has finished its MFMA reads of image(t): the slot is the staging tile now
        unsigned char* stg = ks_smem + KP_IMG_OFF + (t & 1) * KP_STAGE;
        if (active) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)             // row 16 tt + n, 16-byte chunk (8 wave + 4 rt + g) ^ n: conflict-free for the 16 rows of a group
                    *reinterpret_cast<ks_f32x4_t*>(stg + (16 * tt + n) * 1024 + (((8 * wave + 4 * rt + g) ^ n) * 16)) = acc[rt][tt];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            // 4 store instructions per wave and iteration, every wave (the hand-counted waits rely on it; only the ragged last tile of the whole
            // problem issues fewer, in its workgroup's last iteration)
            const int rw = 4 * wave;
            const ks_f32x4_t o0 = *reinterpret_cast<const ks_f32x4_t*>(stg + (rw + 0) * 1024 + ((lane ^ ((rw + 0) & 15)) * 16));
            const ks_f32x4_t o1 = *reinterpret_cast<const ks_f32x4_t*>(stg + (rw + 1) * 1024 + ((lane ^ ((rw + 1) & 15)) * 16));
            const ks_f32x4_t o2 = *reinterpret_cast<const ks_f32x4_t*>(stg + (rw + 2) * 1024 + ((lane ^ ((rw + 2) & 15)) * 16));
            const ks_f32x4_t o3 = *reinterpret_cast<const ks_f32x4_t*>(stg + (rw + 3) * 1024 + ((lane ^ ((rw + 3) & 15)) * 16));
            const bool colok = 4 * lane < S.n_valid;
            float* crow = C + (r0 + rw) * ldc + 4 * lane;
            if (colok && r0 + rw + 0 < M) *reinterpret_cast<ks_f32x4_t*>(crow) = o0;
            if (colok && r0 + rw + 1 < M) *reinterpret_cast<ks_f32x4_t*>(crow + ldc) = o1;
            if (colok && r0 + rw + 2 < M) *reinterpret_cast<ks_f32x4_t*>(crow + 2L * ldc) = o2;
            if (colok && r0 + rw + 3 < M) *reinterpret_cast<ks_f32x4_t*>(crow + 3L * ldc) = o3;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // image(t + 1) complete; image(t) and raw(t + 1) no longer read
    }
}

// fp32 weight [256, 256] -> the resident-operand image: [hi | lo][wave 8][rt 2][ks 8][lane 64][8 halves]
__global__ __launch_bounds__(256) void k256s_pack_kernel(const float* __restrict__ w, uint16_t* __restrict__ out)
{
    const int i = (int)blockIdx.x * 256 + threadIdx.x;           // one thread per (fragment, lane): 8 x 2 x 8 x 64 = 8192
    if (i >= 8192) return;
    const int lane = i & 63, ks = (i >> 6) & 7, rt = (i >> 9) & 1, wave = i >> 10;
    const int m = lane & 15, g = lane >> 4;
    const float* src = w + (long)(32 * wave + 16 * rt + m) * 256 + 32 * ks + 8 * g;
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ks_split2(src[2 * e], src[2 * e + 1], h[e], l[e]);
    *reinterpret_cast<uint4*>(out + (long)i * 8) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(out + (8192L + i) * 8) = make_uint4(l[0], l[1], l[2], l[3]);
}

}  // namespace dtlr

using namespace dtlr;

extern "C" int dtlr_k256s_pack_weights(const float* w, void* out, void* stream)
{
    clear_stale_error();
    if (!w || !out) return DTLR_EINVAL;
    hipLaunchKernelGGL(k256s_pack_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)out);
    return check_launch();
}

// One pass over A [M, 256] fp32 for `nslices` (1..8) output slices (the kernel's notes; dtlr_hip.h documents the structure).
extern "C" int dtlr_gemm_k256s_multi(const float* A, long M, const dtlr_k256s_slice* slices, int nslices, const unsigned char* row_mask, int res_rows, void* stream)
{
    clear_stale_error();
    if (!A || !slices) return DTLR_EINVAL;
    if (M <= 0 || M > 0x7fffffffL || nslices < 1 || nslices > KS_MAX_SLICES || res_rows < 0) return DTLR_EINVAL;
    if (res_rows > 0 && ((res_rows % KP_TOK) || (M % res_rows))) return DTLR_ESHAPE;
    const dtlr_k256s_slice* sl = slices;
    KsMulti P{};
    for (int i = 0; i < nslices; ++i) {
        if (!sl[i].Wp || !sl[i].C) return DTLR_EINVAL;
        if (sl[i].n_valid < 32 || sl[i].n_valid > 256 || (sl[i].n_valid & 31) || sl[i].ldc < sl[i].n_valid || (sl[i].ldc & 3)) return DTLR_ESHAPE;
        if (sl[i].R && (sl[i].ldr < sl[i].n_valid || (sl[i].ldr & 3))) return DTLR_ESHAPE;
        if (((size_t)sl[i].C & 15) || ((size_t)sl[i].R & 15)) return DTLR_ESHAPE;
        P.s[i] = KsSlice{(const uint16_t*)sl[i].Wp, sl[i].bias, sl[i].R, sl[i].C, row_mask, sl[i].ldc, sl[i].ldr, sl[i].n_valid, sl[i].relu};
    }
    for (int i = nslices; i < KS_MAX_SLICES; ++i) P.s[i] = P.s[0];
    const int ntiles = (int)((M + KP_TOK - 1) / KP_TOK);
    // one workgroup per CU (129 KB of LDS): gridDim.x * nslices <= 256; workgroup b of a slice owns tiles [b per, (b + 1) per)
    int gx = 256 / nslices;
    if (gx > ntiles) gx = ntiles;
    const int per = (ntiles + gx - 1) / gx;
    gx = (ntiles + per - 1) / per;
    const int n_img = res_rows > 0 ? (int)(M / res_rows) : 0;
    hipStream_t st = (hipStream_t)stream;
    if (row_mask) {
        static DevOnce once;
        if (once.first()) { (void)hipFuncSetAttribute((const void*)gemm_k256s_multi_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, KP_LDS); (void)hipGetLastError(); }
        hipLaunchKernelGGL((gemm_k256s_multi_kernel<true>), dim3(gx, nslices), dim3(512), KP_LDS, st, A, P, (int)M, per, res_rows, n_img);
    } else {
        static DevOnce once;
        if (once.first()) { (void)hipFuncSetAttribute((const void*)gemm_k256s_multi_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, KP_LDS); (void)hipGetLastError(); }
        hipLaunchKernelGGL((gemm_k256s_multi_kernel<false>), dim3(gx, nslices), dim3(512), KP_LDS, st, A, P, (int)M, per, res_rows, n_img);
    }
    return check_launch();
}

// A, C (and R) [M, 256] fp32; Wp = dtlr_k256s_pack_weights(W [256, 256]) (256 KB); bias [256] fp32 or NULL.
//   R == NULL, gamma == NULL: C = A W^T + bias, rows with row_mask[m] != 0 (may be NULL) written as zeros.
//   R != NULL:                C = LayerNorm(R + A W^T + bias) with gamma / beta [256] fp32 (row_mask ignored).
//   R == NULL, gamma != NULL: C = LayerNorm(bias + A W^T), the product of rows with row_mask[m] != 0 (may be NULL) taken as zero.
extern "C" int dtlr_gemm_k256s(const float* A, const void* Wp, const float* bias, const float* R, const unsigned char* row_mask,
                               const float* gamma, const float* beta, float eps, float* C, long M, void* stream)
{
    clear_stale_error();
    if (!A || !Wp || !C) return DTLR_EINVAL;
    if (M <= 0 || M > 0x7fffffffL) return DTLR_EINVAL;
    if ((R || gamma || beta) && (!gamma || !beta)) return DTLR_EINVAL;
    const int ntiles = (int)((M + KS_TOK - 1) / KS_TOK);
    int nwg = 256;                                               // one workgroup per CU (133 KB of LDS)
    if (nwg > ntiles) nwg = ntiles;
    const int per = (ntiles + nwg - 1) / nwg;
    nwg = (ntiles + per - 1) / per;
    hipStream_t st = (hipStream_t)stream;
#define KS_LAUNCH(MODE, HASM, RES, MASK)                                                                                                       \
    {                                                                                                                                       \
        static DevOnce once;                                                                                                                \
        if (once.first()) { (void)hipFuncSetAttribute((const void*)gemm_k256s_kernel<MODE, HASM>, hipFuncAttributeMaxDynamicSharedMemorySize, KS_LDS); (void)hipGetLastError(); } \
        hipLaunchKernelGGL((gemm_k256s_kernel<MODE, HASM>), dim3(nwg), dim3(512), KS_LDS, st, A, (const uint16_t*)Wp, bias, RES, MASK, gamma, beta, eps, C, (int)M, per); \
    }
    if (R) KS_LAUNCH(1, false, R, (const uint8_t*)nullptr)
    else if (gamma && row_mask) KS_LAUNCH(2, true, (const float*)nullptr, row_mask)
    else if (gamma) KS_LAUNCH(2, false, (const float*)nullptr, row_mask)
    else if (row_mask) KS_LAUNCH(0, true, (const float*)nullptr, row_mask)
    else KS_LAUNCH(0, false, (const float*)nullptr, row_mask)
#undef KS_LAUNCH
    return check_launch();
}
