// Weight-resident projection GEMM for K = 256 (bf16):   C[M, N] = epi( A[M, 256] . W[N, 256]^T ),  N = 256 or 384.
//
// The encoder's value / [offsets|weights] projections and the decoder's batched value projection (SURVEY.md appendix C:
// T x 256 x 256, T x 256 x 384, T x 256 x 1536 with T = 174,080 tokens) move 512 bytes in and 512..768 bytes out per token
// for 131..197 kflop: they are HBM streams, and the tiled kernel (gemm.hip: 128x128 tiles, 4 K-slabs per tile, an epilogue
// during which its loader waves stall) holds them at 3.3-3.8 TB/s of compulsory bytes.  Here the roles are turned around:
//   * the WEIGHT is the resident operand: wave w of the 8 keeps its N/8 output channels x 256 k as MFMA A-fragments in
//     registers for the whole kernel (64 VGPRs at N = 256, 96 at N = 384), loaded once per workgroup from L2;
//   * TOKENS stream: tiles of 64 tokens (32 KB) are DMA'd global -> LDS (global_load_lds_dwordx4: no VGPR staging, no
//     ds_write pass) through a 4-stage ring, three tiles in flight per CU; every DMA instruction reads 8 rows x 128 bytes
//     (full lines).  The 16-byte chunks of a row are permuted on the SOURCE side (chunk c of row r lands in slot c ^ r of its
//     128-byte LDS row) so that the B-fragment reads of the 16x16x32 MFMA -- 16 tokens x 4 k-chunks per ds_read_b128 group --
//     hit 16 distinct 16-byte slots (cdna_hip_programming.md rule 21: linear destination, permuted source, same permutation
//     on the read);
//   * one barrier per 64 tokens (publishes the tile, frees the stage read two barriers ago); no K loop state, no tile
//     epilogue stall: a wave's accumulators ARE its output slice of the tile (C^T layout: a lane holds 4 consecutive channels
//     of one token), rounded, paired with v_permlane16_swap into 16-byte stores exactly like gemm.hip's epilogue;
//   * persistent: one workgroup per CU walks a contiguous run of token tiles.
// Epilogue options: + bias[N] (fp32); + R[(m % res_rows), :] (bf16, row-broadcast: the encoder's pos . W^T term of an
// unpadded batch, see engine.py); rows with row_mask[m] != 0 written as zeros (value.masked_fill, ms_deform_attn.py:95-96);
// C row stride ldc >= N (column slices of a wider matrix: the six decoder value projections share one [T, 1536] buffer).
// HBM-bound by construction: per CU and 64-token tile 64..80 KB of traffic against 64..96 MFMAs per wave.
#include "dtlr_common.h"

namespace dtlr {

typedef __attribute__((ext_vector_type(8))) h16_hw_t k2_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float k2_f32x4_t;

constexpr int K2_TOK = 64;                       // tokens per stage
constexpr int K2_STAGE = K2_TOK * 512;           // 32 KB
constexpr int K2_NS = 4;                         // ring stages
constexpr int K2_LDS = K2_NS * K2_STAGE;         // 128 KB: one workgroup per CU

// LDS-DMA, 64 lanes x 16 bytes: destination = wave-uniform LDS byte address + 16 * lane (counted by hand: section 5.7)
__device__ __forceinline__ void k2_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint4 k2_load16(const void* p) {
    uint4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ uint2 k2_load8(const void* p) {
    uint2 r;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
// scalar base + 32-bit lane offset + immediate: one address register per load instead of two, and the per-row-tile column step is free
template <int OFF> __device__ __forceinline__ uint2 k2_load8_so(const void* sbase, unsigned voff) {
    uint2 r;
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(r) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
    return r;
}
__device__ __forceinline__ unsigned k2_load_u8_so(const void* sbase, unsigned voff) {
    unsigned r;
    asm volatile("global_load_ubyte %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
    return r;
}
__device__ __forceinline__ unsigned k2_load_u8(const void* p) {
    unsigned r;
    asm volatile("global_load_ubyte %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ k2_f32x4_t k2_mma(const uint4& a, const uint4& b, k2_f32x4_t c) {
    return DTLR_MFMA_16x16x32_H16(__builtin_bit_cast(k2_bf16x8_t, a), __builtin_bit_cast(k2_bf16x8_t, b), c, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void k2_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// NRT = 16-channel row tiles per wave: N = 128 * NRT (2 -> 256, 3 -> 384).  Wp: the weight in fragment order (dtlr_k256_pack:
// block ((wave * NRT + rt) * 8 + ks) = 64 lanes x 8 elements, lane (m = l & 15, g = l >> 4) <- W[(wave*NRT + rt)*16 + m][32 ks + 8 g ..]).
template <int NRT, bool RES>
__global__ __launch_bounds__(512, 2) void gemm_k256_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias,
    const uint16_t* __restrict__ resid, int res_rows, const uint8_t* __restrict__ row_mask,
    uint16_t* __restrict__ C, int ldc, int M, int tiles_per_wg, int n_img)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char k2_smem[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)k2_smem;
    constexpr int N = 128 * NRT;
    constexpr int E = (NRT / 2) * 4 + (NRT & 1) * 4;          // global stores per tile per wave (16-byte pairs + 8-byte odd tile)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = lane & 15, g = lane >> 4;
    const int ntiles = (M + K2_TOK - 1) / K2_TOK;
    // workgroup b runs on XCD b % 8; logical ids are contiguous inside an XCD
    const int nwg = (int)gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = (int)blockIdx.x & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ((int)blockIdx.x >> 3);
    const int t_begin = logical * tiles_per_wg;
    const int t_end = min(t_begin + tiles_per_wg, ntiles);
    if (t_begin >= t_end) return;
    const int nt = t_end - t_begin;
    // Tile order.  n_img == 0: tile t = rows [64 t, 64 t + 64).  n_img > 0 (broadcast residual over n_img images of res_rows rows,
    // res_rows % 64 == 0): tile t = position tile t / n_img of image t % n_img -- a workgroup's run of tiles, and the whole band of
    // tiles one XCD owns, reuse a few 64-row residual tiles out of L2 instead of sweeping the whole residual once per image
    // (FETCH_SIZE in row order: 351 MB per launch for 223 MB of operands, the 4 MB residual re-fetched for every image).
    auto row0 = [&](int t) -> long {
        if (RES && n_img > 0) { const int pt = t / n_img; return (long)(t - pt * n_img) * res_rows + (long)pt * K2_TOK; }
        return (long)t * K2_TOK;
    };

    // ---- DMA of token tile t into ring slot: 32 blocks of 8 rows x 128 B; this wave issues blocks j = 4 wave + u ------------
    const int dr = lane >> 3, dc = (lane & 7) ^ dr;           // row within the block; SOURCE chunk that lands in slot (lane & 7)
    auto issue = [&](int t, int slot) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = 4 * wave + u, tt8 = j >> 2, kb = j & 3;
            const long tok = min(row0(t) + tt8 * 8 + dr, (long)M - 1);
            k2_glds16(A + tok * 256 + kb * 64 + dc * 8, lds_base + (unsigned)(slot * K2_STAGE + j * 1024));
        }
    };
    // prologue: up to three tiles in flight, then the resident operand
    issue(t_begin, 0);
    if (nt > 1) issue(t_begin + 1, 1);
    if (nt > 2) issue(t_begin + 2, 2);
    uint4 wf[NRT][8];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wf[rt][ks] = k2_load16(Wp + ((long)((wave * NRT + rt) * 8 + ks) * 64 + lane) * 8);
    float4 bv[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
        bv[rt] = bias ? *reinterpret_cast<const float4*>(bias + (wave * NRT + rt) * 16 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);

    // B-fragment read addresses: token tile tt, k-step ks: block (2 tt + (n >> 3), ks >> 1), row n & 7, slot ((ks & 1) * 4 + g) ^ (n & 7)
    const unsigned rd0 = (unsigned)((n >> 3) * 4096 + (n & 7) * 128 + ((g ^ (n & 7)) * 16));
    const unsigned rd1 = (unsigned)((n >> 3) * 4096 + (n & 7) * 128 + (((4 + g) ^ (n & 7)) * 16));

    // Row-wise epilogue operands (RES): tile i + 1's residual rows and padding flags are loaded in the epilogue of tile i BEFORE the
    // next DMA group and tile i's stores, waited for at the end of that epilogue with vmcnt(E + 4) -- everything but those E stores and
    // that DMA group -- and folded into the INITIAL value of tile i + 1's accumulators (bias + residual), so they occupy no register
    // across the MFMA phase and are never live in a register the compiler might copy before the data has landed.
    // (First version: loaded at the top of iteration i and waited for in the same iteration with vmcnt(4); the counter is in order,
    // so that wait also drained the PREVIOUS tile's stores, a write acknowledgement from HBM per tile: 73 us against 50 us for the
    // same launch without the residual.)
    // Row order (n_img == 0): residual row of this lane's tokens advanced by 64 per tile (no 64-bit modulo per tile).
    int rrow[4] = {0, 0, 0, 0};
    if constexpr (RES) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) rrow[tt] = (int)(((long)t_begin * K2_TOK + tt * 16 + n) % res_rows);
    }
    const int radv = (RES && res_rows > 0) ? K2_TOK % res_rows : 0;
    uint2 rs[NRT][4];
    unsigned msk[4] = {0, 0, 0, 0};
    auto load_row_operands = [&](int t, uint2 (&r)[NRT][4], unsigned (&m)[4]) {
        if (n_img > 0) {
            const int pt = t / n_img;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) rrow[tt] = pt * K2_TOK + tt * 16 + n;
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const unsigned voff = (unsigned)rrow[tt] * (unsigned)(N * 2) + (unsigned)(wave * NRT * 32 + 8 * g);
            r[0][tt] = k2_load8_so<0>(resid, voff);
            if constexpr (NRT > 1) r[1][tt] = k2_load8_so<32>(resid, voff);
            if constexpr (NRT > 2) r[2][tt] = k2_load8_so<64>(resid, voff);
            rrow[tt] += radv;
            if (rrow[tt] >= res_rows) rrow[tt] -= res_rows;
        }
        if (row_mask) {
            const int r0 = (int)row0(t);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) m[tt] = k2_load_u8_so(row_mask, (unsigned)min(r0 + tt * 16 + n, M - 1));
        }
    };
    if constexpr (RES) load_row_operands(t_begin, rs, msk);
    k2_wait<0>();

    k2_f32x4_t acc[NRT][4];
    unsigned zero_bits = 0;
    auto seed_accumulators = [&]() {                          // RES: acc <- bias + residual row; padding flags -> bits
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            if (row_mask && msk[tt]) zero_bits |= 1u << tt;
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
                acc[rt][tt] = k2_f32x4_t{bv[rt].x + h16_lo(rs[rt][tt].x), bv[rt].y + h16_hi(rs[rt][tt].x),
                                         bv[rt].z + h16_lo(rs[rt][tt].y), bv[rt].w + h16_hi(rs[rt][tt].y)};
        }
    };
    if constexpr (RES) seed_accumulators();

    for (int i = 0; i < nt; ++i) {
        const int t = t_begin + i;
        const int slot = i & 3;
        __builtin_amdgcn_s_barrier();                         // tile i published by every wave; stage (i + 3) & 3 no longer read
        const long trow = row0(t);
        if constexpr (!RES) {
            if (row_mask) {
                // (they must be OLDER than the next DMA group: waited with vmcnt(4))
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) msk[tt] = k2_load_u8(row_mask + min(trow + tt * 16 + n, (long)M - 1));
            }
            if (i + 3 < nt) issue(t + 3, (i + 3) & 3);
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc[rt][tt] = k2_f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        const unsigned char* sb = k2_smem + slot * K2_STAGE;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            uint4 bf[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                bf[tt] = *reinterpret_cast<const uint4*>(sb + ((ks & 1) ? rd1 : rd0) + tt * 8192 + (ks >> 1) * 1024);
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc[rt][tt] = k2_mma(wf[rt][ks], bf[tt], acc[rt][tt]);
        }
        // ---- epilogue: bias, broadcast residual, padding rows, bf16, paired 16-byte stores ---------------------------------
        if constexpr (RES) {
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < nt) load_row_operands(t + 1, rs, msk);
            if (i + 3 < nt) issue(t + 3, (i + 3) & 3);
        } else {
            zero_bits = 0;
            if (row_mask) {
                if (i + 3 < nt) k2_wait<4>(); else k2_wait<0>();
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) if (msk[tt]) zero_bits |= 1u << tt;
            }
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const long tok = trow + tt * 16 + n;
            const bool live = tok < M;
            const bool zero = (zero_bits >> tt) & 1u;
            uint32_t pk_lo = 0, pk_hi = 0;
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                float v[4] = {acc[rt][tt][0], acc[rt][tt][1], acc[rt][tt][2], acc[rt][tt][3]};
                if constexpr (!RES) { v[0] += bv[rt].x; v[1] += bv[rt].y; v[2] += bv[rt].z; v[3] += bv[rt].w; }
                if (zero) { v[0] = v[1] = v[2] = v[3] = 0.f; }
                const uint32_t lo = pack_bf16x2(v[0], v[1]), hi = pack_bf16x2(v[2], v[3]);
                if ((rt & 1) == 0 && rt + 1 < NRT) { pk_lo = lo; pk_hi = hi; }
                else if (rt & 1) {
                    // pair the row tiles (rt - 1, rt): exchange halves between lane rows g and g ^ 1 -> a lane owns 8 consecutive channels
                    const auto s0 = __builtin_amdgcn_permlane16_swap(pk_lo, lo, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(pk_hi, hi, false, false);
                    uint16_t* dst = C + tok * (long)ldc + (wave * NRT + rt - 1 + (g & 1)) * 16 + 8 * (g >> 1);
                    if (live) *reinterpret_cast<uint4*>(dst) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                } else {                                      // odd NRT: the last row tile alone, 8-byte stores
                    uint16_t* dst = C + tok * (long)ldc + (wave * NRT + rt) * 16 + 4 * g;
                    if (live) *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
                }
            }
        }
        if constexpr (RES) {
            // in flight after this wait: this tile's E stores and the DMA group issued just before them (none in the last three
            // iterations).  Older, hence landed: the next tile's row operands and DMA groups i + 1, i + 2.  A ragged tile (fewer
            // stores than E) is always a workgroup's last, where nothing is waited for.
            zero_bits = 0;
            if (i + 3 < nt) k2_wait<E + 4>();
            else if (i + 1 < nt) k2_wait<E>();
            if (i + 1 < nt) seed_accumulators();
        } else {
            // tile i + 1 must have landed (mine) before the next barrier: everything but the two newest DMA groups and the
            // stores issued around them may stay in flight
            if (i + 3 < nt) k2_wait<2 * E + 8 <= 16 ? 16 : 24>();
            else k2_wait<0>();
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Output projection + residual + LayerNorm in the same weight-resident form:   Y = LayerNorm(R + A W^T + b),  all [M, 256] bf16.
// (`src = norm1(src + self_attn(...))`, the last op of self_attn being output_proj: models/dino/deformable_transformer.py:810-815,
// ops/modules/ms_deform_attn.py:124.)  ffn.hip's proj_ln_bf16_kernel keeps the TOKENS in registers and re-streams the 128 KB weight
// from L2 for every 64 tokens (348 MB of L2 -> LDS traffic per encoder call on top of 267 MB of HBM traffic); here the weight stays in
// registers and the A and R tiles of 64 tokens are DMA'd through a 2-stage LDS ring (64 KB per stage).  W's rows are assigned to MFMA
// rows so that a lane's two accumulator tiles are 8 CONSECUTIVE channels (row tile e of wave w, MFMA row m <-> channel
// 32 w + 8 (m >> 2) + 4 e + (m & 3)): the residual is one 16-byte LDS read and the result one 16-byte store per token, no lane
// exchange.  LayerNorm statistics are two-pass fp32 like ATen's; the 8 waves of a token exchange partial sums through LDS
// (three barriers per 64 tokens including the one that publishes the tile).
constexpr int PK_STAGE = 2 * K2_STAGE;                      // A tile | R tile
constexpr int PK_LN_OFF = 2 * PK_STAGE;                     // two stages, then [2][64 tokens][8 waves] floats
constexpr int PK_LDS = PK_LN_OFF + 2 * 64 * 8 * 4;

__global__ __launch_bounds__(512, 2) void proj_ln_k256_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias, const uint16_t* __restrict__ R,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, uint16_t* __restrict__ Y, int M, int tiles_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char k2_smem[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)k2_smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = lane & 15, g = lane >> 4;
    const int ntiles = (M + K2_TOK - 1) / K2_TOK;
    const int t_begin = (int)blockIdx.x * tiles_per_wg;
    const int t_end = min(t_begin + tiles_per_wg, ntiles);
    if (t_begin >= t_end) return;
    const int nt = t_end - t_begin;
    const int dr = lane >> 3, dc = (lane & 7) ^ dr;
    auto issue = [&](int t, int slot) {                      // 8 DMA instructions per wave: its 4 blocks of the A tile and of the R tile
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = 4 * wave + u, tt8 = j >> 2, kb = j & 3;
            const long tok = min((long)t * K2_TOK + tt8 * 8 + dr, (long)M - 1);
            k2_glds16(A + tok * 256 + kb * 64 + dc * 8, lds_base + (unsigned)(slot * PK_STAGE + j * 1024));
            k2_glds16(R + tok * 256 + kb * 64 + dc * 8, lds_base + (unsigned)(slot * PK_STAGE + K2_STAGE + j * 1024));
        }
    };
    issue(t_begin, 0);
    uint4 wf[2][8];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wf[e][ks] = k2_load16(Wp + ((long)((wave * 2 + e) * 8 + ks) * 64 + lane) * 8);
    const int ch = 32 * wave + 8 * g;                         // this lane's 8 consecutive channels
    float bs[8], gm[8], bt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs[e] = bias[ch + e]; gm[e] = gamma[ch + e]; bt[e] = beta[ch + e]; }
    k2_wait<0>();
    const unsigned rd0 = (unsigned)((n >> 3) * 4096 + (n & 7) * 128 + ((g ^ (n & 7)) * 16));
    const unsigned rd1 = (unsigned)((n >> 3) * 4096 + (n & 7) * 128 + (((4 + g) ^ (n & 7)) * 16));
    // residual chunk of this lane: 16-byte chunk 4 wave + g of the row -> block kb = wave >> 1, chunk c = 4 (wave & 1) + g
    const unsigned rres = (unsigned)(K2_STAGE + (n >> 3) * 4096 + (wave >> 1) * 1024 + (n & 7) * 128 + (((4 * (wave & 1) + g) ^ (n & 7)) * 16));
    float* lnsum = reinterpret_cast<float*>(k2_smem + PK_LN_OFF);
    float* lnsq = lnsum + 64 * 8;

    for (int i = 0; i < nt; ++i) {
        const int t = t_begin + i;
        const int slot = i & 1;
        __builtin_amdgcn_s_barrier();                         // tile i published; stage (i + 1) & 1 and the LN buffers are free
        if (i + 1 < nt) issue(t + 1, slot ^ 1);
        k2_f32x4_t acc[2][4];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[e][tt] = k2_f32x4_t{0.f, 0.f, 0.f, 0.f};
        const unsigned char* sb = k2_smem + slot * PK_STAGE;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            uint4 bf[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                bf[tt] = *reinterpret_cast<const uint4*>(sb + ((ks & 1) ? rd1 : rd0) + tt * 8192 + (ks >> 1) * 1024);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc[e][tt] = k2_mma(wf[e][ks], bf[tt], acc[e][tt]);
        }
        // v = acc + bias + residual ; first pass: row sums
        float v[4][8], sm[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const uint4 rr = *reinterpret_cast<const uint4*>(sb + rres + tt * 8192);
            const uint32_t rw[4] = {rr.x, rr.y, rr.z, rr.w};
            sm[tt] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float r = (e & 1) ? h16_hi(rw[e >> 1]) : h16_lo(rw[e >> 1]);
                v[tt][e] = acc[e >> 2][tt][e & 3] + bs[e] + r;
                sm[tt] += v[tt][e];
            }
            sm[tt] += __shfl_xor(sm[tt], 16, 64);
            sm[tt] += __shfl_xor(sm[tt], 32, 64);
            if (g == 0) lnsum[(tt * 16 + n) * 8 + wave] = sm[tt];
        }
        __syncthreads();
        float mean[4], q[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const float4 a = *reinterpret_cast<const float4*>(lnsum + (tt * 16 + n) * 8), c = *reinterpret_cast<const float4*>(lnsum + (tt * 16 + n) * 8 + 4);
            mean[tt] = ((a.x + a.y) + (a.z + a.w) + (c.x + c.y) + (c.z + c.w)) * (1.0f / 256.0f);
            q[tt] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[tt][e] - mean[tt]; q[tt] += d * d; }
            q[tt] += __shfl_xor(q[tt], 16, 64);
            q[tt] += __shfl_xor(q[tt], 32, 64);
            if (g == 0) lnsq[(tt * 16 + n) * 8 + wave] = q[tt];
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const float4 a = *reinterpret_cast<const float4*>(lnsq + (tt * 16 + n) * 8), c = *reinterpret_cast<const float4*>(lnsq + (tt * 16 + n) * 8 + 4);
            const float rstd = rsqrtf(((a.x + a.y) + (a.z + a.w) + (c.x + c.y) + (c.z + c.w)) * (1.0f / 256.0f) + eps);
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[tt][e] - mean[tt]) * rstd * gm[e] + bt[e];
            const long tok = (long)t * K2_TOK + tt * 16 + n;
            if (tok < M)
                *reinterpret_cast<uint4*>(Y + tok * 256 + ch) =
                    make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
        }
        // my pieces of tile i + 1 must have landed before the next barrier; this tile's 4 stores may stay in flight
        if (i + 1 < nt) k2_wait<4>(); else k2_wait<0>();
    }
}

}  // namespace dtlr

using namespace dtlr;

// W [256, 256] row-major bf16 (host) -> fragment order with the channel permutation of proj_ln_k256_kernel (host, 65536 elements):
// block ((wave * 2 + e) * 8 + ks) lane (m, g) <- W[32 wave + 8 (m >> 2) + 4 e + (m & 3)][32 ks + 8 g ..]
extern "C" int dtlr_proj_ln_k256_pack_weights(const unsigned short* w_host, unsigned short* wp_host)
{
    if (!w_host || !wp_host) return DTLR_EINVAL;
    for (int wave = 0; wave < 8; ++wave)
        for (int e = 0; e < 2; ++e)
            for (int ks = 0; ks < 8; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int m = lane & 15, g = lane >> 4;
                    const int row = 32 * wave + 8 * (m >> 2) + 4 * e + (m & 3);
                    for (int x = 0; x < 8; ++x)
                        wp_host[((long)((wave * 2 + e) * 8 + ks) * 64 + lane) * 8 + x] = w_host[(long)row * 256 + ks * 32 + g * 8 + x];
                }
    return DTLR_OK;
}

extern "C" int dtlr_proj_ln_k256(const void* A, const void* Wp, const float* bias, const void* R, const float* gamma, const float* beta,
                                 float eps, void* Y, int M, void* stream)
{
    clear_stale_error();
    if (!A || !Wp || !bias || !R || !gamma || !beta || !Y) return DTLR_EINVAL;
    if (M <= 0) return DTLR_EINVAL;
    const int ntiles = (M + K2_TOK - 1) / K2_TOK;
    const int ncu = 256;
    const int grid = ntiles < ncu ? ntiles : ncu;
    const int per = (ntiles + grid - 1) / grid;
    const int g2 = (ntiles + per - 1) / per;
    static DevOnce once;
    if (once.first()) { (void)hipFuncSetAttribute((const void*)proj_ln_k256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PK_LDS); (void)hipGetLastError(); }
    hipLaunchKernelGGL(proj_ln_k256_kernel, dim3(g2), dim3(512), PK_LDS, (hipStream_t)stream, (const uint16_t*)A, (const uint16_t*)Wp, bias,
                       (const uint16_t*)R, gamma, beta, eps, (uint16_t*)Y, M, per);
    return check_launch();
}

// W [N, 256] row-major bf16 (host memory) -> fragment order (host memory, N * 256 elements).
extern "C" int dtlr_k256_pack_weights(const unsigned short* w_host, unsigned short* wp_host, int N)
{
    if (!w_host || !wp_host) return DTLR_EINVAL;
    if (N != 256 && N != 384) return DTLR_ESHAPE;
    const int NRT = N / 128;
    for (int wave = 0; wave < 8; ++wave)
        for (int rt = 0; rt < NRT; ++rt)
            for (int ks = 0; ks < 8; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int m = lane & 15, g = lane >> 4;
                    const int row = (wave * NRT + rt) * 16 + m;
                    for (int e = 0; e < 8; ++e)
                        wp_host[((long)((wave * NRT + rt) * 8 + ks) * 64 + lane) * 8 + e] = w_host[(long)row * 256 + ks * 32 + g * 8 + e];
                }
    return DTLR_OK;
}

extern "C" int dtlr_gemm_k256(const void* A, const void* Wp, const float* bias, const void* resid, int res_rows,
                              const unsigned char* row_mask, void* C, int ldc, int M, int N, void* stream)
{
    clear_stale_error();
    if (!A || !Wp || !C) return DTLR_EINVAL;
    if (M <= 0 || ldc < N || (ldc & 7)) return DTLR_EINVAL;
    if (resid && (res_rows <= 0 || (long)res_rows * N * 2 >= (1L << 31))) return DTLR_EINVAL;
    if (N != 256 && N != 384) return DTLR_ESHAPE;
    const int ntiles = (M + K2_TOK - 1) / K2_TOK;
    int ncu = 256;
    {
        static int cached = 0;
        if (!cached) { int d = 0; hipDeviceProp_t p; if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess && p.multiProcessorCount > 0) cached = p.multiProcessorCount; else cached = 256; (void)hipGetLastError(); }
        ncu = cached;
    }
    const int grid = ntiles < ncu ? ntiles : ncu;
    const int per = (ntiles + grid - 1) / grid;
    const int g2 = (ntiles + per - 1) / per;
    hipStream_t st = (hipStream_t)stream;
    static const bool pos_major = exp_env_int("DTLR_K256_POS_MAJOR", 1) != 0;   // experiment builds: A/B timing only
    const int n_img = (pos_major && resid && res_rows % K2_TOK == 0 && M % res_rows == 0 && M / res_rows > 1) ? M / res_rows : 0;
#define K2_LAUNCH(NRT, RES)                                                                        \
    {                                                                                              \
        static DevOnce once;                                                                       \
        if (once.first()) { (void)hipFuncSetAttribute((const void*)gemm_k256_kernel<NRT, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, K2_LDS); (void)hipGetLastError(); } \
        hipLaunchKernelGGL((gemm_k256_kernel<NRT, RES>), dim3(g2), dim3(512), K2_LDS, st, (const uint16_t*)A, (const uint16_t*)Wp, bias, \
                           (const uint16_t*)resid, res_rows, row_mask, (uint16_t*)C, ldc, M, per, n_img);  \
    }
    if (N == 256) { if (resid) K2_LAUNCH(2, true) else K2_LAUNCH(2, false) }
    else { if (resid) K2_LAUNCH(3, true) else K2_LAUNCH(3, false) }
#undef K2_LAUNCH
    return check_launch();
}
