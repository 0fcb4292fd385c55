// Weight-resident streaming GEMM with a streamed residual, bf16:   C = relu?( A W^T + b + R ),  A [M, K], W [N, K], R / C [M, ld]
//
// The last 1x1 convolution of a ResNet bottleneck (`out = relu(bn3(conv3(out)) + identity)`, torchvision resnet50 as wrapped by
// models/dino/backbone.py:62-72,97-106; FrozenBN folded into W and b): K = 64 / 128 / 256 input channels, N = 256 / 512 / 1024 output
// channels, M = B x H x W pixels.  At these shapes the operation is an HBM stream -- per output pixel K x 2 bytes of A, N x 2 of the
// residual in and N x 2 out -- that the tiled kernel (gemm.hip) runs at 3.4-4.4 TB/s: each 128 x 128 tile ends in an epilogue whose
// residual loads are two serialised HBM round trips.  Here, as in gemm_k256.hip's proj_ln_k256_kernel:
//   * the WEIGHT is the resident operand: wave w of 8 keeps 32 NP output channels x K as MFMA A-fragments in registers for the whole
//     kernel (N = 256 NP per launch column; wider layers run N / (256 NP) column slices as blockIdx.y);
//   * TOKENS stream: a 64-token tile of A (64 x K) and of R (64 x 256 NP) is DMA'd global -> LDS (global_load_lds_dwordx4, 8 rows x 128 B
//     per instruction, full lines) through an NS-stage ring, the 16-byte chunks permuted on the source side (chunk c of row r lands in
//     slot c ^ (r & 7)) so the MFMA B-fragment reads and the 16-byte residual reads are conflict-free;
//   * W's rows are assigned to MFMA rows so that a lane's two accumulator tiles of a pair are 8 CONSECUTIVE channels (pair p of wave w,
//     tile e, MFMA row m <-> channel 32 (w NP + p) + 8 (m >> 2) + 4 e + (m & 3)): the residual is one 16-byte LDS read and the result
//     one 16-byte store per token and pair, no lane exchange;
//   * ONE barrier per 64 tokens (publishes the tile, frees the stage the next DMA overwrites); DMA groups and stores share the in-order
//     vmcnt counter and are counted by hand so that a tile's stores are never waited for.
#include "dtlr_common.h"
#include <stdlib.h>

namespace dtlr {

typedef __attribute__((ext_vector_type(8))) h16_hw_t kr_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float kr_f32x4_t;

constexpr int KR_TOK = 64;

__device__ __forceinline__ void kr_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint4 kr_load16(const void* p) {
    uint4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ kr_f32x4_t kr_mma(const uint4& a, const uint4& b, kr_f32x4_t c) {
    return DTLR_MFMA_16x16x32_H16(__builtin_bit_cast(kr_bf16x8_t, a), __builtin_bit_cast(kr_bf16x8_t, b), c, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void kr_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// KB = K / 64 (128-byte k blocks per token row), NP = row-tile pairs per wave (output columns per launch column = 256 NP), NS = stages.
// Wp: fragment order, block (((slice * 8 + wave) * NP + p) * 2 + e) * KS + ks (KS = 2 KB k-steps of 32) = 64 lanes x 8 elements.
// NBR = 128-byte blocks of a residual row that exist (0 = no residual; 4 NP = the full launch column; 6 with NP = 2 = a 384-channel
// output computed as a zero-padded 512-channel column: the pairs beyond n_valid are skipped).  res_rows > 0: the residual is ONE
// [res_rows, n_valid] matrix shared by the n_img = M / res_rows images (row m pairs with row m % res_rows; the encoder's pos . W^T
// term), and tiles are walked position-major (tile t = position tile t / n_img of image t % n_img) so a workgroup, and the band of
// tiles an XCD owns, re-read a few residual tiles out of L2.
//
// CAT: the K axis is the concatenation of TWO activation matrices, A [M, K/2] | A2 [M, K/2] (k blocks KB/2.. come from A2): the first
// bottleneck of layer1, whose shortcut is itself a 1x1 convolution of the block input -- relu(conv3(t) + b3 + downsample(x) + bd) is ONE
// GEMM over [t | x] with the weights [W3 | Wd] and the bias b3 + bd: no shortcut map is written (268 MB at B = 32) or read back.
// NQ2 > 0 (needs NP = 1: a 256-channel launch column = the whole output row): the NEXT bottleneck's first 1x1 convolution runs on the
// tile while it is on chip -- C2 = relu(C W2^T + b2), N2 = 64 NQ2 channels.  The rounded 16-bit results of a tile are also written to an
// LDS image laid out exactly like a K = 256 activation tile (so they are read back as MFMA B-fragments with the same conflict-free
// formula), a second barrier publishes it, and wave w computes token tile w & 3 x channel pairs NQ2 (w >> 2) .. of C2 with W2's fragments
// resident in registers (Wp2 = dtlr_gemm_kres_pack_weights of W2 [N2, 256]: the zero-padded 256-column image the unfused launch takes).
// Same operands, same k order as the unfused launch on the stored C: C2 is bit-identical to it; the 268 MB read of C is gone.
// KB1 > 0 generalises CAT: k blocks 0 .. KB1 - 1 come from A [M, 64 KB1], the rest from A2 (64 (KB - KB1) channels per row).  With
// s2 = {Hout, Wout, Hin, Win} (Wout > 0) row m = (b, i, j) of the Hout x Wout output grid reads A2 at pixel (b, 2 i, 2 j) of an
// Hin x Win map: the STRIDED 1x1 shortcut convolution of layer2's first bottleneck (`downsample`: conv1x1 stride 2) as K columns
// 128..383 of its tail GEMM -- neither the 512-channel shortcut map nor its gather launch exist any more.
struct KrS2 { int hout, wout, hin, win; };
template <int KB, int NP, int NS, int NBR, int KB1 = 0, int NQ2 = 0>
__global__ __launch_bounds__(512, NQ2 > 0 ? 1 : 2) void gemm_kres_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias, const uint16_t* __restrict__ R,
    uint16_t* __restrict__ C, int ld, int M, int tiles_per_wg, int relu, int n_valid, int res_rows, int n_img,
    const uint16_t* __restrict__ A2 = nullptr, const uint16_t* __restrict__ Wp2 = nullptr, const float* __restrict__ bias2 = nullptr,
    uint16_t* __restrict__ C2 = nullptr, KrS2 s2 = KrS2{0, 0, 0, 0})
{
    constexpr bool HAS_R = NBR > 0;
    constexpr bool CAT = KB1 > 0;
    static_assert(NQ2 == 0 || NP == 1, "the fused second GEMM reads whole 256-channel rows of the tile");
    static_assert(KB1 < KB, "at least one k block from the second source");
    extern __shared__ __attribute__((aligned(16))) unsigned char kr_smem[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)kr_smem;
    constexpr int K = 64 * KB, KS = 2 * KB, NC = 256 * NP, NB = NBR;               // NB = 128-byte blocks per residual row
    constexpr int A_BYTES = KR_TOK * K * 2, STAGE = A_BYTES + KR_TOK * NB * 128;
    constexpr int G = KB + (HAS_R ? NB : 0);                                       // DMA instructions per wave per tile
    constexpr int E = 4 * NP + NQ2;                                                // stores per wave per tile (+ the second GEMM's)
    constexpr int YT = NS * STAGE;                                                 // the tile's 16-bit output image [64 tokens][256 channels], 32 KB
    constexpr int N2 = 64 * NQ2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = lane & 15, g = lane >> 4;
    const int col0 = (int)blockIdx.y * NC;                                         // this launch column's first output channel
    const int ntiles = (M + KR_TOK - 1) / KR_TOK;
    // workgroup b runs on XCD b % 8; logical ids are contiguous inside an XCD
    const int nwg = (int)gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = (int)blockIdx.x & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ((int)blockIdx.x >> 3);
    const int t_begin = logical * tiles_per_wg;
    const int t_end = min(t_begin + tiles_per_wg, ntiles);
    if (t_begin >= t_end) return;
    const int nt = t_end - t_begin;
    auto row0 = [&](int t) -> long {
        if (n_img > 0) { const int pt = t / n_img; return (long)(t - pt * n_img) * res_rows + (long)pt * KR_TOK; }
        return (long)t * KR_TOK;
    };

    // DMA of tile t into a stage: wave w moves row group w (token rows 8 w .. 8 w + 7): its KB blocks of A and NB blocks of R
    const int dr = lane >> 3, dc = (lane & 7) ^ dr;
    auto issue = [&](int t, int slot) {
        const long tok = min(row0(t) + wave * 8 + dr, (long)M - 1);
        const unsigned dst = lds_base + (unsigned)(slot * STAGE);
        long tok2 = tok;
        if constexpr (CAT) {
            if (s2.wout > 0) {                                  // (b, i, j) of the output grid -> pixel (b, 2 i, 2 j) of the input map
                const int hw = s2.hout * s2.wout, ti = (int)tok, bi = ti / hw, rem = ti - bi * hw, ii = rem / s2.wout, jj = rem - ii * s2.wout;
                tok2 = ((long)bi * s2.hin + 2 * ii) * s2.win + 2 * jj;
            }
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const uint16_t* src = A + tok * K + kb * 64 + dc * 8;
            if constexpr (CAT) src = kb < KB1 ? A + tok * (64 * KB1) + kb * 64 + dc * 8 : A2 + tok2 * (64 * (KB - KB1)) + (kb - KB1) * 64 + dc * 8;
            kr_glds16(src, dst + (unsigned)((wave * KB + kb) * 1024));
        }
        if constexpr (HAS_R) {
            const long rrow = n_img > 0 ? (long)(t / n_img) * KR_TOK + wave * 8 + dr : tok;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                kr_glds16(R + rrow * (long)(n_img > 0 ? n_valid : ld) + col0 + nb * 64 + dc * 8, dst + (unsigned)(A_BYTES + (wave * NB + nb) * 1024));
        }
    };
    // prologue: NS - 1 tiles in flight, then the resident operand
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nt) issue(t_begin + s, s);
    uint4 wf[NP][2][KS];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                wf[p][e][ks] = kr_load16(Wp + ((long)((((int)blockIdx.y * 8 + wave) * NP + p) * 2 + e) * KS + ks) * 512 + lane * 8);
    float bs[NP][8];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[p][e] = (bias && col0 + 32 * (wave * NP + p) < n_valid) ? bias[col0 + 32 * (wave * NP + p) + 8 * g + e] : 0.f;
    // second GEMM: wave w owns token tile w & 3 and the channel pairs NQ2 (w >> 2) + j of C2
    uint4 w2f[NQ2 > 0 ? NQ2 : 1][2][8];
    float bs2[NQ2 > 0 ? NQ2 : 1][8];
    if constexpr (NQ2 > 0) {
#pragma unroll
        for (int j = 0; j < NQ2; ++j) {
            const int q2 = NQ2 * (wave >> 2) + j;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) w2f[j][e][ks] = kr_load16(Wp2 + ((long)(q2 * 2 + e) * 8 + ks) * 512 + lane * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) bs2[j][e] = bias2 ? bias2[32 * q2 + 8 * g + e] : 0.f;
        }
    }
    kr_wait<0>();

    // B-fragment of token tile tt, k-step ks: row group 2 tt + (n >> 3), block ks >> 1, row n & 7, slot (4 (ks & 1) + g) ^ (n & 7)
    const unsigned rdA = (unsigned)((n >> 3) * KB * 1024 + (n & 7) * 128);
    const unsigned sw0 = (unsigned)((g ^ (n & 7)) * 16), sw1 = (unsigned)(((4 + g) ^ (n & 7)) * 16);
    // residual chunk of pair p: channels 32 (w NP + p) + 8 g .. + 7 -> block (w NP + p) >> 1, chunk 4 ((w NP + p) & 1) + g
    unsigned rdR[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int q = wave * NP + p;
        rdR[p] = (unsigned)(A_BYTES + ((n >> 3) * NB + (q >> 1)) * 1024 + (n & 7) * 128 + (((4 * (q & 1) + g) ^ (n & 7)) * 16));
    }

    // the output image: channel block q = wave (32 channels) of token n of tile tt is chunk 4 (q & 1) + g of k block q >> 1 -- the address the
    // second GEMM's B-fragment of k-step q reads; its own reads: token tile w & 3, k-step ks
    const unsigned ywr = (unsigned)(YT + (n >> 3) * 4096 + (wave >> 1) * 1024 + (n & 7) * 128 + (((4 * (wave & 1) + g) ^ (n & 7)) * 16));
    const unsigned yrd = (unsigned)(YT + (wave & 3) * 8192 + (n >> 3) * 4096 + (n & 7) * 128);

    for (int i = 0; i < nt; ++i) {
        const int t = t_begin + i;
        const int slot = i % NS;
        __builtin_amdgcn_s_barrier();                         // tile i published by every wave; stage (i - 1) % NS no longer read
        if (i + NS - 1 < nt) issue(t + NS - 1, (i + NS - 1) % NS);

        kr_f32x4_t acc[NP][2][4];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc[p][e][tt] = kr_f32x4_t{0.f, 0.f, 0.f, 0.f};
        const unsigned char* sb = kr_smem + slot * STAGE;
        const long trow = row0(t);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 bf[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                bf[tt] = *reinterpret_cast<const uint4*>(sb + rdA + ((ks & 1) ? sw1 : sw0) + tt * (2 * KB * 1024) + (ks >> 1) * 1024);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (col0 + 32 * (wave * NP + p) >= n_valid) continue;                                   // zero-padded column: nothing to compute
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc[p][e][tt] = kr_mma(wf[p][e][ks], bf[tt], acc[p][e][tt]);
            }
        }
        // ---- epilogue: + bias + residual (one 16-byte LDS read), ReLU, one 16-byte store per token and pair ---------------------
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const long tok = trow + tt * 16 + n;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (col0 + 32 * (wave * NP + p) >= n_valid) continue;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[p][e >> 2][tt][e & 3] + bs[p][e];
                if constexpr (HAS_R) {
                    const uint4 rr = *reinterpret_cast<const uint4*>(sb + rdR[p] + tt * (2 * NB * 1024));
                    const uint32_t rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        v[e] += (e & 1) ? h16_hi(rw[e >> 1]) : h16_lo(rw[e >> 1]);
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                const uint4 pk = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
                if (tok < M) *reinterpret_cast<uint4*>(C + tok * (long)ld + col0 + 32 * (wave * NP + p) + 8 * g) = pk;
                if constexpr (NQ2 > 0) *reinterpret_cast<uint4*>(kr_smem + ywr + tt * 8192) = pk;
            }
        }
        if constexpr (NQ2 > 0) {
            // ---- the next block's 1x1 convolution on the tile: C2[64, N2] = relu(Y[64, 256] W2^T + b2) -------------------------------
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // the tile's image is complete (the next tile's is written after the next top barrier)
            asm volatile("" ::: "memory");
            kr_f32x4_t acc2[NQ2][2];
#pragma unroll
            for (int j = 0; j < NQ2; ++j) { acc2[j][0] = kr_f32x4_t{0.f, 0.f, 0.f, 0.f}; acc2[j][1] = kr_f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const uint4 yb = *reinterpret_cast<const uint4*>(kr_smem + yrd + ((ks & 1) ? sw1 : sw0) + (ks >> 1) * 1024);
#pragma unroll
                for (int j = 0; j < NQ2; ++j) {
                    acc2[j][0] = kr_mma(w2f[j][0][ks], yb, acc2[j][0]);
                    acc2[j][1] = kr_mma(w2f[j][1][ks], yb, acc2[j][1]);
                }
            }
            const long tok2 = trow + (wave & 3) * 16 + n;
#pragma unroll
            for (int j = 0; j < NQ2; ++j) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(acc2[j][e >> 2][e & 3] + bs2[j][e], 0.f);
                if (tok2 < M)
                    *reinterpret_cast<uint4*>(C2 + tok2 * (long)N2 + 32 * (NQ2 * (wave >> 2) + j) + 8 * g) =
                        make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
            }
        }
        // my pieces of tile i + 1 must have landed before the next barrier.  Issued since: the stores of tile i + 2 - NS, then a DMA group
        // and a tile's stores in each of the NS - 2 iterations after it (a ragged tile -- fewer stores -- is always a workgroup's last)
        // (zero-padded column: a wave with fewer valid pairs issues fewer stores than E and waits with vmcnt(0): stricter, never wrong)
        if (i + NS - 1 < nt) {
            if (col0 + 32 * (wave * NP + NP) > n_valid) kr_wait<0>();
            else kr_wait<E + (NS - 2) * (G + E)>();
        } else kr_wait<0>();
    }
}

// W [N, K] row-major bf16 (host) -> fragment order (host, N * K elements): slice = 256 NP output channels
// block ((((slice * 8 + wave) * NP + p) * 2 + e) * KS + ks) lane (m, g) <- W[256 NP slice + 32 (wave NP + p) + 8 (m >> 2) + 4 e + (m & 3)][32 ks + 8 g ..]
static int kres_pack(const unsigned short* w_host, unsigned short* wp_host, int N, int K, int NP, int n_rows)
{
    const int KS = K / 32, nslice = N / (256 * NP);
    for (int sl = 0; sl < nslice; ++sl)
        for (int wave = 0; wave < 8; ++wave)
            for (int p = 0; p < NP; ++p)
                for (int e = 0; e < 2; ++e)
                    for (int ks = 0; ks < KS; ++ks)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int m = lane & 15, g = lane >> 4;
                            const int row = 256 * NP * sl + 32 * (wave * NP + p) + 8 * (m >> 2) + 4 * e + (m & 3);
                            const long blk = ((((long)sl * 8 + wave) * NP + p) * 2 + e) * KS + ks;
                            for (int x = 0; x < 8; ++x)
                                wp_host[(blk * 64 + lane) * 8 + x] = row < n_rows ? w_host[(long)row * K + ks * 32 + g * 8 + x] : (unsigned short)0;
                        }
    return DTLR_OK;
}

// N a multiple of 256, or 64 / 128 / 192 (packed as ONE zero-padded 256-channel column: wp_host then holds 256 * K elements)
extern "C" int dtlr_gemm_kres_pack_weights(const unsigned short* w_host, unsigned short* wp_host, int N, int K)
{
    if (!w_host || !wp_host) return DTLR_EINVAL;
    if ((K != 64 && K != 128 && K != 256 && K != 384) || N <= 0 || (N & 63) || (N > 256 && (N & 255))) return DTLR_ESHAPE;     // 384: dtlr_gemm_kres_cat_s2
    const int Np = N < 256 ? 256 : N;
    return kres_pack(w_host, wp_host, Np, K, (Np % 512 == 0 && K <= 128) ? 2 : 1, N);
}

// W [384, 256] (host) -> the image dtlr_gemm_kres_bcast384 takes: a 512-channel column (two row-tile pairs per wave), rows 384..511 zero;
// wp_host holds 512 * 256 elements.
extern "C" int dtlr_gemm_kres_pack_weights_bcast384(const unsigned short* w_host, unsigned short* wp_host)
{
    if (!w_host || !wp_host) return DTLR_EINVAL;
    return kres_pack(w_host, wp_host, 512, 256, 2, 384);
}

static int kres_launch(const void* A, const void* Wp, const float* bias, const void* R, void* C, int M, int N, int K, int relu,
                       int n_valid, int res_rows, hipStream_t st)
{
    const int NP = (N % 512 == 0 && K <= 128) || res_rows > 0 ? 2 : 1;
    const int nslice = N / (256 * NP);
    const int ntiles = (M + KR_TOK - 1) / KR_TOK;
    const int ncu = 256;
    // one workgroup per CU (the ring takes most of the LDS); the column slices of one token range get workgroup ids a multiple of 8
    // apart (grid.x a multiple of 8 at the shapes of the path): same XCD, so the A tile is fetched from HBM once
    int per_x = (ntiles * nslice + ncu - 1) / ncu;
    if (per_x < 1) per_x = 1;
    const int gx = (ntiles + per_x - 1) / per_x;
    const int n_img = res_rows > 0 ? M / res_rows : 0;
    const int ld = n_valid;                                    // row stride of R and C (N itself except for the zero-padded column forms)
#define KR_LAUNCH(KB_, NP_, NS_, NBR_)                                                             \
    {                                                                                              \
        constexpr int lds_ = NS_ * (KR_TOK * 64 * KB_ * 2 + KR_TOK * NBR_ * 128);                  \
        static DevOnce once;                                                                       \
        if (once.first()) { (void)hipFuncSetAttribute((const void*)gemm_kres_kernel<KB_, NP_, NS_, NBR_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_); (void)hipGetLastError(); } \
        hipLaunchKernelGGL((gemm_kres_kernel<KB_, NP_, NS_, NBR_>), dim3(gx, nslice), dim3(512), lds_, st, (const uint16_t*)A, (const uint16_t*)Wp, bias, \
                           (const uint16_t*)R, (uint16_t*)C, ld, M, per_x, relu, n_valid, res_rows, n_img); \
    }
    if (res_rows > 0) KR_LAUNCH(4, 2, 2, 6)                                 // K = 256, 384 channels as a zero-padded 512 column
    else if (K == 64) {
        if (NP == 2) { if (R) KR_LAUNCH(1, 2, 2, 8) else KR_LAUNCH(1, 2, 4, 0) }
        else { if (R) KR_LAUNCH(1, 1, 4, 4) else KR_LAUNCH(1, 1, 4, 0) }
    } else if (K == 128) {
        if (NP == 2) { if (R) KR_LAUNCH(2, 2, 2, 8) else KR_LAUNCH(2, 2, 4, 0) }
        else { if (R) KR_LAUNCH(2, 1, 3, 4) else KR_LAUNCH(2, 1, 4, 0) }
    } else {
        if (R) KR_LAUNCH(4, 1, 2, 4) else KR_LAUNCH(4, 1, 4, 0)
    }
#undef KR_LAUNCH
    return check_launch();
}

// A [M, K] bf16 (K = 64 / 128 / 256), Wp from dtlr_gemm_kres_pack_weights, bias [N] fp32 or null, R [M, N] bf16 or null (row stride N),
// C [M, N] bf16, relu: 0 / 1 (applied after the residual).  N a multiple of 256.
extern "C" int dtlr_gemm_kres(const void* A, const void* Wp, const float* bias, const void* R, void* C, int M, int N, int K, int relu, void* stream)
{
    clear_stale_error();
    if (!A || !Wp || !C) return DTLR_EINVAL;
    if (M <= 0) return DTLR_EINVAL;
    if ((K != 64 && K != 128 && K != 256) || N <= 0 || (N & 63) || (N > 256 && (N & 255))) return DTLR_ESHAPE;
    if (N < 256 && R) return DTLR_ESHAPE;                       // the zero-padded column form has no residual tile
    return kres_launch(A, Wp, bias, R, C, M, N < 256 ? 256 : N, K, relu, N, 0, (hipStream_t)stream);
}

// The bottleneck tails of layer1 with their neighbours fused (the kernel's CAT / NQ2 notes):  N = 256 output channels, 64-channel inputs.
//   C  = relu?( [A | A2] Wp^T + bias (+ R) )      A [M, 64]; A2 [M, 64] or null; R [M, 256] or null -- exactly one of A2 and R;
//                                                 Wp = dtlr_gemm_kres_pack_weights of W [256, 128] = [W3 | Wd] with A2, of W [256, 64] with R
//   C2 = relu( C Wp2^T + bias2 )                  Wp2 = dtlr_gemm_kres_pack_weights of W2 [N2, 256], N2 = 64 or 128; null: no second GEMM (A2 form only)
// C2 is bit-identical to dtlr_gemm_kres run on the stored C; with R, C is bit-identical to dtlr_gemm_kres; with A2, the shortcut
// convolution is accumulated in fp32 instead of being rounded to 16 bits first.
extern "C" int dtlr_gemm_kres_chain(const void* A, const void* A2, const void* Wp, const float* bias, const void* R, void* C, int M, int relu,
                                    const void* Wp2, const float* bias2, void* C2, int N2, void* stream)
{
    clear_stale_error();
    if (!A || !Wp || !C || M <= 0) return DTLR_EINVAL;
    if ((A2 != nullptr) == (R != nullptr)) return DTLR_EINVAL;
    if (Wp2 ? (!C2 || (N2 != 64 && N2 != 128)) : (!A2)) return DTLR_EINVAL;
    if (Wp2 && A2 && N2 != 64) return DTLR_ESHAPE;
    const int ntiles = (M + KR_TOK - 1) / KR_TOK;
    int per_x = (ntiles + 255) / 256;
    if (per_x < 1) per_x = 1;
    const int gx = (ntiles + per_x - 1) / per_x;
    hipStream_t st = (hipStream_t)stream;
#define KR_CHAIN(KB_, NS_, NBR_, KB1_, NQ2_)                                                       \
    {                                                                                              \
        constexpr int lds_ = NS_ * (KR_TOK * 64 * KB_ * 2 + KR_TOK * NBR_ * 128) + (NQ2_ > 0 ? KR_TOK * 512 : 0); \
        static DevOnce once;                                                                       \
        if (once.first()) { (void)hipFuncSetAttribute((const void*)gemm_kres_kernel<KB_, 1, NS_, NBR_, KB1_, NQ2_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_); (void)hipGetLastError(); } \
        hipLaunchKernelGGL((gemm_kres_kernel<KB_, 1, NS_, NBR_, KB1_, NQ2_>), dim3(gx, 1), dim3(512), lds_, st, (const uint16_t*)A, (const uint16_t*)Wp, bias, \
                           (const uint16_t*)R, (uint16_t*)C, 256, M, per_x, relu, 256, 0, 0, (const uint16_t*)A2, (const uint16_t*)Wp2, bias2, (uint16_t*)C2); \
    }
    if (A2) { if (Wp2) KR_CHAIN(2, 4, 0, 1, 1) else KR_CHAIN(2, 4, 0, 1, 0) }
    else if (N2 == 64) KR_CHAIN(1, 3, 4, 0, 1)
    else KR_CHAIN(1, 3, 4, 0, 2)
#undef KR_CHAIN
    return check_launch();
}

// layer2's first bottleneck tail with its STRIDED shortcut convolution as extra K columns (the kernel's KB1 / s2 notes):
//     C[(b, i, j), :] = relu?( [A[(b, i, j), :] | X[b, 2 i, 2 j, :]] Wp^T + bias )
// A [B Hout Wout, 128] (the 3x3 stride-2 convolution's output), X [B, Hin, Win, 256] (the block input), Hout = (Hin - 1) / 2 + 1 (same for
// W), Wp = dtlr_gemm_kres_pack_weights of W [512, 384] = [W3 | Wd], bias = b3 + bd, C [B Hout Wout, 512].
extern "C" int dtlr_gemm_kres_cat_s2(const void* A, const void* X, const void* Wp, const float* bias, void* C, int B, int Hin, int Win, int relu,
                                     void* stream)
{
    clear_stale_error();
    if (!A || !X || !Wp || !C || B <= 0 || Hin <= 0 || Win <= 0) return DTLR_EINVAL;
    const int Hout = (Hin - 1) / 2 + 1, Wout = (Win - 1) / 2 + 1;
    const long Ml = (long)B * Hout * Wout;
    if (Ml >= (1L << 31) || (long)B * Hin * Win >= (1L << 31)) return DTLR_ESHAPE;
    const int M = (int)Ml, nslice = 2;
    const int ntiles = (M + KR_TOK - 1) / KR_TOK;
    int per_x = (ntiles * nslice + 255) / 256;
    if (per_x < 1) per_x = 1;
    const int gx = (ntiles + per_x - 1) / per_x;
    constexpr int lds_ = 3 * (KR_TOK * 64 * 6 * 2);
    static DevOnce once;
    if (once.first()) { (void)hipFuncSetAttribute((const void*)gemm_kres_kernel<6, 1, 3, 0, 2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_); (void)hipGetLastError(); }
    hipLaunchKernelGGL((gemm_kres_kernel<6, 1, 3, 0, 2, 0>), dim3(gx, nslice), dim3(512), lds_, (hipStream_t)stream, (const uint16_t*)A, (const uint16_t*)Wp, bias,
                       (const uint16_t*)nullptr, (uint16_t*)C, 512, M, per_x, relu, 512, 0, 0, (const uint16_t*)X, (const uint16_t*)nullptr, (const float*)nullptr,
                       (uint16_t*)nullptr, KrS2{Hout, Wout, Hin, Win});
    return check_launch();
}

// The encoder's [offsets | attention logits] projection with the position term as a row-broadcast residual:
//     C[m, :] = A[m, :] W^T + R[m % res_rows, :]        A [M, 256], W [384, 256], R [res_rows, 384], C [M, 384], all bf16
// (ops/modules/ms_deform_attn.py:97-98 applied to query = src + pos: (src + pos) W^T + b = src W^T + (pos W^T + b); the second term is
// the same for every image of an unpadded batch).  Wp = dtlr_gemm_kres_pack_weights of W zero-padded to 512 rows (N = 512, K = 256).
// res_rows a multiple of 64 and M a multiple of res_rows (DTLR_ESHAPE otherwise: use dtlr_gemm_k256).
extern "C" int dtlr_gemm_kres_bcast384(const void* A, const void* Wp, const void* R, int res_rows, void* C, int M, void* stream)
{
    clear_stale_error();
    if (!A || !Wp || !R || !C) return DTLR_EINVAL;
    if (M <= 0 || res_rows <= 0) return DTLR_EINVAL;
    if ((res_rows % KR_TOK) || (M % res_rows) || (long)res_rows * 384 * 2 >= (1L << 31)) return DTLR_ESHAPE;
    return kres_launch(A, Wp, nullptr, R, C, M, 512, 256, 0, 384, res_rows, (hipStream_t)stream);
}

}  // namespace dtlr
