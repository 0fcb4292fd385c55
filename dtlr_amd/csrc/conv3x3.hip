// 3x3 / stride 1 / pad 1 convolution, bf16 NHWC, with the input patch RESIDENT in LDS (the middle convolution of the ResNet
// bottlenecks of layer1-3: torchvision resnet50 `conv2` + folded FrozenBN + ReLU as run by models/dino/backbone.py:62-72,97-106).
//
// gemm.hip's implicit-GEMM form gathers, for every 128-byte K slab, the slab's 128 token rows again from L2: the nine taps re-read
// every input pixel nine times per channel tile, and at these shapes the kernel is bound by L2 -> LDS operand delivery (conv 256 -> 256
// on 32 x 8 x 128: 1180 KB per 128 x 128 output tile, 0.6 GB per launch at ~11 TB/s = the 51 us it takes; 758 TFLOP/s).  Here a
// workgroup owns an 8 x 16 pixel tile of one image and 64 or 128 output channels:
//   * the (8 + 2) x (16 + 2) input patch is DMA'd ONCE into LDS (global_load_lds_dwordx4, zero line outside the image), as Cin / 64
//     planes of [pixel][64 channels]: a plane row is the 128-byte K slab of one pixel, so tap (dy, dx) of channel block cb is the same
//     plane read at pixel offset 18 dy + dx -- the MFMA B-fragments of all nine taps come straight out of the patch
//     (16-byte chunk c of pixel p sits in slot c ^ (p & 7): conflict-free ds_read_b128 for any tap shift);
//   * only the WEIGHTS stream: slab (tap, cb) = [BN channels][128 bytes], DMA'd through a 4-stage ring with the same swizzle;
//   * 4 waves, each a 64-pixel x BN/2-channel sub-tile (16x16x32 MFMAs, fp32 accumulators for the whole K sweep), one barrier per slab;
//   * bias + ReLU epilogue with the paired 16-byte stores of gemm.hip.
// L2 -> LDS bytes per 128-pixel x 128-channel tile at Cin = 256: 92 KB patch + 36 x 16 KB weights = 668 KB (0.57x).
#include "dtlr_common.h"
#include <stdlib.h>

namespace dtlr {

typedef __attribute__((ext_vector_type(8))) h16_hw_t cp_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float cp_f32x4_t;

constexpr int CP_TH = 8, CP_TW = 16, CP_PW = CP_TW + 2, CP_PH = CP_TH + 2, CP_NPIX = CP_PW * CP_PH;      // 18 x 10 = 180 patch pixels
constexpr int CP_PLANE = 184 * 128;                         // 23 DMA blocks of 8 pixels x 128 B per channel-block plane
constexpr int CP_NS = 4;                                    // weight ring stages
__device__ __attribute__((aligned(16))) unsigned int g_cp_zero_line[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void cp_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void cp_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// CB = Cin / 64, BN = output channels per workgroup (64 or 128).  Wt: [Cout][3][3][Cin] bf16.  grid = (ceil(W / 16), ceil(H / 8), B * Cout / BN)
// NW = 4 or 8 waves: wave w owns the 64-pixel half w & 1 and channel slice w >> 1 of BN / (NW / 2) channels.  With one workgroup per CU
// (Cin >= 128: the patch and the ring take most of the LDS) four waves leave one wave per SIMD and nothing to hide a fragment read's
// latency behind; eight waves split the same MFMA work two per SIMD.
template <int CB, int BN, int NW>
__global__ __launch_bounds__(64 * NW, 1) void conv3x3_patch_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wt,
                                                               const float* __restrict__ bias, uint16_t* __restrict__ Y,
                                                               int H, int W, int Cout, int relu)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cp_smem[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)cp_smem;
    constexpr int Cin = 64 * CB, K = 9 * Cin, NSL = 9 * CB, WST = BN * 128, PATCH = CB * CP_PLANE, NSLICE = NW / 2, CW = BN / NSLICE, CI = CW / 16,
                  P = BN / (8 * NW);
    static_assert(CI >= 2 && (CI & 1) == 0 && P >= 1, "channel slice: an even number of 16-channel tiles; at least one weight block per wave");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = lane & 15, g = lane >> 4;
    const int nN = Cout / BN;
    const int b = (int)blockIdx.z / nN, n0 = ((int)blockIdx.z % nN) * BN;
    const int x0 = (int)blockIdx.x * CP_TW, y0 = (int)blockIdx.y * CP_TH;
    const uint16_t* Xb = X + (long)b * H * W * Cin;

    // ---- the input patch, once: plane cb, block j = 8 patch pixels x 128 B; this wave takes blocks wave, wave + NW, ... ---------
    const int pr = lane >> 3, slot = lane & 7;
    for (int blk = wave; blk < CB * 23; blk += NW) {
        const int cb = blk / 23, j = blk - cb * 23;
        const int pi = min(8 * j + pr, CP_NPIX - 1);                     // (the last block's tail rows land in the plane's padding)
        const int py = pi / CP_PW, px = pi - py * CP_PW;
        const int y = y0 - 1 + py, x = x0 - 1 + px;
        const bool ok = y >= 0 && y < H && x >= 0 && x < W;
        const int c = slot ^ ((8 * j + pr) & 7);
        const void* src = ok ? (const void*)(Xb + ((long)y * W + x) * Cin + cb * 64 + c * 8) : (const void*)g_cp_zero_line;
        cp_glds16(src, lds_base + (unsigned)(cb * CP_PLANE + j * 1024));
    }
    // ---- weight slabs: slab s = (tap = s / CB, cb = s % CB): rows n0 .. n0 + BN - 1, 128 B each; this wave issues blocks u = wave + 4 i
    auto issue_w = [&](int s) {
        const int tap = s / CB, cb = s - tap * CB;
        const unsigned dst = lds_base + (unsigned)(PATCH + (s % CP_NS) * WST);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int u = wave + NW * i, r = 8 * u + pr;
            const int c = slot ^ pr;                                      // (r & 7) == pr
            cp_glds16(Wt + (long)(n0 + r) * K + tap * Cin + cb * 64 + c * 8, dst + (unsigned)(u * 1024));
        }
    };
#pragma unroll
    for (int s = 0; s < CP_NS - 1; ++s)
        if (s < NSL) issue_w(s);
    cp_wait<0>();

    const int wm = wave & 1, wn = wave >> 1;
    cp_f32x4_t acc[CI][4];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) acc[ci][ti] = cp_f32x4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned wrow = (unsigned)((wn * CW + n) * 128);                   // this lane's weight row inside a stage (+ ci * 2048)

    for (int s = 0; s < NSL; ++s) {
        __builtin_amdgcn_s_barrier();                                        // slab s published; stage (s - 1) % NS no longer read
        if (s + CP_NS - 1 < NSL) issue_w(s + CP_NS - 1);
        const int tap = s / CB, cb = s - tap * CB;
        const int dy = tap / 3, dx = tap - 3 * dy;
        const unsigned char* wst = cp_smem + PATCH + (s % CP_NS) * WST + wrow;
        const unsigned char* pl = cp_smem + cb * CP_PLANE;
        const int pi0 = (wm * 4 + dy) * CP_PW + n + dx;                      // patch pixel of this lane's token in tile row ti = 0
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
            uint4 wf[CI], xf[4];
            const unsigned wsw = (unsigned)((((kq * 4 + g) ^ (n & 7))) * 16);
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) wf[ci] = *reinterpret_cast<const uint4*>(wst + ci * 2048 + wsw);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                const int pi = pi0 + ti * CP_PW;
                xf[ti] = *reinterpret_cast<const uint4*>(pl + pi * 128 + (((kq * 4 + g) ^ (pi & 7)) * 16));
            }
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti)
                    acc[ci][ti] = DTLR_MFMA_16x16x32_H16(__builtin_bit_cast(cp_bf16x8_t, wf[ci]), __builtin_bit_cast(cp_bf16x8_t, xf[ti]),
                                                                          acc[ci][ti], 0, 0, 0);
        }
        // my pieces of slab s + 1 must have landed before the next barrier; the (NS - 2) slabs issued after it may stay in flight
        if (s + CP_NS - 1 < NSL) cp_wait<(CP_NS - 2) * P>();
        else cp_wait<0>();
    }

    // ---- epilogue: + bias, ReLU, bf16, paired 16-byte stores (lane (n, g): channels 16 ci + 4 g + r of token n of tile row ti) -------
    float4 bv[CI];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
        bv[ci] = bias ? *reinterpret_cast<const float4*>(bias + n0 + wn * CW + ci * 16 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
        const int y = y0 + wm * 4 + ti, x = x0 + n;
        const bool live = y < H && x < W;
        uint16_t* dstrow = Y + (((long)b * H + y) * W + x) * Cout + n0 + wn * CW;
        uint32_t pk_lo = 0, pk_hi = 0;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
            float v[4] = {acc[ci][ti][0] + bv[ci].x, acc[ci][ti][1] + bv[ci].y, acc[ci][ti][2] + bv[ci].z, acc[ci][ti][3] + bv[ci].w};
            if (relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            const uint32_t lo = pack_bf16x2(v[0], v[1]), hi = pack_bf16x2(v[2], v[3]);
            if ((ci & 1) == 0) { pk_lo = lo; pk_hi = hi; }
            else {
                // pair the channel tiles (ci - 1, ci): exchange halves between lane rows g and g ^ 1 -> a lane owns 8 consecutive channels
                const auto s0 = __builtin_amdgcn_permlane16_swap(pk_lo, lo, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(pk_hi, hi, false, false);
                if (live) *reinterpret_cast<uint4*>(dstrow + (ci - 1 + (g & 1)) * 16 + 8 * (g >> 1)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
        }
    }
}

// ---- the same structure for the SPLIT-fp32 engine (round 6): fp32 NHWC in / out, every product as three fp16 MFMAs on hi + lo halves --------
// Through the tiled split GEMM (gemm.hip, GT<f32s_t>, implicit-GEMM form) layer1's 3x3 convolutions take 244 us each at B = 32 (134 MB in,
// 134 MB out: 1.1 TB/s; 38.7 GFLOP x 3 products: 0.19 of the 16-bit MFMA peak): every 128-byte K slab gathers its 128 token rows again,
// as fp32 (twice the bytes of the bf16 form per tap), and converts them again -- nine times per input pixel and channel tile.  Here:
//   * the (8 + 2) x (16 + 2) fp32 input patch is read ONCE (two 16-byte loads per lane and 8 channels), split ONCE into fp16 hi and lo
//     halves and written to two LDS planes per 64-channel block with the bf16 kernel's layout (plane row = the 128-byte K slab of one
//     pixel, chunk c of pixel p in slot c ^ (p & 7)): the nine taps read their B-fragments (hi, lo) straight out of the planes;
//   * the WEIGHTS stream from the engine's split slab image (dtlr_split_pack_weights of [Cout][3][3][Cin]: per 32 k of a row 64 B of hi
//     halves then 64 B of lo halves): slab (tap, cb) = [BN channels][256 B], DMA'd through an NS-stage ring, the sixteen 16-byte chunks
//     of a row permuted on the source side (chunk c in slot c ^ (row & 15)): conflict-free A-fragment reads at a 256-byte row pitch;
//   * per (k-quarter, channel tile, pixel row): acc += W_hi X_lo + W_lo X_hi + W_hi X_hi (gemm_k256s.hip's order), fp32 accumulators
//     for the whole K sweep, bias + ReLU, 16-byte fp32 stores.
typedef __attribute__((ext_vector_type(8))) _Float16 cps_f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 cps_f16x2_t;
__device__ __forceinline__ cp_f32x4_t cps_mma(const uint4& a, const uint4& b, cp_f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cps_f16x8_t, a), __builtin_bit_cast(cps_f16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ void cps_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    const cps_f16x2_t a = __builtin_convertvector(f32x2_hw_t{x0, x1}, cps_f16x2_t);
    const cps_f16x2_t b = __builtin_convertvector(f32x2_hw_t{x0 - (float)a[0], x1 - (float)a[1]}, cps_f16x2_t);
    hi = __builtin_bit_cast(uint32_t, a);
    lo = __builtin_bit_cast(uint32_t, b);
}

// CB = Cin / 64, BN = output channels per workgroup, NW waves (wave w: 64-pixel half w & 1, channel slice w >> 1), NS ring stages.
// Wt: the split slab image of [Cout][3][3][Cin] (byte offset of (row, k) slab = (row K + 32 slab) 4).  grid = (ceil(W / 16), ceil(H / 8), B Cout / BN)
template <int CB, int BN, int NW, int NS>
__global__ __launch_bounds__(64 * NW, (CB == 1 && NS <= 2) ? 2 : 1) void conv3x3_patch_f32s_kernel(const float* __restrict__ X, const unsigned char* __restrict__ Wt,
                                                                                                const float* __restrict__ bias, float* __restrict__ Y,
                                                                                                int H, int W, int Cout, int relu)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cp_smem[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)cp_smem;
    constexpr int Cin = 64 * CB, K = 9 * Cin, NSL = 9 * CB, WST = BN * 256, PATCH = 2 * CB * CP_PLANE, NSLICE = NW / 2, CW = BN / NSLICE, CI = CW / 16,
                  P = BN / (4 * NW);                                         // DMA instructions per wave and slab: 4 rows x 256 B each
    static_assert(CI >= 1 && P >= 1 && BN % (4 * NW) == 0, "channel slice / weight blocks per wave");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = lane & 15, g = lane >> 4;
    const int nN = Cout / BN;
    const int b = (int)blockIdx.z / nN, n0 = ((int)blockIdx.z % nN) * BN;
    const int x0 = (int)blockIdx.x * CP_TW, y0 = (int)blockIdx.y * CP_TH;
    const float* Xb = X + (long)b * H * W * Cin;

    // ---- weight slabs first (the DMA flies under the patch conversion): slab s = (tap, cb); this wave issues row groups u = wave + NW i
    const int wr = lane >> 4, wp = lane & 15;                                // row inside a 4-row group, 16-byte slot inside the 256-byte row
    auto issue_w = [&](int s) {
        const int tap = s / CB, cb = s - tap * CB;
        const unsigned dst = lds_base + (unsigned)(PATCH + (s % NS) * WST);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int u = wave + NW * i, r = 4 * u + wr;
            const int c = wp ^ (r & 15);
            cp_glds16(Wt + ((long)(n0 + r) * K + tap * Cin + cb * 64) * 4 + c * 16, dst + (unsigned)(u * 1024));
        }
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < NSL) issue_w(s);

    // ---- the input patch, once: block (cb, j) = 8 patch pixels x 64 channels; lane (pr, slot): pixel 8 j + pr, channels 8 slot .. + 7 ----
    const int pr = lane >> 3, slot = lane & 7;
    for (int blk = wave; blk < CB * 23; blk += NW) {
        const int cb = blk / 23, j = blk - cb * 23;
        const int pp = 8 * j + pr;                                           // plane row (the last block's tail rows are plane padding)
        const int pi = min(pp, CP_NPIX - 1);
        const int py = pi / CP_PW, px = pi - py * CP_PW;
        const int y = y0 - 1 + py, x = x0 - 1 + px;
        const bool ok = y >= 0 && y < H && x >= 0 && x < W && pp < CP_NPIX;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            const float* src = Xb + ((long)y * W + x) * Cin + cb * 64 + slot * 8;
            a = *reinterpret_cast<const float4*>(src);
            c4 = *reinterpret_cast<const float4*>(src + 4);
        }
        uint4 hi, lo;
        cps_split2(a.x, a.y, hi.x, lo.x); cps_split2(a.z, a.w, hi.y, lo.y); cps_split2(c4.x, c4.y, hi.z, lo.z); cps_split2(c4.z, c4.w, hi.w, lo.w);
        unsigned char* dst = cp_smem + (2 * cb) * CP_PLANE + pp * 128 + ((slot ^ (pp & 7)) * 16);
        *reinterpret_cast<uint4*>(dst) = hi;
        *reinterpret_cast<uint4*>(dst + CP_PLANE) = lo;
    }
    cp_wait<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    const int wm = wave & 1, wn = wave >> 1;
    cp_f32x4_t acc[CI][4];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) acc[ci][ti] = cp_f32x4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned wrow = (unsigned)((wn * CW + n) * 256);                   // this lane's weight row inside a stage (+ ci * 4096); row & 15 == n

    for (int s = 0; s < NSL; ++s) {
        __builtin_amdgcn_s_barrier();                                        // slab s (and, at s = 0, the patch) published; stage (s - 1) % NS free
        if (s + NS - 1 < NSL) issue_w(s + NS - 1);
        const int tap = s / CB, cb = s - tap * CB;
        const int dy = tap / 3, dx = tap - 3 * dy;
        const unsigned char* wst = cp_smem + PATCH + (s % NS) * WST + wrow;
        const unsigned char* pl = cp_smem + (2 * cb) * CP_PLANE;
        const int pi0 = (wm * 4 + dy) * CP_PW + n + dx;
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
            uint4 wh[CI], wl[CI], xh[4], xl[4];
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) {
                wh[ci] = *reinterpret_cast<const uint4*>(wst + ci * 4096 + (((kq * 8 + g) ^ n) * 16));
                wl[ci] = *reinterpret_cast<const uint4*>(wst + ci * 4096 + (((kq * 8 + 4 + g) ^ n) * 16));
            }
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                const int pi = pi0 + ti * CP_PW;
                const unsigned char* q = pl + pi * 128 + (((kq * 4 + g) ^ (pi & 7)) * 16);
                xh[ti] = *reinterpret_cast<const uint4*>(q);
                xl[ti] = *reinterpret_cast<const uint4*>(q + CP_PLANE);
            }
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    acc[ci][ti] = cps_mma(wh[ci], xl[ti], acc[ci][ti]);
                    acc[ci][ti] = cps_mma(wl[ci], xh[ti], acc[ci][ti]);
                    acc[ci][ti] = cps_mma(wh[ci], xh[ti], acc[ci][ti]);
                }
        }
        // my pieces of slab s + 1 must have landed before the next barrier; the (NS - 2) slabs issued after it may stay in flight
        if (s + NS - 1 < NSL) cp_wait<(NS - 2) * P>();
        else cp_wait<0>();
    }

    // ---- epilogue: + bias, ReLU, fp32: lane (n, g) stores channels 16 ci + 4 g .. + 3 of pixel n of tile row ti (16 bytes) ----------
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
        const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n0 + wn * CW + ci * 16 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            const int y = y0 + wm * 4 + ti, x = x0 + n;
            float4 v = make_float4(acc[ci][ti][0] + bv.x, acc[ci][ti][1] + bv.y, acc[ci][ti][2] + bv.z, acc[ci][ti][3] + bv.w);
            if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            if (y < H && x < W) *reinterpret_cast<float4*>(Y + (((long)b * H + y) * W + x) * Cout + n0 + wn * CW + ci * 16 + 4 * g) = v;
        }
    }
}

// 1 when dtlr_conv3x3_patch_f32s takes this shape
extern "C" int dtlr_conv3x3_patch_f32s_supported(int Cin, int Cout)
{
    return ((Cin == 64 && Cout > 0 && (Cout % 64) == 0) || (Cin == 128 && Cout > 0 && (Cout % 128) == 0)) ? 1 : 0;
}

// X [B, H, W, Cin] fp32 NHWC; Wt = dtlr_split_pack_weights of [Cout, 3, 3, Cin] (rows = Cout, K = 9 Cin); bias [Cout] fp32 or null;
// Y [B, H, W, Cout] fp32; relu != 0: ReLU after the bias.  3x3 / stride 1 / pad 1.
extern "C" int dtlr_conv3x3_patch_f32s(const float* X, const void* Wt, const float* bias, float* Y, int B, int H, int W, int Cin, int Cout,
                                       int relu, void* stream)
{
    clear_stale_error();
    if (!X || !Wt || !Y) return DTLR_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0) return DTLR_EINVAL;
    if (!dtlr_conv3x3_patch_f32s_supported(Cin, Cout)) return DTLR_ESHAPE;
    const int BN = Cin == 64 ? 64 : 128;
    const long gz = (long)B * (Cout / BN);
    if (gz > 65535) return DTLR_ESHAPE;
    const dim3 grid((unsigned)((W + CP_TW - 1) / CP_TW), (unsigned)((H + CP_TH - 1) / CP_TH), (unsigned)gz);
    hipStream_t st = (hipStream_t)stream;
#define CPS_LAUNCH(CB_, BN_, NW_, NS_)                                                             \
    {                                                                                              \
        constexpr int lds_ = 2 * CB_ * CP_PLANE + NS_ * BN_ * 256;                                 \
        static DevOnce once;                                                                       \
        if (once.first()) { (void)hipFuncSetAttribute((const void*)conv3x3_patch_f32s_kernel<CB_, BN_, NW_, NS_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_); (void)hipGetLastError(); } \
        hipLaunchKernelGGL((conv3x3_patch_f32s_kernel<CB_, BN_, NW_, NS_>), grid, dim3(64 * NW_), lds_, st, X, (const unsigned char*)Wt, bias, Y, H, W, Cout, relu); \
    }
    if (Cin == 64) CPS_LAUNCH(1, 64, 4, 2)          // 46 KB of planes + 2 x 16 KB: two workgroups per CU
    else CPS_LAUNCH(2, 128, 8, 2)                   // 92 KB + 2 x 32 KB = 156 KB: one workgroup of eight waves
#undef CPS_LAUNCH
    return check_launch();
}

// 1 when dtlr_conv3x3_patch_bf16 takes this shape
extern "C" int dtlr_conv3x3_patch_supported(int Cin, int Cout)
{
    return (Cin == 64 || Cin == 128 || Cin == 256) && Cout > 0 && (Cout % 64) == 0 ? 1 : 0;
}

// X [B, H, W, Cin] bf16 NHWC; Wt [Cout, 3, 3, Cin] bf16; bias [Cout] fp32 or null; Y [B, H, W, Cout] bf16; relu != 0: ReLU after the bias.
extern "C" int dtlr_conv3x3_patch_bf16(const void* X, const void* Wt, const float* bias, void* Y, int B, int H, int W, int Cin, int Cout,
                                       int relu, void* stream)
{
    clear_stale_error();
    if (!X || !Wt || !Y) return DTLR_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0) return DTLR_EINVAL;
    if (!dtlr_conv3x3_patch_supported(Cin, Cout)) return DTLR_ESHAPE;
    const int BN = (Cout % 128 == 0) ? 128 : 64;
    const long gz = (long)B * (Cout / BN);
    if (gz > 65535) return DTLR_ESHAPE;
    const dim3 grid((unsigned)((W + CP_TW - 1) / CP_TW), (unsigned)((H + CP_TH - 1) / CP_TH), (unsigned)gz);
    hipStream_t st = (hipStream_t)stream;
#define CP_LAUNCH(CB_, BN_, NW_)                                                                   \
    {                                                                                              \
        constexpr int lds_ = CB_ * CP_PLANE + CP_NS * BN_ * 128;                                   \
        static DevOnce once;                                                                       \
        if (once.first()) { (void)hipFuncSetAttribute((const void*)conv3x3_patch_kernel<CB_, BN_, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_); (void)hipGetLastError(); } \
        hipLaunchKernelGGL((conv3x3_patch_kernel<CB_, BN_, NW_>), grid, dim3(64 * NW_), lds_, st, (const uint16_t*)X, (const uint16_t*)Wt, bias, (uint16_t*)Y, H, W, Cout, relu); \
    }
    static const int nw8 = exp_env_int("DTLR_CONV_PATCH_NW8", 1);     // experiment builds: =0 four waves everywhere (A/B timing)
    if (BN == 128) {
        if (Cin == 64) CP_LAUNCH(1, 128, 4)
        else if (Cin == 128) { if (nw8) CP_LAUNCH(2, 128, 8) else CP_LAUNCH(2, 128, 4) }
        else { if (nw8) CP_LAUNCH(4, 128, 8) else CP_LAUNCH(4, 128, 4) }
    } else {
        if (Cin == 64) CP_LAUNCH(1, 64, 4) else if (Cin == 128) CP_LAUNCH(2, 64, 4) else CP_LAUNCH(4, 64, 4)
    }
#undef CP_LAUNCH
    return check_launch();
}

}  // namespace dtlr
