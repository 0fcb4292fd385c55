// ResNet stem: 7x7 / stride 2 / pad 3 convolution of the 3-channel input image, on the bf16 matrix cores,
// reading the NCHW fp32 image directly (torchvision resnet50.conv1 as run by models/dino/backbone.py:97-106; the
// FrozenBN scale is folded into the weights, its shift + ReLU are applied by the max-pool pass that follows).
//
// Why its own kernel: Cin = 3 does not fit the slab loader of the implicit-GEMM kernel (a K slab must be 128
// contiguous bytes of one tap), and the library path cost 0.32 ms per step (NCHW->NHWC/bf16 conversion passes +
// MIOpen igemm) for 0.1 GB of input.
//
// Formulation: out^T[64 ch, pixels] = Wk[64, K] . patches[K, pixels] with K ordered (ci, kh, kw') and kw' padded 7 -> 8:
// K = 21 (ci,kh) pairs x 8 = 168 -> 192 = 6 MFMA k-steps of 32 (pairs 21..23 and kw' = 7 carry zero weights).
//   * For a fixed (ci, kh) the 8 taps of an output pixel are 8 CONSECUTIVE input pixels of one NCHW row:
//     cols 2 ow - 3 .. 2 ow + 4.  The workgroup stages the rows it needs in LDS as bf16, shifted by 3 so that the
//     window of pixel ow starts at the EVEN element 2 ow.
//   * A lane's B-fragment (k-slots 8g..8g+7 = the 8 taps of pair 4 ks + g) is then 16 bytes of one LDS row.  Pixels are
//     interleaved over the four MFMA column tiles (tile t owns pixels 4 n + t): a lane reads the two ALIGNED 16-byte
//     pieces n, n+1 of its row once, and tile t's fragment is the dwords t..t+3 of that pair -- a compile-time register
//     window, no shifts, no unaligned LDS access.  2 ds_read_b128 feed 16 MFMAs.
//   * All 24 weight fragments (24 KB) live in registers (96 VGPRs) for the whole kernel.
// Workgroup = 4 output rows x 256 output columns of one image; wave = (row, 64-column strip) units.
#include "dtlr_common.h"

namespace dtlr {

typedef __attribute__((ext_vector_type(8))) h16_hw_t stem_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float stem_f32x4_t;

constexpr int STEM_ROWS = 4, STEM_COLS = 256;               // conv outputs per workgroup
constexpr int STEM_IN_ROWS = 2 * STEM_ROWS + 5;             // 13 input rows per channel
constexpr int STEM_IN_COLS = 2 * STEM_COLS + 8;             // 520 staged input columns
constexpr int STEM_PITCH = 640;                             // elements per LDS row: 1280 B = 5 x 256 B (rows bank-aligned)
constexpr int STEM_LDS = 3 * STEM_IN_ROWS * STEM_PITCH * 2; // 49920 B

__device__ __forceinline__ stem_f32x4_t stem_mma(const uint4& a, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, stem_f32x4_t c) {
    const uint4 b = make_uint4(b0, b1, b2, b3);
    return DTLR_MFMA_16x16x32_H16(__builtin_bit_cast(stem_bf16x8_t, a), __builtin_bit_cast(stem_bf16x8_t, b), c, 0, 0, 0);
}

// x [B,3,H,W] fp32 ; wfrag [4][6][64][8] bf16 (fragment-major, see dtlr_stem_pack_weights) ; y [B,Ho,Wo,64] bf16
__global__ __launch_bounds__(512, 2) void stem_conv7x7_kernel(const float* __restrict__ x, const uint16_t* __restrict__ wfrag,
                                                              uint16_t* __restrict__ y, int H, int W, int Ho, int Wo)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_stem[];
    uint16_t* img = reinterpret_cast<uint16_t*>(smem_stem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int ow_b0 = blockIdx.x * STEM_COLS, oh0 = blockIdx.y * STEM_ROWS, b = blockIdx.z;
    const int ir0 = 2 * oh0 - 3, ic0 = 2 * ow_b0 - 3;

    // ---- stage 3 x 13 input rows x 520 columns as 16-bit pairs (zero outside the image).  ALL of a thread's loads are issued before the
    // first conversion (clamped addresses, the zero padding applied by a select): as a load -> convert -> LDS-store loop the 20
    // iterations were 20 dependent HBM round trips per workgroup and the kernel ran at 1.6 TB/s of its output bytes.
    const float* xb = x + (long)b * 3 * H * W;
    constexpr int NP = 3 * STEM_IN_ROWS * (STEM_IN_COLS / 2);
    constexpr int NIT = (NP + 511) / 512;
    float v0[NIT], v1[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = min((int)threadIdx.x + 512 * it, NP - 1);
        const int row = p / (STEM_IN_COLS / 2), cc = (p % (STEM_IN_COLS / 2)) * 2;
        const int ci = row / STEM_IN_ROWS, ir = ir0 + row % STEM_IN_ROWS, ic = ic0 + cc;
        const float* src = xb + ((long)ci * H + min(max(ir, 0), H - 1)) * W;
        const float a0 = src[min(max(ic, 0), W - 1)], a1 = src[min(max(ic + 1, 0), W - 1)];
        const bool rok = ir >= 0 && ir < H;
        v0[it] = (rok && ic >= 0 && ic < W) ? a0 : 0.f;
        v1[it] = (rok && ic + 1 >= 0 && ic + 1 < W) ? a1 : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = (int)threadIdx.x + 512 * it;
        if (p < NP) {
            const int row = p / (STEM_IN_COLS / 2), cc = (p % (STEM_IN_COLS / 2)) * 2;
            *reinterpret_cast<uint32_t*>(img + row * STEM_PITCH + cc) = pack_bf16x2(v0[it], v1[it]);
        }
    }
    // ---- weights: 24 fragments, lane's 16 bytes each, held for the rest of the kernel (requested after the staging values have left
    // their registers -- together they would exceed the 128 VGPRs of two workgroups per CU -- and landing behind the barrier) ----
    __builtin_amdgcn_sched_barrier(0);
    uint4 wa[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) wa[i][ks] = *reinterpret_cast<const uint4*>(wfrag + ((i * 6 + ks) * 64 + lane) * 8);
    __syncthreads();

    // per-lane LDS row of pair 4 ks + g at output-row offset 0 (pairs >= 21 have zero weights: any finite row will do)
    int rowoff[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        const int pair = min(4 * ks + g, 20);
        rowoff[ks] = ((pair / 7) * STEM_IN_ROWS + pair % 7) * STEM_PITCH;
    }
    for (int u = wave; u < STEM_ROWS * (STEM_COLS / 64); u += 8) {
        const int ro = u >> 2, s = u & 3;
        const int oh = oh0 + ro;
        if (oh >= Ho) continue;                                 // wave-uniform
        stem_f32x4_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][t] = stem_f32x4_t{0.f, 0.f, 0.f, 0.f};
        const uint16_t* base = img + 2 * ro * STEM_PITCH + (16 * s + n) * 8;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const uint4 p0 = *reinterpret_cast<const uint4*>(base + rowoff[ks]);
            const uint4 p1 = *reinterpret_cast<const uint4*>(base + rowoff[ks] + 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][0] = stem_mma(wa[i][ks], p0.x, p0.y, p0.z, p0.w, acc[i][0]);
                acc[i][1] = stem_mma(wa[i][ks], p0.y, p0.z, p0.w, p1.x, acc[i][1]);
                acc[i][2] = stem_mma(wa[i][ks], p0.z, p0.w, p1.x, p1.y, acc[i][2]);
                acc[i][3] = stem_mma(wa[i][ks], p0.w, p1.x, p1.y, p1.z, acc[i][3]);
            }
        }
        // ---- store: lane (n, g) holds channels 16 i + 4 g + r of pixel 64 s + 4 n + t; channel tiles are paired with
        // v_permlane16_swap so that a lane writes 8 consecutive channels (16 bytes), 64 contiguous bytes per pixel
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ow = ow_b0 + 64 * s + 4 * n + t;
            uint16_t* dst = y + (((long)b * Ho + oh) * Wo + ow) * 64;
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const uint32_t lo0 = pack_bf16x2(acc[i][t][0], acc[i][t][1]), hi0 = pack_bf16x2(acc[i][t][2], acc[i][t][3]);
                const uint32_t lo1 = pack_bf16x2(acc[i + 1][t][0], acc[i + 1][t][1]), hi1 = pack_bf16x2(acc[i + 1][t][2], acc[i + 1][t][3]);
                const auto s0 = __builtin_amdgcn_permlane16_swap(lo0, lo1, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(hi0, hi1, false, false);
                // even g: channels 8 (g/2) .. of tile i ; odd g: the same 8 channels of tile i + 1
                if (ow < Wo)
                    *reinterpret_cast<uint4*>(dst + (i + (g & 1)) * 16 + 8 * (g >> 1)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
        }
    }
}


// ---- the same convolution for the SPLIT-fp32 engine (round 4): fp32 image, fp32 output, every product as three fp16 MFMAs on hi + lo halves ----
// The exact-fp32 stem (stem_conv7x7_f32_kernel: a direct convolution on the VALU) takes 0.88 ms of the split engine's 23.7 ms step.  This is
// stem_conv7x7_kernel's formulation with both operands split: the staged input rows exist twice in LDS (an fp16 hi plane and an fp16 lo plane
// of x - hi: same shifted layout, so tile t's B-fragment is the same compile-time dword window of two aligned 16-byte reads in either plane),
// and the 24 weight fragments exist twice (hi, lo) -- 48 KB, kept in LDS instead of registers (48 fragments would be 192 VGPRs):
//     acc += W_hi . x_lo + W_lo . x_hi + W_hi . x_hi          per (k-step, channel tile, pixel tile): 3 MFMAs, 288 per (row, 64-column strip) unit.
// LDS: 2 x 49,920 (planes) + 2 x 24,576 (weights) = 148,992 bytes: one workgroup per CU.  Output fp32 NHWC (the folded-BN shift, ReLU and
// max-pool follow in maxpool3x3s2, as for the exact engine).
typedef __attribute__((ext_vector_type(8))) _Float16 stem_f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 stem_f16x2_t;
constexpr int STEMS_PLANE = 3 * STEM_IN_ROWS * STEM_PITCH * 2;      // 49,920 B
constexpr int STEMS_WOFF = 2 * STEMS_PLANE;
constexpr int STEMS_LDS = STEMS_WOFF + 2 * 24576;
__device__ __forceinline__ stem_f32x4_t stems_mma(const uint4& a, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, stem_f32x4_t c) {
    const uint4 b = make_uint4(b0, b1, b2, b3);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(stem_f16x8_t, a), __builtin_bit_cast(stem_f16x8_t, b), c, 0, 0, 0);
}

// x [B,3,H,W] fp32 ; wfrag_hi / wfrag_lo [4][6][64][8] fp16 (dtlr_stem_pack_weights of the fp16 build applied to fp16(w) and to w - fp16(w)) ; y [B,Ho,Wo,64] fp32
__global__ __launch_bounds__(512, 1) void stem_conv7x7_f32s_kernel(const float* __restrict__ x, const uint16_t* __restrict__ wfrag_hi,
                                                                   const uint16_t* __restrict__ wfrag_lo, float* __restrict__ y,
                                                                   int H, int W, int Ho, int Wo)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_stem[];
    uint16_t* img_hi = reinterpret_cast<uint16_t*>(smem_stem);
    uint16_t* img_lo = reinterpret_cast<uint16_t*>(smem_stem + STEMS_PLANE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int ow_b0 = blockIdx.x * STEM_COLS, oh0 = blockIdx.y * STEM_ROWS, b = blockIdx.z;
    const int ir0 = 2 * oh0 - 3, ic0 = 2 * ow_b0 - 3;

    // ---- stage 3 x 13 input rows x 520 columns as fp16 hi / lo pairs (zero outside the image); all loads before the first conversion ----
    const float* xb = x + (long)b * 3 * H * W;
    constexpr int NP = 3 * STEM_IN_ROWS * (STEM_IN_COLS / 2);
    constexpr int NIT = (NP + 511) / 512;
    float v0[NIT], v1[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = min((int)threadIdx.x + 512 * it, NP - 1);
        const int row = p / (STEM_IN_COLS / 2), cc = (p % (STEM_IN_COLS / 2)) * 2;
        const int ci = row / STEM_IN_ROWS, ir = ir0 + row % STEM_IN_ROWS, ic = ic0 + cc;
        const float* src = xb + ((long)ci * H + min(max(ir, 0), H - 1)) * W;
        const float a0 = src[min(max(ic, 0), W - 1)], a1 = src[min(max(ic + 1, 0), W - 1)];
        const bool rok = ir >= 0 && ir < H;
        v0[it] = (rok && ic >= 0 && ic < W) ? a0 : 0.f;
        v1[it] = (rok && ic + 1 >= 0 && ic + 1 < W) ? a1 : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = (int)threadIdx.x + 512 * it;
        if (p < NP) {
            const int row = p / (STEM_IN_COLS / 2), cc = (p % (STEM_IN_COLS / 2)) * 2;
            const stem_f16x2_t hi = __builtin_convertvector(f32x2_hw_t{v0[it], v1[it]}, stem_f16x2_t);
            const stem_f16x2_t lo = __builtin_convertvector(f32x2_hw_t{v0[it] - (float)hi[0], v1[it] - (float)hi[1]}, stem_f16x2_t);
            *reinterpret_cast<uint32_t*>(img_hi + row * STEM_PITCH + cc) = __builtin_bit_cast(uint32_t, hi);
            *reinterpret_cast<uint32_t*>(img_lo + row * STEM_PITCH + cc) = __builtin_bit_cast(uint32_t, lo);
        }
    }
    // ---- weights: 2 x 24 fragments of 1 KB -> LDS (1536 x 16 bytes per image, 3 per thread) ----
    {
        uint4* wl = reinterpret_cast<uint4*>(smem_stem + STEMS_WOFF);
        const uint4* gh = reinterpret_cast<const uint4*>(wfrag_hi);
        const uint4* gl = reinterpret_cast<const uint4*>(wfrag_lo);
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int i = (int)threadIdx.x + 512 * it;
            wl[i] = gh[i];
            wl[1536 + i] = gl[i];
        }
    }
    __syncthreads();

    int rowoff[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        const int pair = min(4 * ks + g, 20);
        rowoff[ks] = ((pair / 7) * STEM_IN_ROWS + pair % 7) * STEM_PITCH;
    }
    const unsigned char* wfl = smem_stem + STEMS_WOFF + lane * 16;
    for (int u = wave; u < STEM_ROWS * (STEM_COLS / 64); u += 8) {
        const int ro = u >> 2, s = u & 3;
        const int oh = oh0 + ro;
        if (oh >= Ho) continue;                                 // wave-uniform
        stem_f32x4_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][t] = stem_f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int boff = 2 * ro * STEM_PITCH + (16 * s + n) * 8;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const uint4 h0 = *reinterpret_cast<const uint4*>(img_hi + boff + rowoff[ks]);
            const uint4 h1 = *reinterpret_cast<const uint4*>(img_hi + boff + rowoff[ks] + 8);
            const uint4 l0 = *reinterpret_cast<const uint4*>(img_lo + boff + rowoff[ks]);
            const uint4 l1 = *reinterpret_cast<const uint4*>(img_lo + boff + rowoff[ks] + 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 wh = *reinterpret_cast<const uint4*>(wfl + (i * 6 + ks) * 1024);
                const uint4 wl = *reinterpret_cast<const uint4*>(wfl + 24576 + (i * 6 + ks) * 1024);
                acc[i][0] = stems_mma(wh, l0.x, l0.y, l0.z, l0.w, acc[i][0]);
                acc[i][1] = stems_mma(wh, l0.y, l0.z, l0.w, l1.x, acc[i][1]);
                acc[i][2] = stems_mma(wh, l0.z, l0.w, l1.x, l1.y, acc[i][2]);
                acc[i][3] = stems_mma(wh, l0.w, l1.x, l1.y, l1.z, acc[i][3]);
                acc[i][0] = stems_mma(wl, h0.x, h0.y, h0.z, h0.w, acc[i][0]);
                acc[i][1] = stems_mma(wl, h0.y, h0.z, h0.w, h1.x, acc[i][1]);
                acc[i][2] = stems_mma(wl, h0.z, h0.w, h1.x, h1.y, acc[i][2]);
                acc[i][3] = stems_mma(wl, h0.w, h1.x, h1.y, h1.z, acc[i][3]);
                acc[i][0] = stems_mma(wh, h0.x, h0.y, h0.z, h0.w, acc[i][0]);
                acc[i][1] = stems_mma(wh, h0.y, h0.z, h0.w, h1.x, acc[i][1]);
                acc[i][2] = stems_mma(wh, h0.z, h0.w, h1.x, h1.y, acc[i][2]);
                acc[i][3] = stems_mma(wh, h0.w, h1.x, h1.y, h1.z, acc[i][3]);
            }
        }
        // ---- store: lane (n, g) holds channels 16 i + 4 g + r of pixel 64 s + 4 n + t: 16 bytes per (i, t) ----
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ow = ow_b0 + 64 * s + 4 * n + t;
            if (ow < Wo) {
                float* dst = y + (((long)b * Ho + oh) * Wo + ow) * 64 + 4 * g;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<float4*>(dst + 16 * i) = make_float4(acc[i][t][0], acc[i][t][1], acc[i][t][2], acc[i][t][3]);
            }
        }
    }
}


// ---- stem convolution + folded-BN shift + ReLU + 3x3 / stride-2 max-pool in ONE kernel (round 3) ------------------------------------------
// torchvision resnet50: conv1 -> bn1 -> relu -> maxpool (models/dino/backbone.py:97-106).  As two kernels the 64-channel full-resolution
// map (268 MB for 32 lines of 128 x 2048) is written by the convolution and read back by the pooling pass; here it never leaves the CU:
// a workgroup produces 2 pooled rows x 120 pooled columns from the 5 x 244 convolution outputs under them.
//   * the convolution is the kernel above (same staging, same MFMA formulation, same roundings: the conv output is rounded to the 16-bit
//     format exactly where the separate kernel stored it), as (conv row, 64-column strip) wave units; strips start every 60 conv columns
//     (a multiple of 4, so the window reads stay 16-byte aligned) and own 30 pooled columns each;
//   * horizontal max in registers: a lane holds 4 consecutive conv columns j = 4 n .. 4 n + 3 of its strip, pooled column k covers
//     j = 2 k .. 2 k + 2: k = 2 n is in-lane, k = 2 n + 1 takes j = 4 n + 4 from lane n + 1 (one DPP row shift per accumulator);
//     conv columns / rows outside the map count as -inf, like the pooling's padding;
//   * the row-pooled strips (5 x 120 x 64 channels, 86 KB with the bank padding) go to LDS; after one barrier every thread takes the vertical max of three
//     rows for its 8 channels, adds the folded-BN shift, applies the ReLU (max first, then + bias and ReLU: exact, see maxpool3x3s2_kernel)
//     and stores 16 bytes.
constexpr int SP_PROWS = 2, SP_STRIPS = 4, SP_SCOLS = 30;   // pooled rows per workgroup, strips, pooled columns per strip
constexpr int SP_PCOLS = SP_STRIPS * SP_SCOLS;              // 120 pooled columns per workgroup
constexpr int SP_CROWS = 2 * SP_PROWS + 1;                  // 5 conv rows
constexpr int SP_IN_ROWS = 2 * SP_CROWS + 5;                // 15 input rows per channel
constexpr int SP_IMG = 3 * SP_IN_ROWS * STEM_PITCH * 2;     // 57600 B staged input
constexpr int SP_CP = 144;                                  // bytes per pooled column in LDS: 128 + 16, so that the 8-byte stores of a wave
                                                            // (address 2 n SP_CP + 8 g) cover all banks: with 128 the 16 lanes of a group collide
constexpr int SP_HB = SP_CROWS * SP_PCOLS * SP_CP;          // 86400 B row-pooled conv outputs
constexpr int SP_LDS = SP_IMG + SP_HB;

template <int CTRL> __device__ __forceinline__ float sp_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

__global__ __launch_bounds__(512, 2) void stem_pool_kernel(const float* __restrict__ x, const uint16_t* __restrict__ wfrag,
                                                           const float* __restrict__ bias, uint16_t* __restrict__ y,
                                                           int H, int W, int Ho, int Wo, int Hp, int Wp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_stem[];
    uint16_t* img = reinterpret_cast<uint16_t*>(smem_stem);
    unsigned char* hb = smem_stem + SP_IMG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int pwb = blockIdx.x * SP_PCOLS, ph0 = blockIdx.y * SP_PROWS, b = blockIdx.z;
    const int cbW = 2 * pwb - 1, cr0 = 2 * ph0 - 1;             // first conv column / row of the workgroup (may be -1)
    const int ir0 = 2 * cr0 - 3, ic0 = 2 * cbW - 3;

    const float* xb = x + (long)b * 3 * H * W;
    constexpr int NP = 3 * SP_IN_ROWS * (STEM_IN_COLS / 2);
    constexpr int NIT = (NP + 511) / 512;
    float v0[NIT], v1[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = min((int)threadIdx.x + 512 * it, NP - 1);
        const int row = p / (STEM_IN_COLS / 2), cc = (p % (STEM_IN_COLS / 2)) * 2;
        const int ci = row / SP_IN_ROWS, ir = ir0 + row % SP_IN_ROWS, ic = ic0 + cc;
        const float* src = xb + ((long)ci * H + min(max(ir, 0), H - 1)) * W;
        const float a0 = src[min(max(ic, 0), W - 1)], a1 = src[min(max(ic + 1, 0), W - 1)];
        const bool rok = ir >= 0 && ir < H;
        v0[it] = (rok && ic >= 0 && ic < W) ? a0 : 0.f;
        v1[it] = (rok && ic + 1 >= 0 && ic + 1 < W) ? a1 : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = (int)threadIdx.x + 512 * it;
        if (p < NP) {
            const int row = p / (STEM_IN_COLS / 2), cc = (p % (STEM_IN_COLS / 2)) * 2;
            *reinterpret_cast<uint32_t*>(img + row * STEM_PITCH + cc) = pack_bf16x2(v0[it], v1[it]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    uint4 wa[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) wa[i][ks] = *reinterpret_cast<const uint4*>(wfrag + ((i * 6 + ks) * 64 + lane) * 8);
    __syncthreads();

    int rowoff[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        const int pair = min(4 * ks + g, 20);
        rowoff[ks] = ((pair / 7) * SP_IN_ROWS + pair % 7) * STEM_PITCH;
    }
    const uint32_t ninf2 = pack_bf16x2(-INFINITY, -INFINITY);
    for (int u = wave; u < SP_CROWS * SP_STRIPS; u += 8) {
        const int ro = u >> 2, s = u & 3;                         // conv row ro of the workgroup, strip s
        const int cr = cr0 + ro;
        unsigned char* hrow = hb + (ro * SP_PCOLS + SP_SCOLS * s) * SP_CP + (4 * g) * 2;
        if (cr < 0 || cr >= Ho) {                                 // wave-uniform: a padding row of the pooling
            if (n < 15) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    *reinterpret_cast<uint2*>(hrow + (2 * n) * SP_CP + i * 32) = make_uint2(ninf2, ninf2);
                    *reinterpret_cast<uint2*>(hrow + (2 * n + 1) * SP_CP + i * 32) = make_uint2(ninf2, ninf2);
                }
            }
            continue;
        }
        stem_f32x4_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][t] = stem_f32x4_t{0.f, 0.f, 0.f, 0.f};
        const uint16_t* base = img + 2 * ro * STEM_PITCH + (60 * s + 4 * n) * 2;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const uint4 p0 = *reinterpret_cast<const uint4*>(base + rowoff[ks]);
            const uint4 p1 = *reinterpret_cast<const uint4*>(base + rowoff[ks] + 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][0] = stem_mma(wa[i][ks], p0.x, p0.y, p0.z, p0.w, acc[i][0]);
                acc[i][1] = stem_mma(wa[i][ks], p0.y, p0.z, p0.w, p1.x, acc[i][1]);
                acc[i][2] = stem_mma(wa[i][ks], p0.z, p0.w, p1.x, p1.y, acc[i][2]);
                acc[i][3] = stem_mma(wa[i][ks], p0.w, p1.x, p1.y, p1.z, acc[i][3]);
            }
        }
        // conv columns of this lane: c = cbW + 60 s + 4 n + t; outside [0, Wo) they are the pooling's padding
        const int c0 = cbW + 60 * s + 4 * n;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float e[4][4];                                        // [t][r]: rounded to the 16-bit format like the stored conv output
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool ok = (unsigned)(c0 + t) < (unsigned)Wo;
                const uint32_t lo = pack_bf16x2(acc[i][t][0], acc[i][t][1]), hi = pack_bf16x2(acc[i][t][2], acc[i][t][3]);
                e[t][0] = ok ? h16_lo(lo) : -INFINITY; e[t][1] = ok ? h16_hi(lo) : -INFINITY;
                e[t][2] = ok ? h16_lo(hi) : -INFINITY; e[t][3] = ok ? h16_hi(hi) : -INFINITY;
            }
            float m0[4], m1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float nx = sp_dpp<0x101>(e[0][r]);          // row_shl:1 -- lane n receives lane n + 1's first column
                m0[r] = fmaxf(fmaxf(e[0][r], e[1][r]), e[2][r]);
                m1[r] = fmaxf(fmaxf(e[2][r], e[3][r]), nx);
            }
            if (n < 15) {
                *reinterpret_cast<uint2*>(hrow + (2 * n) * SP_CP + i * 32) = make_uint2(pack_bf16x2(m0[0], m0[1]), pack_bf16x2(m0[2], m0[3]));
                *reinterpret_cast<uint2*>(hrow + (2 * n + 1) * SP_CP + i * 32) = make_uint2(pack_bf16x2(m1[0], m1[1]), pack_bf16x2(m1[2], m1[3]));
            }
        }
    }
    __syncthreads();
    // ---- vertical max of three row-pooled rows, + folded-BN shift, ReLU, 16-byte stores ------------------------------------------------
    for (int idx = threadIdx.x; idx < SP_PROWS * SP_PCOLS * 8; idx += 512) {
        const int c8 = idx & 7, col = (idx >> 3) % SP_PCOLS, pr = idx / (8 * SP_PCOLS);
        const int ph = ph0 + pr, pw = pwb + col;
        if (ph >= Hp || pw >= Wp) continue;
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
#pragma unroll
        for (int dr = 0; dr < 3; ++dr) {
            const uint4 t = *reinterpret_cast<const uint4*>(hb + ((2 * pr + dr) * SP_PCOLS + col) * SP_CP + c8 * 16);
            const uint32_t w4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { m[2 * k] = fmaxf(m[2 * k], h16_lo(w4[k])); m[2 * k + 1] = fmaxf(m[2 * k + 1], h16_hi(w4[k])); }
        }
        const float4 b0 = *reinterpret_cast<const float4*>(bias + c8 * 8), b1 = *reinterpret_cast<const float4*>(bias + c8 * 8 + 4);
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k] + bb[k], 0.f);
        *reinterpret_cast<uint4*>(y + (((long)b * Hp + ph) * Wp + pw) * 64 + c8 * 8) =
            make_uint4(pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]), pack_bf16x2(m[4], m[5]), pack_bf16x2(m[6], m[7]));
    }
}


// ---- exact-fp32 stem (the parity engine).  Direct convolution on the vector ALUs: 2 * 147 * 64 flop per output pixel is
// 39 GFLOP for 32 lines of 128 x 2048 -- a millisecond at the fp32 vector rate; the fp32 MFMA (16x16x4) would need the taps
// as 4-wide k-steps of one (ci, kh) row, i.e. the same LDS staging for a kernel only the fp32 parity path runs.
// Workgroup = 8 x 32 output pixels of one image, one thread per pixel with all 64 channel accumulators in registers.
// LDS: the input patch [3][21][72] fp32 (the 7x7/s2 footprint of the tile, zero outside the image) and the weights k-major
// [147][64] fp32: the inner loop reads one patch value per lane and sixteen float4 weight broadcasts (same address in every
// lane: conflict-free), 64 FMAs per 17 LDS reads.  A lane stores its 64 channels as 16 x 16 B (the NHWC pixel is 256 B).
constexpr int SF_ROWS = 8, SF_COLS = 32;
constexpr int SF_IN_ROWS = 2 * SF_ROWS + 5;                 // 21
constexpr int SF_IN_COLS = 2 * SF_COLS + 5;                 // 69
constexpr int SF_PITCH = 72;
constexpr int SF_LDS = (3 * SF_IN_ROWS * SF_PITCH + 147 * 64) * 4;     // 55776 B

__global__ __launch_bounds__(256) void stem_conv7x7_f32_kernel(const float* __restrict__ x, const float* __restrict__ wk,
                                                               float* __restrict__ y, int H, int W, int Ho, int Wo)
{
    extern __shared__ __attribute__((aligned(16))) float sf_lds[];
    float* patch = sf_lds;                                   // [3][21][72]
    float* wl = sf_lds + 3 * SF_IN_ROWS * SF_PITCH;          // [147][64]
    const int b = blockIdx.z;
    const int oh0 = blockIdx.y * SF_ROWS, ow0 = blockIdx.x * SF_COLS;
    const int ih0 = 2 * oh0 - 3, iw0 = 2 * ow0 - 3;
    const float* xb = x + (long)b * 3 * H * W;
    for (int i = threadIdx.x; i < 3 * SF_IN_ROWS * SF_IN_COLS; i += 256) {
        const int c = i % SF_IN_COLS, r = (i / SF_IN_COLS) % SF_IN_ROWS, ci = i / (SF_IN_COLS * SF_IN_ROWS);
        const int ih = ih0 + r, iw = iw0 + c;
        float v = 0.f;
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = xb[((long)ci * H + ih) * W + iw];
        patch[(ci * SF_IN_ROWS + r) * SF_PITCH + c] = v;
    }
    for (int i = threadIdx.x; i < 147 * 16; i += 256)
        reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(wk)[i];
    __syncthreads();
    const int pr = threadIdx.x / SF_COLS, pc = threadIdx.x % SF_COLS;
    float acc[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) acc[c] = 0.f;
    for (int ci = 0; ci < 3; ++ci)
        for (int kh = 0; kh < 7; ++kh) {
            const float* prow = patch + (ci * SF_IN_ROWS + 2 * pr + kh) * SF_PITCH + 2 * pc;
            const float* wrow = wl + ((ci * 7 + kh) * 7) * 64;
#pragma unroll
            for (int kw = 0; kw < 7; ++kw) {
                const float xv = prow[kw];
#pragma unroll
                for (int c4 = 0; c4 < 16; ++c4) {
                    const float4 w4 = *reinterpret_cast<const float4*>(wrow + kw * 64 + c4 * 4);
                    acc[4 * c4 + 0] = fmaf(xv, w4.x, acc[4 * c4 + 0]);
                    acc[4 * c4 + 1] = fmaf(xv, w4.y, acc[4 * c4 + 1]);
                    acc[4 * c4 + 2] = fmaf(xv, w4.z, acc[4 * c4 + 2]);
                    acc[4 * c4 + 3] = fmaf(xv, w4.w, acc[4 * c4 + 3]);
                }
            }
        }
    const int oh = oh0 + pr, ow = ow0 + pc;
    if (oh < Ho && ow < Wo) {
        float4* o = reinterpret_cast<float4*>(y + (((long)b * Ho + oh) * Wo + ow) * 64);
#pragma unroll
        for (int c4 = 0; c4 < 16; ++c4) o[c4] = make_float4(acc[4 * c4], acc[4 * c4 + 1], acc[4 * c4 + 2], acc[4 * c4 + 3]);
    }
}

}  // namespace dtlr

using namespace dtlr;

// x [B,3,H,W] fp32 NCHW ; wk [147][64] fp32 (k = (ci*7 + kh)*7 + kw) ; y [B,Ho,Wo,64] fp32 NHWC
extern "C" int dtlr_stem_conv7x7_f32(const float* x, const float* wk, float* y, int B, int H, int W, void* stream)
{
    clear_stale_error();
    if (!x || !wk || !y) return DTLR_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0) return DTLR_EINVAL;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    static DevOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute((const void*)stem_conv7x7_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SF_LDS); (void)hipGetLastError(); }
    const dim3 grid((Wo + SF_COLS - 1) / SF_COLS, (Ho + SF_ROWS - 1) / SF_ROWS, B);
    if (grid.y > 65535u || grid.z > 65535u) return DTLR_ESHAPE;
    hipLaunchKernelGGL(stem_conv7x7_f32_kernel, grid, dim3(256), SF_LDS, (hipStream_t)stream, x, wk, y, H, W, Ho, Wo);
    return check_launch();
}

extern "C" int dtlr_stem_conv7x7_f32s(const float* x, const void* wfrag_hi, const void* wfrag_lo, float* y, int B, int H, int W, void* stream)
{
    clear_stale_error();
    if (!x || !wfrag_hi || !wfrag_lo || !y) return DTLR_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0) return DTLR_EINVAL;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    static DevOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute((const void*)stem_conv7x7_f32s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, STEMS_LDS); (void)hipGetLastError(); }
    const dim3 grid((Wo + STEM_COLS - 1) / STEM_COLS, (Ho + STEM_ROWS - 1) / STEM_ROWS, B);
    if (grid.y > 65535u || grid.z > 65535u) return DTLR_ESHAPE;
    hipLaunchKernelGGL(stem_conv7x7_f32s_kernel, grid, dim3(512), STEMS_LDS, (hipStream_t)stream,
                       x, (const uint16_t*)wfrag_hi, (const uint16_t*)wfrag_lo, y, H, W, Ho, Wo);
    return check_launch();
}

// conv1.weight (BN scale folded) [64, 3, 7, 7] fp32 (host or device memory readable by the host is NOT assumed: this packs on
// the host side from a host pointer) -> wfrag [4][6][64][8] bf16 as uint16 in host memory.
extern "C" int dtlr_stem_pack_weights(const float* w_oihw_host, unsigned short* wfrag_host)
{
    if (!w_oihw_host || !wfrag_host) return DTLR_EINVAL;
    for (int i = 0; i < 4; ++i)
        for (int ks = 0; ks < 6; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = lane & 15, g = lane >> 4, pair = 4 * ks + g;
                for (int e = 0; e < 8; ++e) {
                    float v = 0.f;
                    if (pair < 21 && e < 7) v = w_oihw_host[(((16 * i + m) * 3 + pair / 7) * 7 + pair % 7) * 7 + e];
                    wfrag_host[((i * 6 + ks) * 64 + lane) * 8 + e] = f32_to_h16_host(v);      // round to nearest even, NaN stays NaN
                }
            }
    return DTLR_OK;
}

extern "C" int dtlr_stem_conv7x7_pool(const float* x, const void* wfrag, const float* bias, void* y, int B, int H, int W, int out_dtype, void* stream)
{
    clear_stale_error();
    if (!x || !wfrag || !bias || !y) return DTLR_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0) return DTLR_EINVAL;
    if (out_dtype != DTLR_H16) return DTLR_EDTYPE;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    const int Hp = (Ho + 2 - 3) / 2 + 1, Wp = (Wo + 2 - 3) / 2 + 1;
    static DevOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute((const void*)stem_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS); (void)hipGetLastError(); }
    const dim3 grid((Wp + SP_PCOLS - 1) / SP_PCOLS, (Hp + SP_PROWS - 1) / SP_PROWS, B);
    if (grid.y > 65535u || grid.z > 65535u) return DTLR_ESHAPE;
    hipLaunchKernelGGL(stem_pool_kernel, grid, dim3(512), SP_LDS, (hipStream_t)stream,
                       x, (const uint16_t*)wfrag, bias, (uint16_t*)y, H, W, Ho, Wo, Hp, Wp);
    return check_launch();
}

extern "C" int dtlr_stem_conv7x7(const float* x, const void* wfrag, void* y, int B, int H, int W, int out_dtype, void* stream)
{
    clear_stale_error();
    if (!x || !wfrag || !y) return DTLR_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0) return DTLR_EINVAL;
    if (out_dtype != DTLR_H16) return DTLR_EDTYPE;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    static DevOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute((const void*)stem_conv7x7_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, STEM_LDS); (void)hipGetLastError(); }
    const dim3 grid((Wo + STEM_COLS - 1) / STEM_COLS, (Ho + STEM_ROWS - 1) / STEM_ROWS, B);
    hipLaunchKernelGGL(stem_conv7x7_kernel, grid, dim3(512), STEM_LDS, (hipStream_t)stream,
                       x, (const uint16_t*)wfrag, (uint16_t*)y, H, W, Ho, Wo);
    return check_launch();
}
