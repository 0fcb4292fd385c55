// Projection / FFN / 1x1-conv GEMM for gfx950:  C[M,N] = epi( (A [+ A2])[M,K] . W[N,K]^T )
//
// Every dense contraction of the transformer and every 1x1 convolution of the NHWC backbone has this
// "NT" form (activations token-major [M,K], nn.Linear weights [N,K]); SURVEY.md appendix C lists the
// shapes (M = B*S = 174,080 tokens or B*900 queries; K, N in 64..2048).
//
// Design (MI355X_MICROARCH / cdna_hip_programming section 5):
//   * 128x128 output tile, 256 threads = 4 wavefronts as 2 (tokens) x 2 (channels); each wave owns a
//     64x64 sub-tile = 4x4 MFMA 16x16 accumulators (64 fp32 regs/lane).
//   * MFMA operands are swapped on purpose:  acc = mfma(A-operand = WEIGHT rows, B-operand = ACTIVATION
//     rows)  =>  the 16x16 accumulator holds C^T: lane (g = l>>4, n = l&15) owns 4 CONSECUTIVE
//     channels (4g..4g+3) of token n.  The epilogue therefore reads bias/residual and writes C in
//     8/16-byte vectors, and a per-token reduction is in-lane + two xor-shuffles.
//   * K is walked in 128-byte slabs (64 bf16 / 32 fp32) staged global -> registers -> LDS (16-byte
//     loads, 8 lanes per 128-byte row: full-line coalescing), double-buffered with ONE barrier per
//     slab; LDS rows are padded to 144 B (9 x 16 B, 9 coprime with 16) so the 16 rows of a fragment
//     land on 16 distinct 16-byte slots for ds_read_b128.
//   * bf16: v_mfma_f32_16x16x32_bf16 (one per 64-byte k-slab);  fp32: v_mfma_f32_16x16x4_f32 (exact
//     fp32, four per 64-byte slab; the lane's 4 consecutive k of a 16-byte read feed MFMA j = 0..3 --
//     a k-permutation applied identically to both operands, so the sum is unchanged).
//   * persistent tile chains: a workgroup walks a contiguous run of output tiles (channel tiles fastest, so
//     a chain re-reads its activation rows from its own XCD's L2) as one continuously pipelined slab
//     stream -- the load-latency prologue is paid once per workgroup, not once per 128x128 tile.
// Fused prologue: A + A2 (query = src + pos, deformable_transformer.py:797-812).
// Fused epilogue: + bias, ReLU, zero masked rows (value.masked_fill, ms_deform_attn.py:95-96),
//                 + residual, ReLU-after-residual (ResNet bottleneck tail), output fp32 or bf16.
#include "dtlr_common.h"

namespace dtlr {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int BM = 128, BN = 128, SLAB = 128, LDS_ROW = 144;      // bytes
constexpr int TILE_BYTES = BM * LDS_ROW;                           // one operand tile in LDS

enum : int { EPI_BIAS = 1, EPI_RELU = 2, EPI_RESIDUAL = 4, EPI_ROWMASK = 8, EPI_RELU_POST = 16 };

template <typename T> struct GT;
template <> struct GT<uint16_t> {   // bf16
    static constexpr int BK = 64;
    static __device__ __forceinline__ uint4 add(uint4 a, uint4 b) {
        const uint32_t x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o[i] = pack_bf16x2(__uint_as_float(x[i] << 16) + __uint_as_float(y[i] << 16),
                               __uint_as_float(x[i] & 0xffff0000u) + __uint_as_float(y[i] & 0xffff0000u));
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
    static __device__ __forceinline__ void mma(const uint4& w, const uint4& x, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), acc, 0, 0, 0);
    }
};
template <> struct GT<float> {
    static constexpr int BK = 32;
    static __device__ __forceinline__ uint4 add(uint4 a, uint4 b) {
        return make_uint4(__float_as_uint(__uint_as_float(a.x) + __uint_as_float(b.x)), __float_as_uint(__uint_as_float(a.y) + __uint_as_float(b.y)),
                          __float_as_uint(__uint_as_float(a.z) + __uint_as_float(b.z)), __float_as_uint(__uint_as_float(a.w) + __uint_as_float(b.w)));
    }
    static __device__ __forceinline__ void mma(const uint4& w, const uint4& x, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(x.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(x.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(x.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(x.w), acc, 0, 0, 0);
    }
};

template <typename OutT> struct Out;
template <> struct Out<float> {
    static __device__ __forceinline__ void ld4(const float* p, float (&v)[4]) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void st4(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Out<uint16_t> {
    static __device__ __forceinline__ void ld4(const uint16_t* p, float (&v)[4]) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u); v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u); }
    static __device__ __forceinline__ void st4(uint16_t* p, const float (&v)[4]) { *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])); }
    static __device__ __forceinline__ float ld(const uint16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(uint16_t* p, float v) { *p = f32_to_bf16(v); }
};

// Implicit-GEMM convolution (CONV = true): the "A" operand is gathered on the fly from an NHWC image,
// M = B*Ho*Wo output pixels, K = KH*KW*Cin ordered (kh, kw, ci) so that a 128-byte K slab lies inside
// one filter tap (Cin*sizeof(T) % 128 == 0) and is one contiguous, 16-byte-aligned run of channels;
// taps that fall into the zero padding contribute zeros.  Weights are packed [Cout][KH][KW][Cin].
struct ConvP { int H, W, Cin, Ho, Wo, KH, KW, stride, pad; };

template <typename T, typename OutT, bool HAS_A2, bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(
    const T* __restrict__ A, const T* __restrict__ A2, const T* __restrict__ W,
    const float* __restrict__ bias, const OutT* __restrict__ residual, const uint8_t* __restrict__ row_mask,
    OutT* __restrict__ C, int M, int N, int K, int flags, int nN, int ntiles, int tiles_per_block, ConvP cp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 stages][W tile | X tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int BK = GT<T>::BK;
    const int nk = K / BK;

    // PERSISTENT tile chain: this block owns tiles [t_begin, t_end) of the (tm, tn) grid, tn fastest, and
    // walks them as ONE flat stream of K slabs, software-pipelined across tile boundaries: the global
    // loads of the next tile's first slab are in flight while the current tile's last slab is being
    // multiplied.  With K = 256 a tile is only 4 slabs, so paying the ~2 us load latency once per
    // block instead of once per tile is worth more than any in-tile tuning; consecutive tiles of a
    // chain share their activation rows (tn fastest), re-read from L2.
    const int t_begin = blockIdx.x * tiles_per_block;
    const int t_end = min(t_begin + tiles_per_block, ntiles);
    if (t_begin >= t_end) return;
    const int total = (t_end - t_begin) * nk;

    // staging: each operand tile = 128 rows x 128 B = 1024 16-byte chunks; thread t takes rows srow + 32 i
    // (8 consecutive lanes cover one 128-byte row slab), kc = tid & 7.
    const int srow = tid >> 3, kc = tid & 7;
    const int lds0 = srow * LDS_ROW + kc * 16;
    long a_off[4], w_off[4];                                    // loader state: byte offsets at k-slab 0
    int hi0[4], wi0[4];                                         // CONV: top-left input coordinate of the pixel
    const int slabs_per_tap = CONV ? (cp.Cin * (int)sizeof(T)) / SLAB : 1;
    const char* Ab = reinterpret_cast<const char*>(A);
    const char* A2b = reinterpret_cast<const char*>(A2);
    const char* Wb = reinterpret_cast<const char*>(W);

#define SET_LOAD_TILE(TILE)                                                                        \
    {                                                                                              \
        const int lm0_ = ((TILE) / nN) * BM, ln0_ = ((TILE) % nN) * BN;                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
            const long ar = min(lm0_ + srow + 32 * i, M - 1), wr = min(ln0_ + srow + 32 * i, N - 1); \
            w_off[i] = (wr * K) * (long)sizeof(T) + kc * 16;                                       \
            if (CONV) {                                                                            \
                const int hw = cp.Ho * cp.Wo;                                                      \
                const int bimg = (int)(ar / hw), rem = (int)(ar % hw);                             \
                hi0[i] = (rem / cp.Wo) * cp.stride - cp.pad;                                       \
                wi0[i] = (rem % cp.Wo) * cp.stride - cp.pad;                                       \
                a_off[i] = (long)bimg * cp.H * cp.W * cp.Cin * (long)sizeof(T) + kc * 16;          \
            } else {                                                                               \
                hi0[i] = wi0[i] = 0;                                                               \
                a_off[i] = (ar * K) * (long)sizeof(T) + kc * 16;                                   \
            }                                                                                      \
        }                                                                                          \
    }

    // Staging registers are named scalars on purpose: as arrays written under a condition hipcc
    // (ROCm 7.2) leaves them in scratch memory (global_load -> scratch_store ... scratch_load -> ds_write).
    uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
#define GLOAD1(I, OFF)                                                                             \
    rw##I = *reinterpret_cast<const uint4*>(Wb + w_off[I] + (OFF));                                \
    if (CONV) {                                                                                    \
        const int hi_ = hi0[I] + kh_, wi_ = wi0[I] + kw_;                                          \
        const bool ok_ = hi_ >= 0 && hi_ < cp.H && wi_ >= 0 && wi_ < cp.W;                         \
        const long po_ = ((long)(ok_ ? hi_ : 0) * cp.W + (ok_ ? wi_ : 0)) * cp.Cin * (long)sizeof(T) + coff_; \
        const uint4 t_ = *reinterpret_cast<const uint4*>(Ab + a_off[I] + po_);                     \
        ra##I = ok_ ? t_ : make_uint4(0u, 0u, 0u, 0u);                                             \
    } else {                                                                                       \
        ra##I = *reinterpret_cast<const uint4*>(Ab + a_off[I] + (OFF));                            \
        if (HAS_A2) ra##I = GT<T>::add(ra##I, *reinterpret_cast<const uint4*>(A2b + a_off[I] + (OFF))); \
    }
#define GLOAD(KT)                                                                                  \
    {                                                                                              \
        const long off_ = (long)(KT) * SLAB;                                                       \
        const int tap_ = (KT) / slabs_per_tap;                                                     \
        const int kh_ = CONV ? tap_ / cp.KW : 0, kw_ = CONV ? tap_ % cp.KW : 0;                    \
        const long coff_ = (long)((KT) % slabs_per_tap) * SLAB;                                    \
        (void)kh_; (void)kw_; (void)coff_;                                                         \
        GLOAD1(0, off_) GLOAD1(1, off_) GLOAD1(2, off_) GLOAD1(3, off_)                            \
    }
#define LSTORE1(I)                                                                                 \
    *reinterpret_cast<uint4*>(wt_ + I * 32 * LDS_ROW) = rw##I;                                     \
    *reinterpret_cast<uint4*>(wt_ + TILE_BYTES + I * 32 * LDS_ROW) = ra##I;
#define LSTORE(STAGE)                                                                              \
    {                                                                                              \
        unsigned char* wt_ = smem + (STAGE) * 2 * TILE_BYTES + lds0;                               \
        LSTORE1(0) LSTORE1(1) LSTORE1(2) LSTORE1(3)                                                \
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) acc[ci][ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    SET_LOAD_TILE(t_begin)
    GLOAD(0)
    LSTORE(0)
    __syncthreads();
    int kt = 0, tile = t_begin;              // slab being multiplied
    int lkt = 0, ltile = t_begin;            // slab whose loads were issued last
    for (int s = 0; s < total; ++s) {
        const int cur = s & 1;
        const bool more = s + 1 < total;
        if (more) {
            if (++lkt == nk) { lkt = 0; ++ltile; SET_LOAD_TILE(ltile) }
            GLOAD(lkt)
        }
        const unsigned char* wt = smem + cur * 2 * TILE_BYTES + (wn * 64 + n) * LDS_ROW + g * 16;
        const unsigned char* xt = smem + cur * 2 * TILE_BYTES + TILE_BYTES + (wm * 64 + n) * LDS_ROW + g * 16;
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
            uint4 wf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wf[i] = *reinterpret_cast<const uint4*>(wt + i * 16 * LDS_ROW + kq * 64);
                xf[i] = *reinterpret_cast<const uint4*>(xt + i * 16 * LDS_ROW + kq * 64);
            }
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) GT<T>::mma(wf[ci], xf[ti], acc[ci][ti]);
        }
        if (more) LSTORE(cur ^ 1)
        __syncthreads();
        if (++kt < nk) continue;

        // ---- tile finished: epilogue (the next tile's first slab is already in LDS) ----------------------
        // lane (g,n) holds channels ch = n0 + wn*64 + ci*16 + 4g + r of token m0 + wm*64 + ti*16 + n
        const int m0 = (tile / nN) * BM, n0 = (tile % nN) * BN;
        const bool vec_ok = (N & 3) == 0;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            const int tok = m0 + wm * 64 + ti * 16 + n;
            const bool tok_ok = tok < M;
            const bool masked = tok_ok && (flags & EPI_ROWMASK) && row_mask[tok];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                const int ch = n0 + wn * 64 + ci * 16 + 4 * g;
                float v[4] = {acc[ci][ti][0], acc[ci][ti][1], acc[ci][ti][2], acc[ci][ti][3]};
                acc[ci][ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (!tok_ok || ch >= N) continue;
                const bool full = vec_ok && ch + 3 < N;
                if (flags & EPI_BIAS) {
                    if (full) { const float4 bb = *reinterpret_cast<const float4*>(bias + ch); v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w; }
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (ch + r < N) v[r] += bias[ch + r];
                    }
                }
                if (flags & EPI_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                if (masked) { v[0] = v[1] = v[2] = v[3] = 0.f; }
                OutT* cptr = C + (long)tok * N + ch;
                if (full) {
                    if (flags & EPI_RESIDUAL) { float rr[4]; Out<OutT>::ld4(residual + (long)tok * N + ch, rr); v[0] += rr[0]; v[1] += rr[1]; v[2] += rr[2]; v[3] += rr[3]; }
                    if (flags & EPI_RELU_POST) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    Out<OutT>::st4(cptr, v);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ch + r < N) {
                            float x = v[r];
                            if (flags & EPI_RESIDUAL) x += Out<OutT>::ld(residual + (long)tok * N + ch + r);
                            if (flags & EPI_RELU_POST) x = fmaxf(x, 0.f);
                            Out<OutT>::st(cptr + r, x);
                        }
                }
            }
        }
        kt = 0; ++tile;
    }
#undef GLOAD
#undef LSTORE
#undef GLOAD1
#undef LSTORE1
#undef SET_LOAD_TILE
}

// tiles per block: enough chains to fill the chip (2 resident workgroups x 256 CUs) a few times over
static inline int plan_chain(long ntiles) {
    const long target_blocks = 2 * 256 * 2;
    long per = (ntiles + target_blocks - 1) / target_blocks;
    if (per < 1) per = 1;
    if (per > 64) per = 64;
    return (int)per;
}

template <typename T, typename OutT>
static int launch_conv(const void* X, const void* W, const float* bias, const void* residual, void* C,
                       int M, int N, int K, int flags, const ConvP& cp, hipStream_t st)
{
    const int nM = (M + BM - 1) / BM, nN = (N + BN - 1) / BN;
    const long nwg = (long)nM * nN;
    if (nwg > 0x7fffffffL) return DTLR_ESHAPE;
    const size_t lds = 4 * TILE_BYTES;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<T, OutT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    const int per = plan_chain(nwg);
    const unsigned grid = (unsigned)((nwg + per - 1) / per);
    hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, false, true>), dim3(grid), dim3(256), lds, st,
                       (const T*)X, (const T*)nullptr, (const T*)W, bias, (const OutT*)residual, (const uint8_t*)nullptr, (OutT*)C,
                       M, N, K, flags, nN, (int)nwg, per, cp);
    return check_launch();
}

template <typename T, typename OutT>
static int launch_gemm(const void* A, const void* A2, const void* W, const float* bias, const void* residual,
                       const uint8_t* row_mask, void* C, int M, int N, int K, int flags, hipStream_t st)
{
    const ConvP cp{};
    const int per = plan_chain((long)((M + BM - 1) / BM) * ((N + BN - 1) / BN));
    const unsigned grid = (unsigned)(((long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) + per - 1) / per);
    const int nM = (M + BM - 1) / BM, nN = (N + BN - 1) / BN;
    const long nwg = (long)nM * nN;
    if (nwg > 0x7fffffffL) return DTLR_ESHAPE;
    const size_t lds = 4 * TILE_BYTES;
    if (A2) {
        static bool attr_a2 = false;
        if (!attr_a2) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<T, OutT, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_a2 = true; }
        hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, true, false>), dim3(grid), dim3(256), lds, st,
                           (const T*)A, (const T*)A2, (const T*)W, bias, (const OutT*)residual, row_mask, (OutT*)C, M, N, K, flags, nN, (int)nwg, per, cp);
    } else {
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<T, OutT, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
        hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, false, false>), dim3(grid), dim3(256), lds, st,
                           (const T*)A, (const T*)nullptr, (const T*)W, bias, (const OutT*)residual, row_mask, (OutT*)C, M, N, K, flags, nN, (int)nwg, per, cp);
    }
    return check_launch();
}

}  // namespace dtlr

using namespace dtlr;

extern "C" int dtlr_gemm_nt(const void* A, const void* A2, const void* W, const float* bias,
                            const void* residual, const unsigned char* row_mask, void* C,
                            int M, int N, int K, int relu, int in_dtype, int out_dtype, void* stream)
{
    if (!A || !W || !C) return DTLR_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0) return DTLR_EINVAL;
    int flags = (bias ? EPI_BIAS : 0) | (relu == 1 ? EPI_RELU : 0) | (relu == 2 ? EPI_RELU_POST : 0) |
                (residual ? EPI_RESIDUAL : 0) | (row_mask ? EPI_ROWMASK : 0);
    hipStream_t st = (hipStream_t)stream;
    if (in_dtype == DTLR_BF16) {
        if (K % 64) return DTLR_ESHAPE;
        if (out_dtype == DTLR_BF16) return launch_gemm<uint16_t, uint16_t>(A, A2, W, bias, residual, row_mask, C, M, N, K, flags, st);
        if (out_dtype == DTLR_F32) return launch_gemm<uint16_t, float>(A, A2, W, bias, residual, row_mask, C, M, N, K, flags, st);
        return DTLR_EDTYPE;
    }
    if (in_dtype == DTLR_F32) {
        if (K % 32) return DTLR_ESHAPE;
        if (out_dtype == DTLR_F32) return launch_gemm<float, float>(A, A2, W, bias, residual, row_mask, C, M, N, K, flags, st);
        return DTLR_EDTYPE;
    }
    return DTLR_EDTYPE;
}

extern "C" int dtlr_conv2d_nhwc(const void* X, const void* W, const float* bias, const void* residual, void* Y,
                                int B, int H, int Wd, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                int relu, int dtype, void* stream)
{
    if (!X || !W || !Y) return DTLR_EINVAL;
    if (B <= 0 || H <= 0 || Wd <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return DTLR_EINVAL;
    const int elem = dtype == DTLR_BF16 ? 2 : dtype == DTLR_F32 ? 4 : 0;
    if (!elem) return DTLR_EDTYPE;
    if ((Cin * elem) % SLAB) return DTLR_ESHAPE;                 // a K slab must stay inside one tap
    ConvP cp;
    cp.H = H; cp.W = Wd; cp.Cin = Cin; cp.KH = KH; cp.KW = KW; cp.stride = stride; cp.pad = pad;
    cp.Ho = (H + 2 * pad - KH) / stride + 1;
    cp.Wo = (Wd + 2 * pad - KW) / stride + 1;
    if (cp.Ho <= 0 || cp.Wo <= 0) return DTLR_ESHAPE;
    const long M = (long)B * cp.Ho * cp.Wo;
    if (M > 0x7fffffffL) return DTLR_ESHAPE;
    const int K = KH * KW * Cin;
    int flags = (bias ? EPI_BIAS : 0) | (relu == 1 ? EPI_RELU : 0) | (relu == 2 ? EPI_RELU_POST : 0) | (residual ? EPI_RESIDUAL : 0);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DTLR_BF16) return launch_conv<uint16_t, uint16_t>(X, W, bias, residual, Y, (int)M, Cout, K, flags, cp, st);
    return launch_conv<float, float>(X, W, bias, residual, Y, (int)M, Cout, K, flags, cp, st);
}
