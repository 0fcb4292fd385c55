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
//     slab; LDS rows are 128 B with their 16-byte chunks XOR-swizzled by (row & 7) so the 16 lanes of
//     every ds_read_b128 service group land on 16 distinct 16-byte slots.
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
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace dtlr {

typedef __attribute__((ext_vector_type(8))) h16_hw_t bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int BM = 128, BN = 128, SLAB = 128, LDS_ROW = 128;      // bytes (rows unpadded; XOR-swizzled 16-byte chunks)
constexpr int TILE_BYTES = BM * LDS_ROW;                           // one operand tile in LDS

// Staging loads are issued through inline asm so that hipcc does not count them: across the loop
// back-edge its waitcnt pass is conservative and drains vmcnt to 0 at the first LDS store of the older
// register set, which collapses the two-slab prefetch distance to one.  The kernel places the counted
// wait itself (cdna_hip_programming.md section 5.7, form iii): one `s_waitcnt vmcnt(N)` + sched_barrier
// before the LDS stores of a set, N = number of staging loads issued after that set's loads.  vmcnt
// also counts the epilogue's stores/loads issued in between; that only makes the wait stricter.
__device__ __forceinline__ uint4 asm_load16(const char* p) {
    uint4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Ablation switches for timing experiments (tools/profile_kernels.py --only gemm_ablate): only honoured when
// the library is built with -DDTLR_GEMM_ABLATION, so the production hot loop carries no extra branches.
#ifdef DTLR_GEMM_ABLATION
#define ABLATE(BIT) (flags & (BIT))
#else
#define ABLATE(BIT) false
#endif
#ifdef DTLR_GEMM_TRACE
// Timeline instrumentation (profiling builds only): lane 0 of wave 0 (MFMA role) and of wave 4 (loader role) of the first
// 8 logical workgroups append (cycle counter << 8 | event code) records; read back with dtlr_debug_gemm_trace.
//   loader: 1 wait begin, 2 data landed (vmcnt), 3 LDS stores done, 4 next loads issued, 5 barrier passed
//   MFMA  : 10 barrier passed (slab start), 11 slab's MFMAs issued, 12 epilogue done
#define TR(...) __VA_ARGS__
#define TL_BLOCKS 8
#define TL_EVENTS 1024
__device__ unsigned long long g_gemm_tl[TL_BLOCKS * 2 * TL_EVENTS];
#define TL_INIT(ROLE)                                                                              \
    unsigned long long* tl_ = nullptr; int tl_n_ = 0;                                              \
    if ((threadIdx.x & 63) == 0 && logical0_ < TL_BLOCKS) tl_ = g_gemm_tl + ((long)logical0_ * 2 + (ROLE)) * TL_EVENTS;
#define TL_EV(CODE) { if (tl_ && tl_n_ < TL_EVENTS) tl_[tl_n_++] = ((unsigned long long)__builtin_readcyclecounter() << 8) | (CODE); }
#define TL_PARAMS , unsigned long long* tl_, int& tl_n_
#define TL_ARGS , tl_, tl_n_
#define TL_ARGS_NONE , tl_none_, tl_n_none_
#else
#define TL_PARAMS
#define TL_ARGS
#define TL_ARGS_NONE
#define TR(...)
#define TL_INIT(ROLE)
#define TL_EV(CODE)
#endif

enum : int { EPI_BIAS = 1, EPI_RELU = 2, EPI_RESIDUAL = 4, EPI_ROWMASK = 8, EPI_RELU_POST = 16,
              EPI_GELU = 64,        // exact GELU (erf) after the bias: Swin's Mlp (swin_transformer.py:18-36, nn.GELU)
              EPI_ROWMAX = 32,      // C is NOT written: C[token] (fp32, M entries, pre-filled with -inf) <- max over channels of acc + bias
              DBG_NO_LOAD = 256, DBG_NO_MMA = 512, DBG_NO_LDS = 1024, DBG_NO_EPI = 2048, DBG_NO_STORE = 4096 };   // ablation switches (env DTLR_GEMM_ABLATE), timing only

template <typename T> struct GT;
template <> struct GT<uint16_t> {   // bf16
    static constexpr int BK = 64;
    static __device__ __forceinline__ uint4 add(uint4 a, uint4 b) {
        const uint32_t x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o[i] = pack_bf16x2(h16_lo(x[i]) + h16_lo(y[i]),
                               h16_hi(x[i]) + h16_hi(y[i]));
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
    static __device__ __forceinline__ void mma(const uint4& w, const uint4& x, f32x4_t& acc) {
        acc = DTLR_MFMA_16x16x32_H16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), acc, 0, 0, 0);
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

// ---- split-fp16 operands ("f32s", round 4): fp32-grade products at the 16-bit matrix rate / 3 ----------------------------------
// The parity engine's GEMMs ran on v_mfma_f32_16x16x4_f32 (exact fp32, 1/16 of the 16-bit rate: 567 lines/s against 3556).  Here an
// fp32 operand x is carried as TWO fp16 numbers, hi = fp16(x) and lo = fp16(x - hi) (x - hi is exact in fp32; |lo| <= 2^-11 |hi|).
// Precision of the pair: 22 significand bits while lo is a NORMAL fp16 number, i.e. for |x| >= 2^-3; below that lo is an fp16 subnormal
// (gfx950's f16 MFMA keeps subnormal inputs -- measured) with the ABSOLUTE spacing 2^-24, so the pair carries x to 2^-25 absolute:
// 2^-21 relative at |x| = 1/16 (the synthetic weights), ~2^-18 at |x| = 1e-2 (a trained checkpoint's typical weight), nothing below
// 2^-25.  Activations are O(1) post-normalisation values (full 22 bits); the weights are where the floor shows.  And
//     a . w  ~=  a_hi w_hi + a_lo w_hi + a_hi w_lo            (the dropped a_lo w_lo term is 2^-22 relative)
// is three v_mfma_f32_16x16x32_f16 with fp32 accumulation: 48 matrix cycles per 32 k against 256 for the exact-fp32 MFMAs.
//   * ACTIVATIONS stay fp32 in HBM (the residual streams, LayerNorm / GroupNorm inputs and the reference-point chain are never
//     rounded); the LOADER waves split them while staging: a lane's 16-byte load is 4 consecutive k of one row -> 8 bytes of hi and
//     8 bytes of lo.  The LDS row keeps its 128 bytes: 16-byte chunk c < 4 = hi of k 8c..8c+7, chunk 4 + c = lo of the same k
//     (chunks XOR-swizzled by row & 7 like every other tile of this file), so an MFMA lane's two ds_read_b128 per row -- chunk g and
//     chunk 4 + g, the addresses the 16-bit kernel reads for its two k-halves -- are exactly its hi and lo B-fragments.
//   * WEIGHTS are split ONCE (dtlr_split_pack_weights) into the same slab image, an array with the size and shape of the fp32
//     weight: the loader copies it to LDS unchanged.
// Error of one term: <= 2^-21 |a w| + 2^-24 (|a| + |w|) (representation + dropped term + the subnormal floor of the lo halves;
// restated and swept in tests/test_host_logic.py) under fp32 accumulation; measured against fp64 in tests/test_gpu_kernels.py.
// (Pre-scaling lo by 2^11 would lift the floor but needs a second accumulator set for the correction products: 64 more VGPRs in the
// MFMA waves of a kernel that sits at its 128-register bound.)  Range: |x| < 65504
// (fp16 hi); every activation of this network is a post-normalisation value, a ReLU of one or a frozen-BN'd convolution output.
struct f32s_t { float v; };
template <typename T> constexpr bool kSplit = std::is_same<T, f32s_t>::value;
// XOR swizzle of the 16-byte chunks of LDS row r.  16-bit / fp32 operands: r & 7 (every ds_read_b128 service group lands on 16 distinct
// slots; their loaders write whole 128-byte rows).  Split operands: the activation loader writes a row's hi and lo halves as 8-byte stores
// (ds_write_b64: 16 CONTIGUOUS lanes per service group = rows r, r + 1, banks taken modulo 128 bytes), and with r & 7 both rows put their
// four hi chunks on the same 64 bytes -- a 2-way conflict on every store, the constant ~20% conflict share of the split GEMM's LDS cycles
// (profiles/r04_sq_f32s_v1.txt).  Toggling chunk bit 2 with the row's parity sends the odd row's hi chunks to the other 64 bytes: stores
// conflict-free, reads still 16 distinct slots per group (tools/lds_bank_model.py enumerates both).
template <typename T> __device__ __forceinline__ int lds_swz(int r) {
    if constexpr (kSplit<T>) return (r & 7) ^ ((r & 1) << 2);
    else return r & 7;
}
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
template <> struct GT<f32s_t> {
    static constexpr int BK = 32;
    static __device__ __forceinline__ uint4 add(uint4 a, uint4 b) { return GT<float>::add(a, b); }
    static __device__ __forceinline__ void mma(const uint4&, const uint4&, f32x4_t&) {}      // (the split kernels multiply whole slabs: mma_slab)
    // 4 consecutive fp32 k -> 4 fp16 hi (8 bytes) + 4 fp16 lo (8 bytes); v_cvt_pk_f16_f32 rounds to nearest even
    static __device__ __forceinline__ void split4(const uint4& v, uint2& hi, uint2& lo) {
        const float x0 = __uint_as_float(v.x), x1 = __uint_as_float(v.y), x2 = __uint_as_float(v.z), x3 = __uint_as_float(v.w);
        const f16x2_t h01 = __builtin_convertvector(f32x2_hw_t{x0, x1}, f16x2_t), h23 = __builtin_convertvector(f32x2_hw_t{x2, x3}, f16x2_t);
        const f16x2_t l01 = __builtin_convertvector(f32x2_hw_t{x0 - (float)h01[0], x1 - (float)h01[1]}, f16x2_t);
        const f16x2_t l23 = __builtin_convertvector(f32x2_hw_t{x2 - (float)h23[0], x3 - (float)h23[1]}, f16x2_t);
        hi = make_uint2(__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23));
        lo = make_uint2(__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23));
    }
    static __device__ __forceinline__ void mma_h(const uint4& w, const uint4& x, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w), __builtin_bit_cast(f16x8_t, x), acc, 0, 0, 0);
    }
    // one 32-k slab of a 64 x 64 wave tile: wf[0] / xf[0] = hi fragments, wf[1] / xf[1] = lo fragments.  The correction terms go
    // first and every accumulator is touched once per term, so consecutive MFMAs never depend on each other.
    static __device__ __forceinline__ void mma_slab(const uint4 (&wf)[2][4], const uint4 (&xf)[2][4], f32x4_t (&acc)[4][4]) {
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) mma_h(wf[1][ci], xf[0][ti], acc[ci][ti]);      // w_lo . a_hi
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) mma_h(wf[0][ci], xf[1][ti], acc[ci][ti]);      // w_hi . a_lo
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) mma_h(wf[0][ci], xf[0][ti], acc[ci][ti]);      // w_hi . a_hi
    }
};

template <typename OutT> struct Out;
template <> struct Out<float> {
    using raw4 = float4;
    static __device__ __forceinline__ void unpack4(const float4& t, float (&v)[4]) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void ld4(const float* p, float (&v)[4]) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void st4(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Out<uint16_t> {
    using raw4 = uint2;
    static __device__ __forceinline__ void unpack4(const uint2& t, float (&v)[4]) {
        v[0] = h16_lo(t.x); v[1] = h16_hi(t.x); v[2] = h16_lo(t.y); v[3] = h16_hi(t.y); }
    static __device__ __forceinline__ void ld4(const uint16_t* p, float (&v)[4]) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = h16_lo(t.x); v[1] = h16_hi(t.x); v[2] = h16_lo(t.y); v[3] = h16_hi(t.y); }
    static __device__ __forceinline__ void st4(uint16_t* p, const float (&v)[4]) { *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])); }
    static __device__ __forceinline__ float ld(const uint16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(uint16_t* p, float v) { *p = f32_to_bf16(v); }
};

// Epilogue of one 64x64 wave sub-tile: lane (g,n) holds, for ti = 0..3 and ci = 0..3, channels
// ch0 + ci*16 + r (r = 0..3) of token tok0 + ti*16.  Zeroes the accumulators for the next tile.
//
// Interior sub-tiles take the batched path: EVERY load of the epilogue (bias, padding mask, residual) is issued
// before the first store.  Loads and stores share the one in-order vmcnt counter on gfx9-class ISAs, so a load
// issued after a store can only be waited for with vmcnt(0), i.e. together with that store's full round trip to
// L2/HBM.  The first version interleaved them per 4-channel group: 16 serialised round trips per tile, measured
// (cycle attribution, tools/profile_kernels.py --only gemm_trace) at 2/3 of the K = 256 kernels' time.
// CF >= 0: the epilogue flags are a compile-time constant (the hot combinations get their own kernel instantiation): the
// timeline trace put the generic epilogue at ~50 VALU instructions + 6 branches per 4-channel group -- 40% of a K = 256 tile,
// VALU-bound -- most of them for options a given call does not use.  CF < 0: runtime flags (every other combination).
template <typename OutT, int CF = -1>
__device__ __forceinline__ void epilogue_tile(f32x4_t (&acc)[4][4], OutT* __restrict__ C, const float* __restrict__ bias,
                                              const OutT* __restrict__ residual, const uint8_t* __restrict__ row_mask,
                                              int M, int N, int flags_rt, int tok0, int ch0 TL_PARAMS, int res_rows = 0)
{
    const int flags = CF >= 0 ? CF : flags_rt;
    // row-broadcast residual (res_rows > 0): the encoder's [offsets | logits] projection of an unpadded batch, (src + pos) W^T + b =
    // src W^T + (pos W^T + b) with the second term ONE [S, N] matrix for every image (L2-resident)
#define RES_ROW(TOK) (res_rows > 0 ? (long)((TOK) % res_rows) : (long)(TOK))
    if constexpr (sizeof(OutT) == 4 && CF < 0) {
        if (flags & EPI_ROWMAX) {
            // Row-max epilogue (two-stage selection: torch.topk needs only max_c of the class head, deformable_transformer.py:345):
            // a lane holds 16 channels of each of its 4 tokens; the token's other 48 channels of this sub-tile sit in the lanes
            // g' != g (xor 16, 32).  One float atomic max per (token, 64-channel sub-tile) instead of a T x C fp32 matrix in HBM
            // (6.4 GB per step for the 7356-class head at bs = 32) and a separate reduction pass.  max is order-independent, so the
            // result is deterministic.  Float max through integer atomics: non-negative floats order like ints, negative ones
            // in reverse as unsigned ints; the destination is pre-filled with -inf.
            const int lane_ = (int)threadIdx.x & 63;
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                const int tok = tok0 + ti * 16;
                float mx = -__builtin_huge_valf();
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    const int ch = ch0 + ci * 16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (ch + r < N) {
                            const float v = acc[ci][ti][r] + ((flags & EPI_BIAS) ? bias[ch + r] : 0.f);
                            mx = fmaxf(mx, v);
                        }
                    }
                    acc[ci][ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                if ((lane_ >> 4) == 0 && tok < M && mx > -__builtin_huge_valf()) {
                    float* dst = reinterpret_cast<float*>(C) + tok;
                    if (mx >= 0.f) atomicMax(reinterpret_cast<int*>(dst), __float_as_int(mx));
                    else atomicMin(reinterpret_cast<unsigned int*>(dst), __float_as_uint(mx));
                }
            }
            return;
        }
    }
    const bool vec_ok = (N & 3) == 0;
    // wave-uniform test (the whole 64x64 sub-tile is interior): the paired bf16 stores exchange data between lanes
    const int lane_ = (int)threadIdx.x & 63;
    if (vec_ok && (tok0 - (lane_ & 15)) + 63 < M && (ch0 - 4 * (lane_ >> 4)) + 63 < N) {
        float4 bv[4];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
            bv[ci] = (flags & EPI_BIAS) ? *reinterpret_cast<const float4*>(bias + ch0 + ci * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
        bool masked[4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) masked[ti] = (flags & EPI_ROWMASK) && row_mask[tok0 + ti * 16];
        TL_EV(20)                                                  // epilogue entered, bias / mask loads issued
        constexpr int TB = sizeof(OutT) == 2 ? 2 : 1;              // token groups per batch (register budget: 128 VGPRs)
#pragma unroll
        for (int t0 = 0; t0 < 4; t0 += TB) {
            typename Out<OutT>::raw4 rr[TB][4];
            if (flags & EPI_RESIDUAL) {
#pragma unroll
                for (int tb = 0; tb < TB; ++tb)
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci)
                        rr[tb][ci] = *reinterpret_cast<const typename Out<OutT>::raw4*>(residual + RES_ROW(tok0 + (t0 + tb) * 16) * N + ch0 + ci * 16);
            }
            TL_EV(21)                                              // this batch's residual loads issued
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                const int ti = t0 + tb;
                uint32_t pk_lo = 0, pk_hi = 0;
                (void)pk_lo; (void)pk_hi;
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    float v[4] = {acc[ci][ti][0] + bv[ci].x, acc[ci][ti][1] + bv[ci].y, acc[ci][ti][2] + bv[ci].z, acc[ci][ti][3] + bv[ci].w};
                    acc[ci][ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    if (flags & EPI_RELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    if (flags & EPI_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752f));
                    }
                    if (masked[ti]) { v[0] = v[1] = v[2] = v[3] = 0.f; }
                    if (flags & EPI_RESIDUAL) { float q[4]; Out<OutT>::unpack4(rr[tb][ci], q); v[0] += q[0]; v[1] += q[1]; v[2] += q[2]; v[3] += q[3]; }
                    if (flags & EPI_RELU_POST) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    if (ABLATE(DBG_NO_STORE)) asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
                    else if constexpr (sizeof(OutT) == 2) {
                        // bf16: pair the channel tiles (ci, ci+1) and exchange halves between lane rows g and g^1
                        // (v_permlane16_swap): a lane then owns 8 consecutive channels = one 16-byte store, and a token's
                        // four lanes write 64 contiguous bytes per instruction instead of 32.  Non-temporal stores were
                        // measured 28% slower here (the partial lines then reach memory unmerged), so the L2 stays in the path.
                        const uint32_t lo = pack_bf16x2(v[0], v[1]), hi = pack_bf16x2(v[2], v[3]);
                        if ((ci & 1) == 0) { pk_lo = lo; pk_hi = hi; }
                        else {
                            const auto s0 = __builtin_amdgcn_permlane16_swap(pk_lo, lo, false, false);
                            const auto s1 = __builtin_amdgcn_permlane16_swap(pk_hi, hi, false, false);
                            // even g: channels 8*(g/2).. of tile ci-1 ; odd g: the same 8 channels of tile ci
                            const int g_ = lane_ >> 4;
                            uint16_t* dst = C + (long)(tok0 + ti * 16) * N + (ch0 - 4 * g_) + (ci - 1 + (g_ & 1)) * 16 + 8 * (g_ >> 1);
                            *reinterpret_cast<uint4*>(dst) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                        }
                    }
                    else Out<OutT>::st4(C + (long)(tok0 + ti * 16) * N + ch0 + ci * 16, v);
                }
            }
            TL_EV(22)                                              // this batch's stores issued
        }
        return;
    }
    // edge sub-tiles (ragged M or N, or N not a multiple of 4): per-group bounds checks, scalar tails
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
        const int tok = tok0 + ti * 16;
        const bool tok_ok = tok < M;
        const bool masked = tok_ok && (flags & EPI_ROWMASK) && row_mask[tok];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
            const int ch = ch0 + ci * 16;
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
            if (flags & EPI_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752f));
            }
            if (masked) { v[0] = v[1] = v[2] = v[3] = 0.f; }
            OutT* cptr = C + (long)tok * N + ch;
            if (full) {
                if (flags & EPI_RESIDUAL) { float rr[4]; Out<OutT>::ld4(residual + RES_ROW(tok) * N + ch, rr); v[0] += rr[0]; v[1] += rr[1]; v[2] += rr[2]; v[3] += rr[3]; }
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
                        if (flags & EPI_RESIDUAL) x += Out<OutT>::ld(residual + RES_ROW(tok) * N + ch + r);
                        if (flags & EPI_RELU_POST) x = fmaxf(x, 0.f);
                        Out<OutT>::st(cptr + r, x);
                    }
            }
        }
    }
}

#undef RES_ROW

// Implicit-GEMM convolution (CONV = true): the "A" operand is gathered on the fly from an NHWC image,
// M = B*Ho*Wo output pixels, K = KH*KW*Cin ordered (kh, kw, ci) so that a 128-byte K slab lies inside
// one filter tap (Cin*sizeof(T) % 128 == 0) and is one contiguous, 16-byte-aligned run of channels;
// taps that fall into the zero padding contribute zeros.  Weights are packed [Cout][KH][KW][Cin].
struct ConvP { int H, W, Cin, Ho, Wo, KH, KW, stride, pad;
               int a2_rows;         // plain GEMM with an A2 prologue: A2 has a2_rows rows, row m of A pairs with row m % a2_rows (0 = M rows)
               int lda;             // plain GEMM: elements between consecutive rows of A (0 = K: contiguous); K columns of a wider matrix otherwise
               int res_rows; };     // residual epilogue: the residual has res_rows rows, row m of C takes row m % res_rows (0 = M rows)
__device__ __attribute__((aligned(16))) unsigned int g_zero_line[4] = {0u, 0u, 0u, 0u};   // what a padding tap reads

// Tile chains.  A workgroup owns `tiles_per_block` consecutive TOKEN tiles (tm) of ONE channel tile (tn) and
// walks them as one continuously pipelined slab stream (load-latency prologue paid once per chain).  Workgroup
// ids are remapped so that the nN chains covering the same token panels -- tn = 0..nN-1 of one chain group --
// get consecutive logical ids inside ONE XCD (the dispatcher puts workgroup b on XCD b % 8): they stream the
// same activation rows at the same time, so the panel is fetched from HBM once and the other nN-1 readers
// hit that XCD's L2.  (First version: tn fastest inside a chain -> the re-read of a panel came a full K sweep
// later, after 64 co-resident chains had pushed it out of the 4 MiB L2; operand delivery then ran at
// beyond-L2 bandwidth, 4-8 TB/s, which capped the 128x128 tile at 300-500 TFLOP/s.)  Speed only: any
// placement gives the same results.   `ntiles` carries nM (token tiles).
#define CHAIN_SETUP()                                                                              \
    const int nM_ = ntiles;                                                                        \
    const int nwg_ = (int)gridDim.x, q8_ = nwg_ >> 3, r8_ = nwg_ & 7, xcd_ = (int)blockIdx.x & 7; \
    const int logical0_ = (xcd_ < r8_ ? xcd_ * (q8_ + 1) : r8_ * (q8_ + 1) + (xcd_ - r8_) * q8_) + ((int)blockIdx.x >> 3); \
    /* split-K (KSPLIT > 1, small grids only): the grid is KSPLIT copies of the tile grid; copy s multiplies K slabs */ \
    /* [s nk, (s+1) nk) and writes its fp32 partial tile to slice s of the workspace C points to */ \
    const int per_split_ = nwg_ / KSPLIT;                                                          \
    const int split_ = logical0_ / per_split_, logical_ = logical0_ % per_split_;                  \
    const int koff = split_ * nk;                                                                  \
    (void)koff;                                                                                    \
    const int tn = logical_ % nN;                                                                  \
    const int t_begin = (logical_ / nN) * tiles_per_block;                                         \
    const int t_end = min(t_begin + tiles_per_block, nM_);                                         \
    if (t_begin >= t_end) return;                                                                  \
    const int total = (t_end - t_begin) * nk;

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
    constexpr int KSPLIT = 1;
    TR(unsigned long long* tl_none_ = nullptr; int tl_n_none_ = 0; (void)tl_none_; (void)tl_n_none_;)

    // PERSISTENT tile chain: this block owns tiles [t_begin, t_end) of the (tm, tn) grid, tn fastest, and
    // walks them as ONE flat stream of K slabs, software-pipelined across tile boundaries: the global
    // loads of the next tile's first slab are in flight while the current tile's last slab is being
    // multiplied.  With K = 256 a tile is only 4 slabs, so paying the ~2 us load latency once per
    // block instead of once per tile is worth more than any in-tile tuning; consecutive tiles of a
    // chain share their activation rows (tn fastest), re-read from L2.
    CHAIN_SETUP()

    // staging: each operand tile = 128 rows x 128 B = 1024 16-byte chunks; thread t takes rows srow + 32 i
    // (8 consecutive lanes cover one 128-byte row slab), kc = tid & 7.
    const int srow = tid >> 3, kc = tid & 7;
    // LDS image: row r = 128 bytes, its 16-byte chunk c stored at chunk position c ^ (r & 7).  For
    // ds_read_b128 (serviced in the non-contiguous 16-lane groups {0-3,12-15,20-27}, ...) the 16 lanes of a
    // group then hit 16 distinct 16-byte slots of the 256-byte bank row; a padded stride (144 B, the first
    // version) is NOT conflict-free for those groups -- 7 of 16 lanes collided (measured: the LDS-read +
    // barrier skeleton alone was a third of the kernel time).
    const int lds0 = srow * LDS_ROW + ((kc ^ (srow & 7)) * 16);
    long a_off[4], w_off[4], a2_off[4];                         // loader state: byte offsets at k-slab 0
    int hi0[4], wi0[4];                                         // CONV: top-left input coordinate of the pixel
    const int slabs_per_tap = CONV ? (cp.Cin * (int)sizeof(T)) / SLAB : 1;
    const char* Ab = reinterpret_cast<const char*>(A);
    const char* A2b = reinterpret_cast<const char*>(A2);
    const char* Wb = reinterpret_cast<const char*>(W);
    const char* zero_line = reinterpret_cast<const char*>(g_zero_line);
    (void)zero_line; (void)a2_off;

#define SET_LOAD_TILE(TILE)                                                                        \
    {                                                                                              \
        const int lm0_ = (TILE) * BM, ln0_ = tn * BN;                                              \
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
                a_off[i] = (ar * (long)(cp.lda > 0 ? cp.lda : K)) * (long)sizeof(T) + kc * 16;                                   \
            }                                                                                      \
            a2_off[i] = HAS_A2 ? (((cp.a2_rows > 0 ? ar % cp.a2_rows : ar) * K) * (long)sizeof(T) + kc * 16) : 0; \
        }                                                                                          \
    }

    // Staging registers are named scalars on purpose: as arrays written under a condition hipcc
    // (ROCm 7.2) leaves them in scratch memory (global_load -> scratch_store ... scratch_load -> ds_write).
    // TWO sets (P, Q): the loads of slab s+2 are issued before slab s is multiplied and are consumed
    // (stored to LDS) after slab s+1 has been multiplied -- two compute phases to cover the L2/HBM
    // latency.  With one set (distance 1) the bf16 kernel stalled ~2500 cycles per slab at the LDS
    // store (fp32 MFMA work is 4x longer and hid it: 65% of its peak vs 15% for bf16).
    uint4 raP0, raP1, raP2, raP3, rwP0, rwP1, rwP2, rwP3;
    uint4 raQ0, raQ1, raQ2, raQ3, rwQ0, rwQ1, rwQ2, rwQ3;
    // asm (uncounted) loads only where the register budget leaves no spills: a spilled in-flight register
    // would be copied before its data lands.  The A2 and fp32-conv variants keep compiler-counted loads.
    constexpr bool ASM_LOADS = !HAS_A2 && !(CONV && sizeof(T) == 4);
    constexpr int LOADS_PER_SLAB = 8;
#define LD16(PTR) (ASM_LOADS ? asm_load16(PTR) : *reinterpret_cast<const uint4*>(PTR))
#define GLOAD1(S, I, OFF)                                                                          \
    rw##S##I = LD16(Wb + w_off[I] + (OFF));                                                        \
    if (CONV) {                                                                                    \
        const int hi_ = hi0[I] + kh_, wi_ = wi0[I] + kw_;                                          \
        const bool ok_ = hi_ >= 0 && hi_ < cp.H && wi_ >= 0 && wi_ < cp.W;                         \
        /* a tap in the zero padding reads the shared zero line instead (no select on in-flight data) */ \
        const long po_ = ((long)hi_ * cp.W + wi_) * cp.Cin * (long)sizeof(T) + coff_;              \
        ra##S##I = LD16(ok_ ? Ab + a_off[I] + po_ : zero_line);                                    \
    } else {                                                                                       \
        ra##S##I = LD16(Ab + a_off[I] + (OFF));                                                    \
        if (HAS_A2) ra##S##I = GT<T>::add(ra##S##I, *reinterpret_cast<const uint4*>(A2b + a2_off[I] + (OFF))); \
    }
#define GLOAD(S, KT)                                                                               \
    {                                                                                              \
        const long off_ = (long)(KT) * SLAB;                                                       \
        const int tap_ = (KT) / slabs_per_tap;                                                     \
        const int kh_ = CONV ? tap_ / cp.KW : 0, kw_ = CONV ? tap_ % cp.KW : 0;                    \
        const long coff_ = (long)((KT) % slabs_per_tap) * SLAB;                                    \
        (void)kh_; (void)kw_; (void)coff_;                                                         \
        GLOAD1(S, 0, off_) GLOAD1(S, 1, off_) GLOAD1(S, 2, off_) GLOAD1(S, 3, off_)                \
    }
#define LSTORE1(S, I)                                                                              \
    *reinterpret_cast<uint4*>(wt_ + I * 32 * LDS_ROW) = rw##S##I;                                  \
    *reinterpret_cast<uint4*>(wt_ + TILE_BYTES + I * 32 * LDS_ROW) = ra##S##I;
#define LSTORE(S, STAGE)                                                                           \
    {                                                                                              \
        unsigned char* wt_ = smem + (STAGE) * 2 * TILE_BYTES + lds0;                               \
        LSTORE1(S, 0) LSTORE1(S, 1) LSTORE1(S, 2) LSTORE1(S, 3)                                    \
    }
// advance the loader by one slab (possibly into the next tile of the chain) and issue its loads into set S
#define ADVANCE_AND_LOAD(S)                                                                        \
    {                                                                                              \
        if (++lkt == nk) { lkt = 0; ++ltile; SET_LOAD_TILE(ltile) }                                \
        if (!ABLATE(DBG_NO_LOAD)) GLOAD(S, lkt)                                                    \
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) acc[ci][ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    SET_LOAD_TILE(t_begin)
    int lkt = 0, ltile = t_begin;            // slab whose loads were issued last
    GLOAD(P, 0)
    if (ASM_LOADS) wait_vmcnt<0>();
    LSTORE(P, 0)                             // slab 0 -> stage 0
    if (total > 1) ADVANCE_AND_LOAD(Q)       // slab 1 in flight in set Q
    __syncthreads();
    int kt = 0, tile = t_begin;              // slab being multiplied

    // one half-iteration: multiply the slab in stage CUR; before it, issue slab s+2 into set LS (the set
    // that held the slab now in stage CUR); after it, store slab s+1 (set SS, loaded one half-iteration
    // ago) into the other stage.  The loop is unrolled by two so P/Q roles are compile-time: a runtime
    // `cur ? P : Q` select makes hipcc merge the two sets through phi copies and wait vmcnt(0) at the join.
#define HALF(CUR, LS, SS)                                                                          \
    {                                                                                              \
        if (s + 2 < total) ADVANCE_AND_LOAD(LS)                                                    \
        const unsigned char* wt = smem + (CUR) * 2 * TILE_BYTES + (wn * 64 + n) * LDS_ROW;         \
        const unsigned char* xt = wt + TILE_BYTES + ((wm - wn) * 64) * LDS_ROW;                    \
        /* all 16 fragment reads of the slab are issued before the first MFMA: the matrix pipe then starts as \
           soon as the first pair lands and the second k-half's LDS latency hides behind the first half's MFMAs */ \
        uint4 wf[2][4], xf[2][4];                                                                  \
        _Pragma("unroll") for (int kq = 0; kq < 2; ++kq) {                                         \
            const int sw = (((kq * 4 + g) ^ (n & 7)) * 16);      /* rows i*16 + n: (row & 7) == (n & 7) */ \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                        \
                wf[kq][i] = *reinterpret_cast<const uint4*>(wt + i * 16 * LDS_ROW + sw);           \
                xf[kq][i] = *reinterpret_cast<const uint4*>(xt + i * 16 * LDS_ROW + sw);           \
            }                                                                                      \
        }                                                                                          \
        _Pragma("unroll") for (int kq = 0; kq < 2; ++kq) {                                         \
            if (!ABLATE(DBG_NO_MMA)) {                                                             \
                _Pragma("unroll") for (int ci = 0; ci < 4; ++ci)                                   \
                    _Pragma("unroll") for (int ti = 0; ti < 4; ++ti) GT<T>::mma(wf[kq][ci], xf[kq][ti], acc[ci][ti]); \
            } else {                                                                               \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) { asm volatile("" :: "v"(wf[kq][i].x), "v"(xf[kq][i].x)); } \
            }                                                                                      \
        }                                                                                          \
        if (s + 1 < total && !ABLATE(DBG_NO_LDS)) {                                                \
            /* slab s+1's loads are older than the LOADS_PER_SLAB loads of slab s+2 (if any were issued) */ \
            if (ASM_LOADS) { if (s + 2 < total) wait_vmcnt<LOADS_PER_SLAB>(); else wait_vmcnt<0>(); } \
            LSTORE(SS, 1 - (CUR))                                                                  \
        }                                                                                          \
        __syncthreads();                                                                           \
        if (++kt == nk) { EPILOGUE() kt = 0; ++tile; }                                             \
        ++s;                                                                                       \
    }

#define EPILOGUE()                                                                                 \
    {                                                                                              \
        const int m0 = tile * BM, n0 = tn * BN;                                                    \
        epilogue_tile<OutT>(acc, C, bias, residual, row_mask, M, N, flags, m0 + wm * 64 + n, n0 + wn * 64 + 4 * g TL_ARGS_NONE, cp.res_rows); \
    }

    int s = 0;
    while (s < total) {
        HALF(0, P, Q)
        if (s >= total) break;
        HALF(1, Q, P)
    }
#undef HALF
#undef EPILOGUE
#undef LD16
#undef GLOAD
#undef LSTORE
#undef GLOAD1
#undef LSTORE1
#undef SET_LOAD_TILE
#undef ADVANCE_AND_LOAD
}

// ---------------------------------------------------------------------------------------------
// Wave-specialised variant (the default): 512 threads = 4 MFMA waves + 4 loader waves, one workgroup per CU.
//
// Measured on the uniform kernel above (tools/profile_kernels.py --only gemm_ablate, K = 2048): the phase costs
// are ADDITIVE -- LDS reads + barriers 123 us, LDS stores ~45 us, MFMA ~60 us, global loads ~108 us of 364 us --
// because the two resident workgroups of a CU execute the same phases in lock-step and contend instead of
// overlapping.  Here the phases run on different waves at the same time:
//   loader waves (4..7): slab s+1's registers -> LDS stage (s+1)&1, then issue slab s+3's global loads
//                        (two register sets, inline-asm loads, one counted vmcnt wait: two full slabs of latency cover)
//   MFMA waves   (0..3): 16 ds_read_b128 + 32 MFMA on stage s&1
// and meet at ONE barrier per slab.  Same tile chain, same LDS image, same epilogue as above.
// ---------------------------------------------------------------------------------------------
template <typename T, typename OutT, bool HAS_A2, bool CONV, int CF = -1>
__global__ __launch_bounds__(512, HAS_A2 ? 2 : 4) void gemm_ws_kernel(
    const T* __restrict__ A, const T* __restrict__ A2, const T* __restrict__ W,
    const float* __restrict__ bias, const OutT* __restrict__ residual, const uint8_t* __restrict__ row_mask,
    OutT* __restrict__ C, int M, int N, int K, int flags, int nN, int ntiles, int tiles_per_block, ConvP cp, int KSPLIT)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 stages][W tile | X tile]
    const int wave = threadIdx.x >> 6;
    constexpr int BK = GT<T>::BK;
    const int nk = K / BK / KSPLIT;                                // slabs this workgroup multiplies per tile
    CHAIN_SETUP()
    C += (long)split_ * M * N;

    if (wave >= 4) {
        // ================================ loader role ================================
        const int tid = threadIdx.x - 256;
        const int srow = tid >> 3, kc = tid & 7;
        const int lds0 = srow * LDS_ROW + ((kc ^ lds_swz<T>(srow)) * 16);
        // split operands: this lane's 4 k (4 kc ..) land in half kc & 1 of chunk kc >> 1 (hi) and of chunk 4 + (kc >> 1) (lo)
        const int xh_ = (((kc >> 1) ^ lds_swz<T>(srow)) * 16) + (kc & 1) * 8, xl_ = (((4 + (kc >> 1)) ^ lds_swz<T>(srow)) * 16) + (kc & 1) * 8;
        (void)xh_; (void)xl_;
        long a_off[4], w_off[4], a2_off[4];
        int hi0[4], wi0[4];
        const int slabs_per_tap = CONV ? (cp.Cin * (int)sizeof(T)) / SLAB : 1;
        const char* Ab = reinterpret_cast<const char*>(A);
        const char* A2b = reinterpret_cast<const char*>(A2);
        const char* Wb = reinterpret_cast<const char*>(W);
        const char* zero_line = reinterpret_cast<const char*>(g_zero_line);
        (void)zero_line; (void)A2b; (void)a2_off;
        // asm (uncounted) loads only in the variants whose ISA audit shows no compiler copy of an in-flight register
        // (the A2 variant gets v_mov shuffles right after the loads); A2 keeps compiler-counted loads -- conservative
        // waits, but they only stall loader waves.
        constexpr bool ASM = !HAS_A2;
        constexpr int NLOAD = 8;                               // asm loads per slab per lane
#define WS_SET_TILE(TILE)                                                                          \
        {                                                                                          \
            const int lm0_ = (TILE) * BM, ln0_ = tn * BN;                                          \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                        \
                const long ar = min(lm0_ + srow + 32 * i, M - 1), wr = min(ln0_ + srow + 32 * i, N - 1); \
                w_off[i] = (wr * K) * (long)sizeof(T) + kc * 16;                                   \
                if (CONV) {                                                                        \
                    const int hw = cp.Ho * cp.Wo;                                                  \
                    const int bimg = (int)(ar / hw), rem = (int)(ar % hw);                         \
                    hi0[i] = (rem / cp.Wo) * cp.stride - cp.pad;                                   \
                    wi0[i] = (rem % cp.Wo) * cp.stride - cp.pad;                                   \
                    a_off[i] = (long)bimg * cp.H * cp.W * cp.Cin * (long)sizeof(T) + kc * 16;      \
                } else {                                                                           \
                    hi0[i] = wi0[i] = 0;                                                           \
                    a_off[i] = (ar * (long)(cp.lda > 0 ? cp.lda : K)) * (long)sizeof(T) + kc * 16;                               \
                }                                                                                  \
                a2_off[i] = HAS_A2 ? (((cp.a2_rows > 0 ? ar % cp.a2_rows : ar) * K) * (long)sizeof(T) + kc * 16) : 0; \
            }                                                                                      \
        }
        uint4 raP0, raP1, raP2, raP3, rwP0, rwP1, rwP2, rwP3, rbP0, rbP1, rbP2, rbP3;
        uint4 raQ0, raQ1, raQ2, raQ3, rwQ0, rwQ1, rwQ2, rwQ3, rbQ0, rbQ1, rbQ2, rbQ3;
        (void)rbP0; (void)rbP1; (void)rbP2; (void)rbP3; (void)rbQ0; (void)rbQ1; (void)rbQ2; (void)rbQ3;
#define WS_LD16(PTR) (ASM ? asm_load16(PTR) : *reinterpret_cast<const uint4*>(PTR))
#define WS_GLOAD1(S, I, OFF)                                                                       \
        rw##S##I = WS_LD16(Wb + w_off[I] + (OFF));                                                 \
        if (CONV) {                                                                                \
            const int hi_ = hi0[I] + kh_, wi_ = wi0[I] + kw_;                                      \
            const bool ok_ = hi_ >= 0 && hi_ < cp.H && wi_ >= 0 && wi_ < cp.W;                     \
            const long po_ = ((long)hi_ * cp.W + wi_) * cp.Cin * (long)sizeof(T) + coff_;          \
            ra##S##I = WS_LD16(ok_ ? Ab + a_off[I] + po_ : zero_line);                             \
        } else {                                                                                   \
            ra##S##I = WS_LD16(Ab + a_off[I] + (OFF));                                             \
            if (HAS_A2) rb##S##I = WS_LD16(A2b + a2_off[I] + (OFF));                               \
        }
#define WS_GLOAD(S, KT)                                                                            \
        {                                                                                          \
            const long off_ = (long)((KT) + koff) * SLAB;                                          \
            const int tap_ = ((KT) + koff) / slabs_per_tap;                                        \
            const int kh_ = CONV ? tap_ / cp.KW : 0, kw_ = CONV ? tap_ % cp.KW : 0;                \
            const long coff_ = (long)(((KT) + koff) % slabs_per_tap) * SLAB;                       \
            (void)kh_; (void)kw_; (void)coff_;                                                     \
            WS_GLOAD1(S, 0, off_) WS_GLOAD1(S, 1, off_) WS_GLOAD1(S, 2, off_) WS_GLOAD1(S, 3, off_) \
        }
#define WS_LSTORE1(S, I)                                                                           \
        *reinterpret_cast<uint4*>(wt_ + I * 32 * LDS_ROW) = rw##S##I;                              \
        if constexpr (kSplit<T>) {      /* fp32 activations -> fp16 hi | lo halves of the row (see GT<f32s_t>) */ \
            uint2 hi_, lo_;                                                                        \
            GT<T>::split4(HAS_A2 ? GT<T>::add(ra##S##I, rb##S##I) : ra##S##I, hi_, lo_);           \
            *reinterpret_cast<uint2*>(xs_ + I * 32 * LDS_ROW + xh_) = hi_;                         \
            *reinterpret_cast<uint2*>(xs_ + I * 32 * LDS_ROW + xl_) = lo_;                         \
        } else                                                                                     \
            *reinterpret_cast<uint4*>(wt_ + TILE_BYTES + I * 32 * LDS_ROW) = HAS_A2 ? GT<T>::add(ra##S##I, rb##S##I) : ra##S##I;
#define WS_LSTORE(S, STAGE)                                                                        \
        {                                                                                          \
            unsigned char* wt_ = smem + (STAGE) * 2 * TILE_BYTES + lds0;                           \
            unsigned char* xs_ = smem + (STAGE) * 2 * TILE_BYTES + TILE_BYTES + srow * LDS_ROW;    \
            (void)xs_;                                                                             \
            WS_LSTORE1(S, 0) WS_LSTORE1(S, 1) WS_LSTORE1(S, 2) WS_LSTORE1(S, 3)                    \
        }
#define WS_ADVANCE_AND_LOAD(S)                                                                     \
        {                                                                                          \
            if (++lkt == nk) { lkt = 0; ++ltile; WS_SET_TILE(ltile) }                              \
            if (!ABLATE(DBG_NO_LOAD)) WS_GLOAD(S, lkt)                                             \
        }
        int lkt = 0, ltile = t_begin;
        TL_INIT(1)
        WS_SET_TILE(t_begin)
        WS_GLOAD(P, 0)
        if (ASM) wait_vmcnt<0>();
        WS_LSTORE(P, 0)                                          // slab 0 -> stage 0
        if (total > 1) WS_ADVANCE_AND_LOAD(Q)                    // slab 1 -> set Q (odd slabs)
        if (total > 2) WS_ADVANCE_AND_LOAD(P)                    // slab 2 -> set P (even slabs)
        __syncthreads();                                         // barrier #0: slab 0 visible
        // iteration i (the MFMA waves multiply slab i): store slab i+1, then refill its set with slab i+3
        int i = 0;
        while (i < total) {
            if (i + 1 < total) {                                 // i even: slab i+1 is odd -> set Q, stage 1
                TL_EV(1)
                if (ASM) { if (i + 2 < total) wait_vmcnt<NLOAD>(); else wait_vmcnt<0>(); }
                TL_EV(2)
                if (!ABLATE(DBG_NO_LDS)) WS_LSTORE(Q, 1)
                TL_EV(3)
                if (i + 3 < total) WS_ADVANCE_AND_LOAD(Q)
                TL_EV(4)
            }
            __syncthreads();
            TL_EV(5)
            if (++i >= total) break;
            if (i + 1 < total) {                                 // i odd: slab i+1 is even -> set P, stage 0
                TL_EV(1)
                if (ASM) { if (i + 2 < total) wait_vmcnt<NLOAD>(); else wait_vmcnt<0>(); }
                TL_EV(2)
                if (!ABLATE(DBG_NO_LDS)) WS_LSTORE(P, 0)
                TL_EV(3)
                if (i + 3 < total) WS_ADVANCE_AND_LOAD(P)
                TL_EV(4)
            }
            __syncthreads();
            TL_EV(5)
            ++i;
        }
#undef WS_LD16
#undef WS_SET_TILE
#undef WS_GLOAD1
#undef WS_GLOAD
#undef WS_LSTORE1
#undef WS_LSTORE
#undef WS_ADVANCE_AND_LOAD
        return;
    }

    // ================================ MFMA role ================================
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, n = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    f32x4_t acc[4][4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) acc[ci][ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int kt = 0, tile = t_begin;
    TL_INIT(0)
    __syncthreads();                                             // barrier #0
    TL_EV(10)
    for (int s = 0; s < total; ++s) {
        const unsigned char* wt = smem + (s & 1) * 2 * TILE_BYTES + (wn * 64 + n) * LDS_ROW;
        const unsigned char* xt = wt + TILE_BYTES + ((wm - wn) * 64) * LDS_ROW;
        uint4 wf[2][4], xf[2][4];
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
            const int sw = (((kq * 4 + g) ^ lds_swz<T>(n)) * 16);        // rows i * 16 + n: same low three bits as n
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wf[kq][i] = *reinterpret_cast<const uint4*>(wt + i * 16 * LDS_ROW + sw);
                xf[kq][i] = *reinterpret_cast<const uint4*>(xt + i * 16 * LDS_ROW + sw);
            }
        }
        if constexpr (kSplit<T>) GT<T>::mma_slab(wf, xf, acc);
        else {
#pragma unroll
        for (int kq = 0; kq < 2; ++kq)
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    if (!ABLATE(DBG_NO_MMA)) GT<T>::mma(wf[kq][ci], xf[kq][ti], acc[ci][ti]);
                    else asm volatile("" :: "v"(wf[kq][ci].x), "v"(xf[kq][ti].x));
                }
        }
        TL_EV(11)
        __syncthreads();
        TL_EV(10)
        if (++kt == nk) {
            const int m0 = tile * BM, n0 = tn * BN;
            if (!ABLATE(DBG_NO_EPI))
                epilogue_tile<OutT, CF>(acc, C, bias, residual, row_mask, M, N, flags, m0 + wm * 64 + n, n0 + wn * 64 + 4 * g TL_ARGS, cp.res_rows);
            kt = 0; ++tile;
            TL_EV(12)
        }
    }
}

// ---------------------------------------------------------------------------------------------
// "Tall" variant for N <= 64 output channels (ResNet layer1: the 3x3 64 -> 64 convolutions and the 256 -> 64 reductions, a
// quarter of the backbone's GEMM time): tile = 256 tokens x 64 channels, MFMA wave w owns tokens 64w..64w+63 x all 64 channels.
// In the 128 x 128 kernel above a 64-channel problem leaves the two wn = 1 MFMA waves multiplying clamped (duplicate) weight
// rows: half of the matrix-pipe and LDS-read work of every tile is thrown away.  Same loader / two-set pipeline / swizzled LDS
// image / epilogue; no A2, no split-K; one channel tile, so a chain is just a run of token tiles.
// ---------------------------------------------------------------------------------------------
// TBN = 128 (round 2): a 256 x 128 tile for the K >= 512 projections.  Those are bound by L2 -> LDS operand delivery, not by HBM or the
// matrix pipe: with 128 x 128 tiles every slab step moves (128 + 128) x 128 B per workgroup, 512 workgroups deep -- 16 MB per step,
// ~1.6 us at the ~10 TB/s the L2s deliver (M = 32768, N = 256, K = 1024: 16 steps = 26 us measured 27).  A 256 x 128 tile moves
// (256 + 128) rows per 2 x the flops: 0.75 x the bytes per flop.  8 MFMA waves (4 token groups x 2 channel halves, 64 x 64 each as
// before) + 4 loader waves = 768 threads, one workgroup per CU (2 x 48 KB of LDS); grid.y walks the 128-channel tiles.
constexpr int TALL_BM = 256;
template <int TBN> struct TallCfg { static constexpr int STAGE = (TBN + TALL_BM) * LDS_ROW, NMW = 4 * (TBN / 64), NT = 64 * (NMW + 4); };

// (split operands + the gathering loader need more than 128 VGPRs: two waves per SIMD there -- a spilled staging register of the hand-counted
// asm loads would be copied before its data lands)
template <typename T, typename OutT, bool CONV, int CF = -1, int TBN = 64>
__global__ __launch_bounds__(TallCfg<TBN>::NT, TBN == 64 ? ((kSplit<T> && CONV) ? 2 : 4) : 1) void gemm_ws_tall_kernel(
    const T* __restrict__ A, const T* __restrict__ W, const float* __restrict__ bias, const OutT* __restrict__ residual,
    OutT* __restrict__ C, int M, int N, int K, int flags, int ntilesM, int tiles_per_block, ConvP cp, int round_robin)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TALL_STAGE = TallCfg<TBN>::STAGE, NMW = TallCfg<TBN>::NMW, WL = TBN / 32;     // WL = weight rows per loader thread
    const int n0 = (int)blockIdx.y * TBN;                                // first output channel of this workgroup's tiles
    const int wave = threadIdx.x >> 6;
    constexpr int BK = GT<T>::BK;
    const int nk = K / BK;
    // workgroup b runs on XCD b % 8: give every XCD a CONTIGUOUS band of token tiles, so that the workgroups resident together in
    // one XCD convolve adjacent image rows and the halo rows of a 3x3 tap are served by that XCD's L2 (round-robin placement put
    // rows y-1, y, y+1 on three different XCDs: FETCH_SIZE showed the activation map read 3x, profiles/r02_step_v1_traffic.json)
    const int nwg = (int)gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = (int)blockIdx.x & 7;
    const int logical = round_robin ? (int)blockIdx.x
                      : (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ((int)blockIdx.x >> 3);
    const int t_begin = logical * tiles_per_block;
    const int t_end = min(t_begin + tiles_per_block, ntilesM);
    if (t_begin >= t_end) return;
    const int total = (t_end - t_begin) * nk;

    if (wave >= NMW) {
        // ================================ loader role ================================
        const int tid = threadIdx.x - NMW * 64;
        const int srow = tid >> 3, kc = tid & 7;
        const int lds0 = srow * LDS_ROW + ((kc ^ lds_swz<T>(srow)) * 16);
        const int slabs_per_tap = CONV ? (cp.Cin * (int)sizeof(T)) / SLAB : 1;
        const char* Ab = reinterpret_cast<const char*>(A);
        const char* Wb = reinterpret_cast<const char*>(W);
        const char* zero_line = reinterpret_cast<const char*>(g_zero_line);
        (void)zero_line;
        long a_off[8], w_off[WL];
        int hi0[8], wi0[8];
        auto set_tile = [&](int tile) {
            const int lm0 = tile * TALL_BM;
#pragma unroll
            for (int i = 0; i < WL; ++i) w_off[i] = ((long)min(n0 + srow + 32 * i, N - 1) * K) * (long)sizeof(T) + kc * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long ar = min(lm0 + srow + 32 * i, M - 1);
                if (CONV) {
                    const int hw = cp.Ho * cp.Wo;
                    const int bimg = (int)(ar / hw), rem = (int)(ar % hw);
                    hi0[i] = (rem / cp.Wo) * cp.stride - cp.pad;
                    wi0[i] = (rem % cp.Wo) * cp.stride - cp.pad;
                    a_off[i] = (long)bimg * cp.H * cp.W * cp.Cin * (long)sizeof(T) + kc * 16;
                } else {
                    hi0[i] = wi0[i] = 0;
                    a_off[i] = (ar * (long)(cp.lda > 0 ? cp.lda : K)) * (long)sizeof(T) + kc * 16;
                }
            }
        };
        uint4 ra[2][8], rw[2][WL];
        auto gload = [&](auto S_, int kt) {
            constexpr int S = decltype(S_)::value;
            const long off = (long)kt * SLAB;
            const int tap = kt / slabs_per_tap;
            const int kh = CONV ? tap / cp.KW : 0, kw = CONV ? tap % cp.KW : 0;
            const long coff = (long)(kt % slabs_per_tap) * SLAB;
#pragma unroll
            for (int i = 0; i < WL; ++i) rw[S][i] = asm_load16(Wb + w_off[i] + off);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (CONV) {
                    const int hi = hi0[i] + kh, wi = wi0[i] + kw;
                    const bool ok = hi >= 0 && hi < cp.H && wi >= 0 && wi < cp.W;
                    const long po = ((long)hi * cp.W + wi) * cp.Cin * (long)sizeof(T) + coff;
                    ra[S][i] = asm_load16(ok ? Ab + a_off[i] + po : zero_line);
                } else {
                    ra[S][i] = asm_load16(Ab + a_off[i] + off);
                }
            }
        };
        auto lstore = [&](auto S_, int stage) {
            constexpr int S = decltype(S_)::value;
            unsigned char* wt = smem + stage * TALL_STAGE + lds0;
#pragma unroll
            for (int i = 0; i < WL; ++i) *reinterpret_cast<uint4*>(wt + i * 32 * LDS_ROW) = rw[S][i];
            if constexpr (kSplit<T>) {      // fp32 activations -> fp16 hi | lo halves of the row (see GT<f32s_t>)
                unsigned char* xs = smem + stage * TALL_STAGE + (TBN + srow) * LDS_ROW;
                const int xh = (((kc >> 1) ^ lds_swz<T>(srow)) * 16) + (kc & 1) * 8, xl = (((4 + (kc >> 1)) ^ lds_swz<T>(srow)) * 16) + (kc & 1) * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    uint2 hi, lo;
                    GT<T>::split4(ra[S][i], hi, lo);
                    *reinterpret_cast<uint2*>(xs + i * 32 * LDS_ROW + xh) = hi;
                    *reinterpret_cast<uint2*>(xs + i * 32 * LDS_ROW + xl) = lo;
                }
            } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(wt + TBN * LDS_ROW + i * 32 * LDS_ROW) = ra[S][i];
            }
        };
        int lkt = 0, ltile = t_begin;
        auto advance_and_load = [&](auto S_) {
            if (++lkt == nk) { lkt = 0; ++ltile; set_tile(ltile); }
            gload(S_, lkt);
        };
        using P_ = std::integral_constant<int, 0>;
        using Q_ = std::integral_constant<int, 1>;
        constexpr int NLOAD = 8 + WL;
        set_tile(t_begin);
        gload(P_{}, 0);
        wait_vmcnt<0>();
        lstore(P_{}, 0);
        if (total > 1) advance_and_load(Q_{});
        if (total > 2) advance_and_load(P_{});
        __syncthreads();
        int i = 0;
        while (i < total) {
            if (i + 1 < total) {
                if (i + 2 < total) wait_vmcnt<NLOAD>(); else wait_vmcnt<0>();
                lstore(Q_{}, 1);
                if (i + 3 < total) advance_and_load(Q_{});
            }
            __syncthreads();
            if (++i >= total) break;
            if (i + 1 < total) {
                if (i + 2 < total) wait_vmcnt<NLOAD>(); else wait_vmcnt<0>();
                lstore(P_{}, 0);
                if (i + 3 < total) advance_and_load(P_{});
            }
            __syncthreads();
            ++i;
        }
        return;
    }

    // ================================ MFMA role ================================
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, n = lane & 15;
    f32x4_t acc[4][4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) acc[ci][ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int kt = 0, tile = t_begin;
    TR(unsigned long long* tl_none_ = nullptr; int tl_n_none_ = 0; (void)tl_none_; (void)tl_n_none_;)
    __syncthreads();
    for (int s = 0; s < total; ++s) {
        const unsigned char* wt = smem + (s & 1) * TALL_STAGE + ((wave >> 2) * 64 + n) * LDS_ROW;
        const unsigned char* xt = smem + (s & 1) * TALL_STAGE + (TBN + (wave & 3) * 64 + n) * LDS_ROW;
        uint4 wf[2][4], xf[2][4];
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
            const int sw = (((kq * 4 + g) ^ lds_swz<T>(n)) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wf[kq][i] = *reinterpret_cast<const uint4*>(wt + i * 16 * LDS_ROW + sw);
                xf[kq][i] = *reinterpret_cast<const uint4*>(xt + i * 16 * LDS_ROW + sw);
            }
        }
        if constexpr (kSplit<T>) GT<T>::mma_slab(wf, xf, acc);
        else {
#pragma unroll
        for (int kq = 0; kq < 2; ++kq)
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) GT<T>::mma(wf[kq][ci], xf[kq][ti], acc[ci][ti]);
        }
        __syncthreads();
        if (++kt == nk) {
            epilogue_tile<OutT, CF>(acc, C, bias, residual, (const uint8_t*)nullptr, M, N, flags, tile * TALL_BM + (wave & 3) * 64 + n,
                                    n0 + (wave >> 2) * 64 + 4 * g TL_ARGS_NONE);
            kt = 0; ++tile;
        }
    }
}

// tiles per block: enough chains to fill the chip (2 resident workgroups x 256 CUs) a few times over
// wave-specialised kernel: ONE workgroup (8 waves) per CU -> chains sized for 256 x 3 workgroups
static inline int plan_chain_ws(long ntiles) {
    const long target_blocks = 2 * 256 * 2;     // two 8-wave workgroups fit a CU (99 VGPRs, 64 KB LDS)
    long per = (ntiles + target_blocks - 1) / target_blocks;
    if (per < 1) per = 1;
    if (per > 128) per = 128;
    return (int)per;
}
static inline bool use_ws() {
    static const int v = exp_env_int("DTLR_GEMM_WS", 1);      // experiment builds: =0 uniform kernel (A/B timing)
    return v == 1;
}

static inline int plan_chain(long ntiles) {
    const long target_blocks = 2 * 256 * 2;
    long per = (ntiles + target_blocks - 1) / target_blocks;
    if (per < 1) per = 1;
    if (per > 64) per = 64;
    return (int)per;
}

// ---- split-K for grids that cannot fill the chip (input_proj[3]: 3x3 stride-2 conv 2048 -> 256 on a 2 x 32 map is 32
// tiles of K = 18432: 0.23 ms on 32 of 256 CUs).  KSPLIT copies of the tile grid each multiply a K range into an fp32
// partial tile (slice s of a workspace); a second small kernel sums the slices and applies the epilogue.  Deterministic
// (no atomics): the slices are added in index order.
// One workspace per (device, STREAM) (round 5; it was per device): the partial tiles of a launch live in it until that launch's reduce
// kernel has run, so two forwards a caller overlaps on two streams of one device must not share it -- with one buffer per device the
// second forward's partial tiles would overwrite the first's between its two kernels (a silent wrong result; the engine itself uses
// one stream and never hit it).  A slot is found by linear search (a handful of streams per process).
// LIFETIME (round 6): a pointer handed out here may be baked into a captured HIP graph, so a buffer is NEVER freed or moved once it
// has been returned: a larger request on the same slot allocates a NEW buffer and RETIRES the old one (it stays allocated for the
// life of the process; sizes grow in powers of two from 1 MiB, so all retired buffers of a slot together are smaller than its live one).
// Round 5 hipFree()d on growth: "capture a 1-line graph, run one larger eager batch, replay" replayed into freed memory.
// Under stream capture no allocation is possible: a stream without a (large enough) slot borrows the largest buffer of its device
// (a captured graph is one linear chain; overlapping its replays with other work on that buffer's stream is the caller's to order),
// and with none the callers fall back to their unsplit launch -- which sums in another order, so callers that capture should call
// dtlr_workspace_reserve() on the capture stream first (DTLREngine does, before its first forward): then eager and captured launches
// of one shape take the same kernels.
struct SplitKSlot { int dev; hipStream_t st; float* p; size_t bytes; };
static SplitKSlot g_splitk_slots[64] = {};
static int g_splitk_n = 0;
static std::mutex g_splitk_mu;
static size_t g_retired_bytes = 0;                             // kept alive on purpose (see LIFETIME): only counted
float* stream_workspace(size_t bytes, hipStream_t st);         // (also used by ffn_split.hip's hidden-dimension split: declared in dtlr_common.h)
static float* splitk_workspace(size_t bytes, hipStream_t st) { return stream_workspace(bytes, st); }
static inline size_t ws_round_up(size_t bytes) {
    size_t r = (size_t)1 << 20;
    while (r < bytes) r <<= 1;
    return r;
}
float* stream_workspace(size_t bytes, hipStream_t st) {
    int d = 0;
    (void)hipGetDevice(&d);
    std::lock_guard<std::mutex> lk(g_splitk_mu);
    SplitKSlot* slot = nullptr;
    SplitKSlot* biggest = nullptr;
    for (int i = 0; i < g_splitk_n; ++i) {
        SplitKSlot& s = g_splitk_slots[i];
        if (s.dev != d) continue;
        if (s.st == st) slot = &s;
        if (!biggest || s.bytes > biggest->bytes) biggest = &s;
    }
    if (slot && slot->bytes >= bytes) return slot->p;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap != hipStreamCaptureStatusNone)                       // capturing: borrow, never allocate
        return (biggest && biggest->bytes >= bytes) ? biggest->p : nullptr;
    if (!slot) {
        if (g_splitk_n < 64) slot = &g_splitk_slots[g_splitk_n++];
        else { slot = &g_splitk_slots[0]; for (int i = 1; i < 64; ++i) if (g_splitk_slots[i].bytes < slot->bytes) slot = &g_splitk_slots[i]; }   // table full: recycle the smallest
        if (slot->p) g_retired_bytes += slot->bytes;             // the recycled slot's buffer may be captured somewhere: retired, not freed
        slot->dev = d; slot->st = st; slot->p = nullptr; slot->bytes = 0;
    }
    float* fresh = nullptr;
    const size_t want = ws_round_up(bytes);
    if (hipMalloc((void**)&fresh, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }     // the old buffer (if any) stays the slot's
    if (slot->p) g_retired_bytes += slot->bytes;                 // retired: never freed (LIFETIME)
    slot->p = fresh;
    slot->bytes = want;
    return slot->p;
}
static inline int plan_split(long nwg, int nk) {
    if (nwg >= 96 || nk < 16) return 1;
    int best = 1;
    for (int s = 2; s <= 16; ++s)
        if (nk % s == 0 && nk / s >= 8 && nwg * s <= 288) best = s;
    return best;
}

template <typename OutT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, const float* __restrict__ bias,
                                                            const OutT* __restrict__ residual, const uint8_t* __restrict__ row_mask,
                                                            OutT* __restrict__ C, int M, int N, int flags)
{
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long MN = (long)M * N;
    if (i4 >= MN) return;                                       // N % 4 == 0 (checked by the launcher)
    const int ch = (int)(i4 % N);
    const long tok = i4 / N;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
        const float4 t = *reinterpret_cast<const float4*>(ws + (long)s * MN + i4);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
    }
    if (flags & EPI_BIAS) { const float4 b = *reinterpret_cast<const float4*>(bias + ch); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
    if (flags & EPI_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    if (flags & EPI_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752f));
    }
    if ((flags & EPI_ROWMASK) && row_mask[tok]) { v[0] = v[1] = v[2] = v[3] = 0.f; }
    if (flags & EPI_RESIDUAL) { float q[4]; Out<OutT>::ld4(residual + i4, q); v[0] += q[0]; v[1] += q[1]; v[2] += q[2]; v[3] += q[3]; }
    if (flags & EPI_RELU_POST) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    Out<OutT>::st4(C + i4, v);
}

// returns DTLR_OK after launching both kernels, or a negative code; `done` false = not applicable (caller runs the plain path)
template <typename T, typename OutT, bool CONV>
static int try_splitk(const void* A, const void* W, const float* bias, const void* residual, const uint8_t* row_mask, void* C,
                      int M, int N, int K, int flags, const ConvP& cp, hipStream_t st, bool& done)
{
    done = false;
    const int nN = (N + BN - 1) / BN, nM = (M + BM - 1) / BM;
    const int nk = K / GT<T>::BK;
    const int S = plan_split((long)nM * nN, nk);
    if (S <= 1 || (N & 3)) return DTLR_OK;
    float* ws = splitk_workspace((size_t)S * M * N * sizeof(float), st);
    if (!ws) return DTLR_OK;                                     // no workspace: fall back to the plain path
    const size_t lds = 4 * TILE_BYTES;
    static DevOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute((const void*)gemm_ws_kernel<T, float, false, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); (void)hipGetLastError(); }
    const unsigned grid = (unsigned)(nN * nM * S);
    hipLaunchKernelGGL((gemm_ws_kernel<T, float, false, CONV>), dim3(grid), dim3(512), lds, st,
                       (const T*)A, (const T*)nullptr, (const T*)W, (const float*)nullptr, (const float*)nullptr, (const uint8_t*)nullptr, ws,
                       M, N, K, 0, nN, nM, 1, cp, S);
    int rc = check_launch();
    if (rc != DTLR_OK) return rc;
    const long n4 = ((long)M * N) / 4;
    hipLaunchKernelGGL((splitk_reduce_kernel<OutT>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, S, bias, (const OutT*)residual, row_mask, (OutT*)C, M, N, flags);
    done = true;
    return check_launch();
}

// launch gemm_ws_kernel<T, OutT, A2, CONV, CF> (setting its dynamic-LDS attribute once)
#define WS_LAUNCH(A2, CONV, CF, GRID, ...)                                                          \
    {                                                                                              \
        static DevOnce attr_;                                                                 \
        if (attr_.first()) { (void)hipFuncSetAttribute((const void*)gemm_ws_kernel<T, OutT, A2, CONV, CF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); (void)hipGetLastError(); } \
        hipLaunchKernelGGL((gemm_ws_kernel<T, OutT, A2, CONV, CF>), dim3(GRID), dim3(512), lds, st, __VA_ARGS__); \
    }
// the flag sets that get a specialised epilogue (bf16 -> bf16 only; everything else runs the generic one)
// (round 4: also the split-fp32 kernels -- their fp32-out tiles ran the generic epilogue, ~50 VALU + 6 branches per 4-channel group)
template <typename T, typename OutT> constexpr bool kSpecialise = (sizeof(T) == 2 && sizeof(OutT) == 2) || (kSplit<T> && sizeof(OutT) == 4);

static inline bool use_tall() {
    static const int v = exp_env_int("DTLR_GEMM_TALL", 1);    // experiment builds: =0 128x128 tiles only (A/B timing)
    return v == 1;
}

// N <= 64, bf16 -> bf16, many token tiles: the 256 x 64 tile kernel.  N a multiple of 128 with K >= 512 (operand delivery from L2 is
// the limit): the 256 x 128 tile kernel.  done = false: not applicable.
static inline bool use_tall128() {
    static const int v = exp_env_int("DTLR_GEMM_TALL128", 1); // experiment builds: =0 128 x 128 tiles (A/B timing)
    return v == 1;
}
template <typename T, typename OutT, bool CONV>
static int try_tall(const void* X, const void* W, const float* bias, const void* residual, void* C,
                    int M, int N, int K, int flags, const ConvP& cp, hipStream_t st, bool& done)
{
    done = false;
    // 16-bit in / out only.  Split-fp32 operands were measured on both tall tiles in round 4 (the kernel bodies take them: the loader's
    // hi / lo conversion and mma_slab are in place): FFN linear1 720 -> 697 us, linear2 569 -> 590 us, value_proj of all decoder layers
    // (N = 1536, K = 256) 551 -> 985 us, the 256 -> 64 reductions 176 -> 173 us -- no gain, one large loss: they stay on 128 x 128 tiles.
    if constexpr (!(sizeof(T) == 2 && sizeof(OutT) == 2)) return DTLR_OK;       // (... incl. the 64-channel 3x3 convolutions on the 256 x 64 tile: 232 -> 228 us)
    else {
        if (!use_tall() || (N & 3) || (flags & ~(EPI_BIAS | EPI_RELU | EPI_RELU_POST | EPI_RESIDUAL | EPI_GELU))) return DTLR_OK;
        const int nM = (M + TALL_BM - 1) / TALL_BM;
        static const int rr = exp_env_int("DTLR_TALL_XCD", 1) == 0 ? 1 : 0;   // experiment builds: =0 round-robin placement (A/B timing)
        if (N <= 64) {
            if (M < 64 * TALL_BM) return DTLR_OK;
            if constexpr (kSplit<T>) {
                // split operands: the 64-channel 3x3 convolutions of layer1 only (at 128 x 128 tiles half of their MFMA work multiplies
                // duplicated weight rows: 232 us per launch at B = 32), specialised epilogues only
                if (!CONV || !(flags == (EPI_BIAS | EPI_RELU_POST) || flags == EPI_BIAS)) return DTLR_OK;
            }
            {
            const long target = 2 * 256 * 2;
            int per = (int)((nM + target - 1) / target);
            if (per < 1) per = 1;
            const unsigned grid = (unsigned)((nM + per - 1) / per);
            const size_t lds = 2 * TallCfg<64>::STAGE;
#define TALL_LAUNCH(CF)                                                                            \
            {                                                                                      \
                static DevOnce once;                                                               \
                if (once.first()) { (void)hipFuncSetAttribute((const void*)gemm_ws_tall_kernel<T, OutT, CONV, CF, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); (void)hipGetLastError(); } \
                hipLaunchKernelGGL((gemm_ws_tall_kernel<T, OutT, CONV, CF, 64>), dim3(grid), dim3(512), lds, st, (const T*)X, (const T*)W, bias, \
                                   (const OutT*)residual, (OutT*)C, M, N, K, flags, nM, per, cp, rr); \
            }
            if (flags == (EPI_BIAS | EPI_RELU_POST)) TALL_LAUNCH((EPI_BIAS | EPI_RELU_POST))
            else if (flags == EPI_BIAS) TALL_LAUNCH(EPI_BIAS)
            else if constexpr (!kSplit<T>) TALL_LAUNCH(-1)
#undef TALL_LAUNCH
            done = true;
            return check_launch();
            }
        }
        if constexpr (kSplit<T>) return DTLR_OK;
        else {
        // 256 x 128 tiles: enough tiles to occupy the chip (one workgroup per CU), K deep enough that operand delivery is the limit
        const int nN = N / 128;
        // measured (profile_ops, same box): the plain K >= 512 projections over >= 32768 rows gain 8-16%; the implicit-GEMM convolutions
        // and the 28800-row decoder projections lose 10-20% (one workgroup per CU: no second workgroup to overlap an epilogue with)
        if (!use_tall128() || CONV || (N & 127) || K < 512 || M < 32768 || (long)nM * nN < 192) return DTLR_OK;
        {
            const long target = 256;
            int per = (int)(((long)nM * nN + target - 1) / target);
            if (per < 1) per = 1;
            const unsigned gx = (unsigned)((nM + per - 1) / per);
            const size_t lds = 2 * TallCfg<128>::STAGE;
#define TALL_LAUNCH(CF)                                                                            \
            {                                                                                      \
                static DevOnce once;                                                               \
                if (once.first()) { (void)hipFuncSetAttribute((const void*)gemm_ws_tall_kernel<T, OutT, CONV, CF, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); (void)hipGetLastError(); } \
                hipLaunchKernelGGL((gemm_ws_tall_kernel<T, OutT, CONV, CF, 128>), dim3(gx, nN), dim3(TallCfg<128>::NT), lds, st, (const T*)X, (const T*)W, bias, \
                                   (const OutT*)residual, (OutT*)C, M, N, K, flags, nM, per, cp, rr); \
            }
            if (flags == (EPI_BIAS | EPI_RELU_POST)) TALL_LAUNCH((EPI_BIAS | EPI_RELU_POST))
            else if (flags == (EPI_BIAS | EPI_RELU_POST | EPI_RESIDUAL)) TALL_LAUNCH((EPI_BIAS | EPI_RELU_POST | EPI_RESIDUAL))
            else if (flags == EPI_BIAS) TALL_LAUNCH(EPI_BIAS)
            else if (flags == (EPI_BIAS | EPI_RELU)) TALL_LAUNCH((EPI_BIAS | EPI_RELU))
            else TALL_LAUNCH(-1)
#undef TALL_LAUNCH
            done = true;
            return check_launch();
        }
        }
    }
}

template <typename T, typename OutT>
static int launch_conv(const void* X, const void* W, const float* bias, const void* residual, void* C,
                       int M, int N, int K, int flags, const ConvP& cp, hipStream_t st)
{
    {
        bool done = false;
        const int rc = try_tall<T, OutT, true>(X, W, bias, residual, C, M, N, K, flags, cp, st, done);
        if (rc != DTLR_OK || done) return rc;
    }
    const int nM = (M + BM - 1) / BM, nN = (N + BN - 1) / BN;
    const long nwg = (long)nM * nN;
    if (nwg > 0x7fffffffL) return DTLR_ESHAPE;
    const size_t lds = 4 * TILE_BYTES;
    if constexpr (!kSplit<T>) {
        static DevOnce attr;
        if (attr.first()) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<T, OutT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
    }
    if (use_ws()) {
        bool done = false;
        const int rc = try_splitk<T, OutT, true>(X, W, bias, residual, nullptr, C, M, N, K, flags, cp, st, done);
        if (rc != DTLR_OK || done) return rc;
        const int per = plan_chain_ws(nwg);
        const unsigned grid = (unsigned)(nN * ((nM + per - 1) / per));
#define CONV_ARGS (const T*)X, (const T*)nullptr, (const T*)W, bias, (const OutT*)residual, (const uint8_t*)nullptr, (OutT*)C, M, N, K, flags, nN, nM, per, cp, 1
        if constexpr (kSpecialise<T, OutT>) {
            if (flags == (EPI_BIAS | EPI_RELU_POST)) { WS_LAUNCH(false, true, (EPI_BIAS | EPI_RELU_POST), grid, CONV_ARGS) return check_launch(); }
            if (flags == (EPI_BIAS | EPI_RELU_POST | EPI_RESIDUAL)) { WS_LAUNCH(false, true, (EPI_BIAS | EPI_RELU_POST | EPI_RESIDUAL), grid, CONV_ARGS) return check_launch(); }
            if (flags == EPI_BIAS) { WS_LAUNCH(false, true, EPI_BIAS, grid, CONV_ARGS) return check_launch(); }
        }
        WS_LAUNCH(false, true, -1, grid, CONV_ARGS)
#undef CONV_ARGS
        return check_launch();
    }
    if constexpr (kSplit<T>) return DTLR_ESHAPE;                 // split operands: the wave-specialised kernel only
    else {
    const int per = plan_chain(nwg);
    const unsigned grid = (unsigned)(nN * ((nM + per - 1) / per));
    hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, false, true>), dim3(grid), dim3(256), lds, st,
                       (const T*)X, (const T*)nullptr, (const T*)W, bias, (const OutT*)residual, (const uint8_t*)nullptr, (OutT*)C,
                       M, N, K, flags, nN, nM, per, cp);
    return check_launch();
    }
}

template <typename T, typename OutT>
static int launch_gemm(const void* A, const void* A2, const void* W, const float* bias, const void* residual,
                       const uint8_t* row_mask, void* C, int M, int N, int K, int flags, hipStream_t st, int a2_rows = 0, int lda = 0,
                       int res_rows = 0)
{
    ConvP cp{};
    cp.a2_rows = a2_rows;
    cp.lda = lda;
    cp.res_rows = res_rows;
    if (!A2 && !row_mask && !res_rows) {
        bool done = false;
        const int rc = try_tall<T, OutT, false>(A, W, bias, residual, C, M, N, K, flags, cp, st, done);
        if (rc != DTLR_OK || done) return rc;
    }
    const int nN = (N + BN - 1) / BN, nM = (M + BM - 1) / BM;
    const long nwg = (long)nM * nN;
    if (nwg > 0x7fffffffL) return DTLR_ESHAPE;
    const size_t lds = 4 * TILE_BYTES;
    if (use_ws()) {
        if (!A2 && !(flags & EPI_ROWMAX) && !res_rows) {
            bool done = false;
            const int rc = try_splitk<T, OutT, false>(A, W, bias, residual, row_mask, C, M, N, K, flags, cp, st, done);
            if (rc != DTLR_OK || done) return rc;
        }
        const int perw = plan_chain_ws(nwg);
        const unsigned gridw = (unsigned)(nN * ((nM + perw - 1) / perw));
#define GEMM_ARGS(A2P) (const T*)A, (const T*)(A2P), (const T*)W, bias, (const OutT*)residual, row_mask, (OutT*)C, M, N, K, flags, nN, nM, perw, cp, 1
        if (A2) {
            if constexpr (kSpecialise<T, OutT>) {
                if (flags == EPI_BIAS) { WS_LAUNCH(true, false, EPI_BIAS, gridw, GEMM_ARGS(A2)) return check_launch(); }
            }
            WS_LAUNCH(true, false, -1, gridw, GEMM_ARGS(A2))
        } else {
            if constexpr (kSpecialise<T, OutT>) {
                if (flags == EPI_BIAS) { WS_LAUNCH(false, false, EPI_BIAS, gridw, GEMM_ARGS(nullptr)) return check_launch(); }
                if (flags == (EPI_BIAS | EPI_RELU)) { WS_LAUNCH(false, false, (EPI_BIAS | EPI_RELU), gridw, GEMM_ARGS(nullptr)) return check_launch(); }
                if (flags == (EPI_BIAS | EPI_RELU_POST)) { WS_LAUNCH(false, false, (EPI_BIAS | EPI_RELU_POST), gridw, GEMM_ARGS(nullptr)) return check_launch(); }
                if (flags == (EPI_BIAS | EPI_RELU_POST | EPI_RESIDUAL)) { WS_LAUNCH(false, false, (EPI_BIAS | EPI_RELU_POST | EPI_RESIDUAL), gridw, GEMM_ARGS(nullptr)) return check_launch(); }
                if constexpr (kSplit<T>) {
                    if (flags == EPI_RESIDUAL) { WS_LAUNCH(false, false, EPI_RESIDUAL, gridw, GEMM_ARGS(nullptr)) return check_launch(); }
                }
            }
            WS_LAUNCH(false, false, -1, gridw, GEMM_ARGS(nullptr))
        }
#undef GEMM_ARGS
        return check_launch();
    }
    if constexpr (kSplit<T>) return DTLR_ESHAPE;
    else {
    const int per = plan_chain(nwg);
    const unsigned grid = (unsigned)(nN * ((nM + per - 1) / per));
    if (A2) {
        static DevOnce attr_a2;
        if (attr_a2.first()) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<T, OutT, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
        hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, true, false>), dim3(grid), dim3(256), lds, st,
                           (const T*)A, (const T*)A2, (const T*)W, bias, (const OutT*)residual, row_mask, (OutT*)C, M, N, K, flags, nN, nM, per, cp);
    } else {
        static DevOnce attr;
        if (attr.first()) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<T, OutT, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
        hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, false, false>), dim3(grid), dim3(256), lds, st,
                           (const T*)A, (const T*)nullptr, (const T*)W, bias, (const OutT*)residual, row_mask, (OutT*)C, M, N, K, flags, nN, nM, per, cp);
    }
    return check_launch();
    }
}

// fp32 weight [rows, K] -> the split slab image (same bytes): per 32-k slab of a row, chunk c < 4 = fp16 hi of k 8c..8c+7, chunk 4 + c = lo
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ w, unsigned char* __restrict__ out, long n4, int K)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;          // one thread per 4 consecutive k
    if (i >= n4) return;
    const int k4 = K >> 2;
    const long row = i / k4;
    const int q = (int)(i % k4), slab = q >> 3, kc = q & 7;
    uint2 hi, lo;
    GT<f32s_t>::split4(*reinterpret_cast<const uint4*>(w + i * 4), hi, lo);
    unsigned char* dst = out + (row * K + slab * 32) * 4 + (kc >> 1) * 16 + (kc & 1) * 8;
    *reinterpret_cast<uint2*>(dst) = hi;
    *reinterpret_cast<uint2*>(dst + 64) = lo;
}

}  // namespace dtlr

using namespace dtlr;

// Pre-size the calling stream's scratch workspace (split-K partial tiles, hidden-split FFN parts) OUTSIDE stream capture, so that
// captured and eager launches of one shape take the same kernels and no allocation is attempted while capturing.  Returns DTLR_OK,
// DTLR_EINVAL (bytes <= 0, or the stream is capturing) or DTLR_ELAUNCH (out of memory).
extern "C" int dtlr_workspace_reserve(long bytes, void* stream)
{
    clear_stale_error();
    if (bytes <= 0) return DTLR_EINVAL;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap != hipStreamCaptureStatusNone) return DTLR_EINVAL;
    return stream_workspace((size_t)bytes, (hipStream_t)stream) ? DTLR_OK : DTLR_ELAUNCH;
}
// bytes of workspace buffers that were replaced by larger ones and are kept allocated because a captured graph may still hold them
extern "C" long dtlr_workspace_retired_bytes(void)
{
    std::lock_guard<std::mutex> lk(g_splitk_mu);
    return (long)g_retired_bytes;
}

extern "C" int dtlr_gemm_nt(const void* A, const void* A2, const void* W, const float* bias,
                            const void* residual, const unsigned char* row_mask, void* C,
                            int M, int N, int K, int relu, int in_dtype, int out_dtype, void* stream)
{
    clear_stale_error();
    if (!A || !W || !C) return DTLR_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0) return DTLR_EINVAL;
    int flags = (bias ? EPI_BIAS : 0) | (relu == 1 ? EPI_RELU : 0) | (relu == 2 ? EPI_RELU_POST : 0) | (relu == 3 ? EPI_GELU : 0) |
                (residual ? EPI_RESIDUAL : 0) | (row_mask ? EPI_ROWMASK : 0);
#ifdef DTLR_GEMM_ABLATION
    flags |= (exp_env_int("DTLR_GEMM_ABLATE", 0) & (DBG_NO_LOAD | DBG_NO_MMA | DBG_NO_LDS | DBG_NO_EPI | DBG_NO_STORE));   // timing experiments only
#endif
    hipStream_t st = (hipStream_t)stream;
    if (in_dtype == DTLR_H16) {
        if (K % 64) return DTLR_ESHAPE;
        if (out_dtype == DTLR_H16) return launch_gemm<uint16_t, uint16_t>(A, A2, W, bias, residual, row_mask, C, M, N, K, flags, st);
        if (out_dtype == DTLR_F32) return launch_gemm<uint16_t, float>(A, A2, W, bias, residual, row_mask, C, M, N, K, flags, st);
        return DTLR_EDTYPE;
    }
    if (in_dtype == DTLR_F32) {
        if (K % 32) return DTLR_ESHAPE;
        if (out_dtype == DTLR_F32) return launch_gemm<float, float>(A, A2, W, bias, residual, row_mask, C, M, N, K, flags, st);
        return DTLR_EDTYPE;
    }
    if (in_dtype == DTLR_F32S) {                                 // fp32 activations, W = dtlr_split_pack_weights image, fp32 result
        if (K % 32) return DTLR_ESHAPE;
        if (out_dtype == DTLR_F32) return launch_gemm<f32s_t, float>(A, A2, W, bias, residual, row_mask, C, M, N, K, flags, st);
        return DTLR_EDTYPE;
    }
    return DTLR_EDTYPE;
}

extern "C" int dtlr_split_pack_weights(const float* w, void* out, long rows, int K, void* stream)
{
    clear_stale_error();
    if (!w || !out || rows <= 0 || K <= 0) return DTLR_EINVAL;
    if (K % 32) return DTLR_ESHAPE;
    const long n4 = rows * (K / 4);
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)out, n4, K);
    return check_launch();
}

// dtlr_gemm_nt with a row-BROADCAST A2: A2 has a2_rows rows and row m of A is paired with row m % a2_rows (the encoder's
// position embedding of an unpadded batch is the same [S, 256] matrix for every image: 2.8 MB that stays in L2 instead of a
// second [B*S, 256] operand streamed from HBM).  bf16 in / bf16 out.
extern "C" int dtlr_gemm_nt_a2bcast(const void* A, const void* A2, int a2_rows, const void* W, const float* bias, void* C,
                                    int M, int N, int K, int dtype, void* stream)
{
    clear_stale_error();
    if (!A || !A2 || !W || !C) return DTLR_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || a2_rows <= 0 || M % a2_rows) return DTLR_EINVAL;
    const int flags = bias ? EPI_BIAS : 0;
    if (dtype == DTLR_H16) {
        if (K % 64) return DTLR_ESHAPE;
        return launch_gemm<uint16_t, uint16_t>(A, A2, W, bias, nullptr, nullptr, C, M, N, K, flags, (hipStream_t)stream, a2_rows);
    }
    if (dtype == DTLR_F32) {
        if (K % 32) return DTLR_ESHAPE;
        return launch_gemm<float, float>(A, A2, W, bias, nullptr, nullptr, C, M, N, K, flags, (hipStream_t)stream, a2_rows);
    }
    if (dtype == DTLR_F32S) {
        if (K % 32) return DTLR_ESHAPE;
        return launch_gemm<f32s_t, float>(A, A2, W, bias, nullptr, nullptr, C, M, N, K, flags, (hipStream_t)stream, a2_rows);
    }
    return DTLR_EDTYPE;
}

// C = A . W^T [+ bias] + residual[m % res_rows]: dtlr_gemm_nt with a row-BROADCAST residual (res_rows rows, M % res_rows == 0).
extern "C" int dtlr_gemm_nt_resbcast(const void* A, const void* W, const float* bias, const void* residual, int res_rows, void* C,
                                     int M, int N, int K, int dtype, void* stream)
{
    clear_stale_error();
    if (!A || !W || !residual || !C) return DTLR_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || res_rows <= 0 || M % res_rows) return DTLR_EINVAL;
    const int flags = (bias ? EPI_BIAS : 0) | EPI_RESIDUAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DTLR_H16) {
        if (K % 64) return DTLR_ESHAPE;
        return launch_gemm<uint16_t, uint16_t>(A, nullptr, W, bias, residual, nullptr, C, M, N, K, flags, st, 0, 0, res_rows);
    }
    if (K % 32) return DTLR_ESHAPE;
    if (dtype == DTLR_F32) return launch_gemm<float, float>(A, nullptr, W, bias, residual, nullptr, C, M, N, K, flags, st, 0, 0, res_rows);
    if (dtype == DTLR_F32S) return launch_gemm<f32s_t, float>(A, nullptr, W, bias, residual, nullptr, C, M, N, K, flags, st, 0, 0, res_rows);
    return DTLR_EDTYPE;
}

// scores[m] = max_n ( A[m,:] . W[n,:] + bias[n] ): the GEMM above with the row-max epilogue; the [M, N] product never reaches HBM.
extern "C" int dtlr_gemm_nt_rowmax(const void* A, const void* W, const float* bias, float* rowmax,
                                   int M, int N, int K, int in_dtype, void* stream)
{
    return dtlr_gemm_nt_rowmax_lda(A, K, W, bias, rowmax, M, N, K, in_dtype, stream);
}

// -inf initialisation of a row-max vector as an ORDINARY kernel (round 5).  It was hipMemsetD32Async: under stream capture that becomes a
// memset node, and a replay of the captured forward after any eager forward of the fp32 engine filled the vector with ZEROS instead of
// 0xff800000 (tools/experiments/graph_replay_probe2.py: every stage identical up to `memory`, two-stage scores off by up to 6.0 = the
// class bias, i.e. atomicMin against 0 instead of -inf; the eager path never differed) -- the runtime's fill pattern is not part of the
// captured node's own state.  A kernel node carries its value argument by value.
__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t* __restrict__ p, uint32_t v, long n)
{
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n && (((size_t)p & 15) == 0)) *reinterpret_cast<uint4*>(p + i) = make_uint4(v, v, v, v);
    else for (long k = i; k < n && k < i + 4; ++k) p[k] = v;
}

extern "C" int dtlr_gemm_nt_rowmax_lda(const void* A, int lda, const void* W, const float* bias, float* rowmax,
                                       int M, int N, int K, int in_dtype, void* stream)
{
    clear_stale_error();
    if (!A || !W || !rowmax) return DTLR_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || lda < K) return DTLR_EINVAL;
    if ((lda * (in_dtype == DTLR_F32 || in_dtype == DTLR_F32S ? 4 : 2)) & 15) return DTLR_ESHAPE;          // rows must stay 16-byte aligned
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)(((long)M + 1023) / 1024)), dim3(256), 0, st, reinterpret_cast<uint32_t*>(rowmax), 0xff800000u, (long)M);
    { const int rc_ = check_launch(); if (rc_ != DTLR_OK) return rc_; }
    const int flags = (bias ? EPI_BIAS : 0) | EPI_ROWMAX;
    if (in_dtype == DTLR_H16) {
        if (K % 64) return DTLR_ESHAPE;
        return launch_gemm<uint16_t, float>(A, nullptr, W, bias, nullptr, nullptr, rowmax, M, N, K, flags, st, 0, lda == K ? 0 : lda);
    }
    if (in_dtype == DTLR_F32) {
        if (K % 32) return DTLR_ESHAPE;
        return launch_gemm<float, float>(A, nullptr, W, bias, nullptr, nullptr, rowmax, M, N, K, flags, st, 0, lda == K ? 0 : lda);
    }
    if (in_dtype == DTLR_F32S) {
        if (K % 32) return DTLR_ESHAPE;
        return launch_gemm<f32s_t, float>(A, nullptr, W, bias, nullptr, nullptr, rowmax, M, N, K, flags, st, 0, lda == K ? 0 : lda);
    }
    return DTLR_EDTYPE;
}

#ifdef DTLR_GEMM_TRACE
// instrumentation builds only (tools/): read and reset the timeline of gemm_ws_kernel: out[TL_BLOCKS][2][TL_EVENTS]
extern "C" int dtlr_debug_gemm_trace(unsigned long long* out, int clear_only)
{
    if (hipDeviceSynchronize() != hipSuccess) return DTLR_ELAUNCH;
    const size_t bytes = sizeof(unsigned long long) * TL_BLOCKS * 2 * TL_EVENTS;
    if (!clear_only && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_tl), bytes) != hipSuccess) return DTLR_ELAUNCH;
    void* dptr = nullptr;
    if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(g_gemm_tl)) != hipSuccess) return DTLR_ELAUNCH;
    if (hipMemset(dptr, 0, bytes) != hipSuccess) return DTLR_ELAUNCH;
    return DTLR_OK;
}
#endif

extern "C" int dtlr_conv3x3_patch_supported(int Cin, int Cout);
extern "C" int dtlr_conv3x3_patch_f32s_supported(int Cin, int Cout);
extern "C" int dtlr_conv3x3_patch_f32s(const float* X, const void* Wt, const float* bias, float* Y, int B, int H, int W, int Cin, int Cout,
                                       int relu, void* stream);
extern "C" int dtlr_conv3x3_patch_bf16(const void* X, const void* Wt, const float* bias, void* Y, int B, int H, int W, int Cin, int Cout,
                                       int relu, void* stream);

extern "C" int dtlr_conv2d_nhwc(const void* X, const void* W, const float* bias, const void* residual, void* Y,
                                int B, int H, int Wd, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                int relu, int dtype, void* stream)
{
    clear_stale_error();
    if (!X || !W || !Y) return DTLR_EINVAL;
    if (B <= 0 || H <= 0 || Wd <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return DTLR_EINVAL;
    const int elem = dtype == DTLR_H16 ? 2 : (dtype == DTLR_F32 || dtype == DTLR_F32S) ? 4 : 0;
    if (!elem) return DTLR_EDTYPE;
    if ((Cin * elem) % SLAB) return DTLR_ESHAPE;                 // a K slab must stay inside one tap
    ConvP cp{};                                                  // a2_rows = lda = res_rows = 0: the epilogue indexes the residual by row m itself
    cp.H = H; cp.W = Wd; cp.Cin = Cin; cp.KH = KH; cp.KW = KW; cp.stride = stride; cp.pad = pad;
    cp.Ho = (H + 2 * pad - KH) / stride + 1;
    cp.Wo = (Wd + 2 * pad - KW) / stride + 1;
    if (cp.Ho <= 0 || cp.Wo <= 0) return DTLR_ESHAPE;
    const long M = (long)B * cp.Ho * cp.Wo;
    if (M > 0x7fffffffL) return DTLR_ESHAPE;
    const int K = KH * KW * Cin;
    int flags = (bias ? EPI_BIAS : 0) | (relu == 1 ? EPI_RELU : 0) | (relu == 2 ? EPI_RELU_POST : 0) | (residual ? EPI_RESIDUAL : 0);
    hipStream_t st = (hipStream_t)stream;
    // 3x3 / stride 1 / pad 1 without residual on enough pixels to fill the chip: the kernel that keeps the input patch in LDS (conv3x3.hip)
    static const bool use_patch = exp_env_int("DTLR_CONV_PATCH", 1) != 0;      // experiment builds: =0 implicit GEMM (A/B timing)
    if (use_patch && dtype == DTLR_H16 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && !residual && M >= 16384
        && dtlr_conv3x3_patch_supported(Cin, Cout) == 1)
        return dtlr_conv3x3_patch_bf16(X, W, bias, Y, B, H, Wd, Cin, Cout, relu ? 1 : 0, stream);
    // ... and its split-fp32 form (round 6): the fp32 patch is split into fp16 hi + lo planes once per workgroup instead of once per tap
    if (use_patch && dtype == DTLR_F32S && KH == 3 && KW == 3 && stride == 1 && pad == 1 && !residual && M >= 16384
        && dtlr_conv3x3_patch_f32s_supported(Cin, Cout) == 1)
        return dtlr_conv3x3_patch_f32s((const float*)X, W, bias, (float*)Y, B, H, Wd, Cin, Cout, relu ? 1 : 0, stream);
    if (dtype == DTLR_H16) return launch_conv<uint16_t, uint16_t>(X, W, bias, residual, Y, (int)M, Cout, K, flags, cp, st);
    if (dtype == DTLR_F32S) return launch_conv<f32s_t, float>(X, W, bias, residual, Y, (int)M, Cout, K, flags, cp, st);
    return launch_conv<float, float>(X, W, bias, residual, Y, (int)M, Cout, K, flags, cp, st);
}
