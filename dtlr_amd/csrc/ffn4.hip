// Fused position-wise feed-forward block, fourth structure (round 6): PERSISTENT workgroups whose waves each carry TWO token tiles a few
// steps apart over ONE cyclic weight stream, so that a tile's epilogue and its successor's load never coincide with the other tile's.
//
//     Y = LayerNorm( X + relu(X W1^T + b1) W2^T + b2 )            X, Y: [M, 256]   W1: [d_ff, 256]   W2: [256, d_ff]
// == forward_ffn + norm2 of the encoder layer (models/dino/deformable_transformer.py:804-823).
//
// Why.  ffn32.hip's ffn3_bf16_kernel (0.52-0.53 of the MFMA peak, matrix pipe busy 68 % of the busy CU time) runs one wave per SIMD with two
// 32-token column tiles per wave, one 256-token tile per workgroup.  Nothing overlaps a workgroup's prologue (the X tile: every CU fetching
// 128 KB at the same moment, ~6 us; residual seeding) and epilogue (LayerNorm + stores, ~4 us): 17 us of each 104 us workgroup-round; and the
// encoder call's 680 tiles are 2.66 rounds of 256 workgroups, so its last partial round ran on another kernel at 0.44 of the peak.
// Here the SAME chunk step (same fragment images: dtlr_ffn32_pack_weights, same register dataflow, same rings) runs in a persistent workgroup:
//   * the weights stream CYCLICALLY through the two 4-stage LDS rings (chunk s mod d_ff/32 at stream step s) for the whole life of the
//     workgroup.  The sum over hidden chunks is order-free, so a tile may START at any chunk;
//   * a wave's two 32-token column tiles are two independent SLOTS.  Slot 0 of the four waves = one 128-token tile, slot 1 another, started
//     `lag` steps apart (a kernel argument, 2).  In the steady state both slots multiply against every weight fragment the wave reads (ffn3's
//     step, unchanged: one ds_read_b128 feeds two MFMAs); in a slot's OWN step the other slot runs a single-slot chunk step first;
//   * a slot's period = d_ff/32 + 1 chunk steps + 1 own step: row sum | centred squares | rows of the NEXT tile start to load | scale, round,
//     store (channel tiles 0..3, then 4..7) | wait for the rows | accumulators = X through the matrix pipe, as in ffn3.  Three READ-ONLY passes
//     over the accumulators (the slot's X^T registers receive the next tile meanwhile).  A slot's first chunk step multiplies phase B by zero
//     H fragments and its last one runs a redundant phase A: 32 wasted MFMAs per tile (1.5 %) for one step body instead of three;
//   * tiles are dealt in PAIRS: workgroup b, iteration i takes tiles 2 (b + i G) and 2 (b + i G) + 1 (G = min(#CU, pairs)) for slot 0 and
//     slot 1 (a missing last tile: clamped rows, no stores).  One kernel for any M >= 1: no tail kernel.
//
// vmcnt bookkeeping.  The counter is in order and shared by the weight DMA (8 pieces per wave and step), the X loads and the Y stores.  The
// pieces of step s - 2 must have landed at the top of step s; what may stay in flight there is everything issued after them:
// post(s - 2) + tot(s - 1), tot = all vector-memory operations of a step, post = those after its pieces (a slot's own step: 16 X loads and
// 16 stores -- counts that must be EXACT: a wave with no row below M issues no stores and counts none).  The values that occur (8, 24, 40)
// are tracked in three scalars and waited for by a three-way uniform branch.
//
// STATUS (round 6, measured on MI355X, tools/experiments/ffn4_scaling.py, profiles/r06_ffn4_*.txt): CORRECT (tests/test_gpu_kernels.py::
// test_ffn4_vs_reference_and_ffn32) and NOT ADOPTED by the engine -- the encoder call (174,080 rows) takes 350-388 us here against 305 us for
// dtlr_ffn32_bf16 over all rows and 337 us for the engine's ffn32 + tail pair on the same boxes.  Where the time goes, per tile-pair period
// (256 rows per compute unit; ffn3's whole workgroup-round: 102.5 us):
//     64 two-slot chunk steps + 2 single-slot ones, nothing else      93.5 us   (1.42 us per step: ffn3's own step, 1.37)
//     + rows of the next tile loaded and seeded in the slot's own step  +22 us   (the load is issued and waited for inside ONE step of ONE wave:
//                                                                               its latency -- every workgroup at the same stream step, so
//                                                                               a 16 MB burst -- stalls BOTH slots of the wave)
//     + LayerNorm / store slices (three read passes over the AGPRs)     +15 us   (7.5 us per 128-row tile; ffn3: 4.2 us per 256 rows with
//                                                                               the dead X^T registers as a scratch copy)
// i.e. the two things the design set out to hide cost more than in ffn3 and are NOT hidden: the other slot lives in the same wave.  What
// would make it pay: the loads one step earlier (the slot's last chunk step without its redundant phase A) and the three passes cut into the
// other slot's MFMA groups like the H epilogue; both need more step bodies, and every extra body in this kernel has cost a fight with the
// register allocator (see the notes below).  The slot lag is a kernel argument (2 steps; half a period measured 10 us slower).
#include "dtlr_common.h"

namespace dtlr {

typedef __attribute__((ext_vector_type(8))) h16_hw_t f4_h16x8_t;
typedef __attribute__((ext_vector_type(16))) float f4_f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned f4_u32x4_t;
typedef __attribute__((ext_vector_type(2))) float f4_f32x2_t;
typedef __attribute__((ext_vector_type(4))) float f4_f32x4_t;
typedef __attribute__((address_space(3))) const f4_f32x4_t* f4_lds4_t;
typedef __attribute__((address_space(3))) const f4_f32x2_t* f4_lds2_t;

constexpr int F4_NS = 4;                                    // ring stages
constexpr int F4_RING = 16384;                              // one W1 (or W2) chunk image: 16 fragments of 1 KB
constexpr int F4_W2_OFF = F4_NS * F4_RING;
constexpr int F4_B1_OFF = 2 * F4_NS * F4_RING;
constexpr int F4_MAX_DFF = 2048;
constexpr int F4_PRM_OFF = F4_B1_OFF + (F4_MAX_DFF + 32) * 4;      // b2 | gamma | beta (3 x 256 floats)
constexpr int F4_LDS = F4_PRM_OFF + 3 * 256 * 4;
constexpr int F4_NW = 2;                                    // weight-fragment registers: refilled F4_NW MFMA groups ahead of use

// LDS-DMA with a wave-uniform base in SGPRs, a 32-bit per-lane byte offset and an immediate that applies to both addresses
template <int OFF> __device__ __forceinline__ void f4_glds16so(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst), "n"(OFF) : "memory");
}
// Every instruction that writes one of the long-lived register arrays (Y^T accumulators, H^T accumulators, X^T fragments) is inline asm with a
// TIED read-write operand: each array element is then ONE live range for the whole kernel, which the register allocator places once.  (As
// builtins / plain assignments the seeding step and the loads DEFINE new values that meet the old ones at the loop header; with every register
// of the file in use hipcc resolved those joins with copies and spills -- 1000 spilled registers, the X^T fragments reloaded from scratch in
// every step.)  The price: the hazard recogniser does not see these MFMAs.  The uses are arranged so that no software wait states are owed:
// an accumulator is re-used as SrcC no earlier than two MFMAs later (8 in phase B), its first VALU reader comes at least two issued MFMAs (H^T)
// or a whole step (Y^T) later, and MFMA A / B operands come from ds_read / long-lived registers (waited for by the compiler: asm inputs).
__device__ __forceinline__ void f4_mma_a(const uint4& a, const uint4& b, f4_f32x16_t& c) {         // c += a b, accumulator in the AGPRs
    const f4_u32x4_t av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    asm volatile("v_mfma_f32_32x32x16_" DTLR_H16_ASM_SUFFIX " %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
}
__device__ __forceinline__ void f4_mma_a0(const uint4& a, const uint4& b, f4_f32x16_t& c) {        // c = a b
    const f4_u32x4_t av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    asm volatile("v_mfma_f32_32x32x16_" DTLR_H16_ASM_SUFFIX " %0, %1, %2, 0" : "+a"(c) : "v"(av), "v"(bv));
}
// Elements 4 q .. 4 q + 3 of an accumulator tile, read through asm with an "a" operand: the element extraction then stays an AGPR
// sub-register.  (Plain C++ element reads put the tile's virtual register into a VALU instruction, i.e. into the VGPR class: hipcc copied
// all eight 16-register tiles of a slot to VGPRs at once -- 128 registers it does not have -- spilled twelve X^T fragments around every
// epilogue step and left a vmcnt(0) wait for their reload inside the chunk loop that follows.)
__device__ __forceinline__ f4_f32x4_t f4_acc_read4(const f4_f32x16_t& c, int q) {
    float x0, x1, x2, x3;
    const float s0 = c[4 * q], s1 = c[4 * q + 1], s2 = c[4 * q + 2], s3 = c[4 * q + 3];
    asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "a"(s0), "a"(s1), "a"(s2), "a"(s3));
    return f4_f32x4_t{x0, x1, x2, x3};
}
// the same two for the seeding step, whose operands (identity fragments built by VALU instructions, X^T fragments the compiler may have just
// re-assembled with v_mov) can be VALU results: a VALU write of an MFMA source register needs wait states in front of the MFMA, which
// nobody inserts for asm (first build: `v_or_b32 v167, ...` directly in front of `v_mfma ..., v[164:167], ...` -- the first tile of slot 0
// came out as garbage)
__device__ __forceinline__ void f4_mma_a_pad(const uint4& a, const uint4& b, f4_f32x16_t& c) {
    const f4_u32x4_t av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_" DTLR_H16_ASM_SUFFIX " %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
}
__device__ __forceinline__ void f4_mma_a0_pad(const uint4& a, const uint4& b, f4_f32x16_t& c) {
    const f4_u32x4_t av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_" DTLR_H16_ASM_SUFFIX " %0, %1, %2, 0" : "+a"(c) : "v"(av), "v"(bv));
}
__device__ __forceinline__ void f4_mma_v0(const uint4& a, const uint4& b, f4_f32x16_t& c) {        // the same in ARCHITECTURAL VGPRs (H^T: read by the VALU)
    const f4_u32x4_t av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    asm volatile("v_mfma_f32_32x32x16_" DTLR_H16_ASM_SUFFIX " %0, %1, %2, 0" : "=&v"(c) : "v"(av), "v"(bv));     // a true definition: H^T is dead between a step's last slice and this
}
__device__ __forceinline__ void f4_mma_v(const uint4& a, const uint4& b, f4_f32x16_t& c) {
    const f4_u32x4_t av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    asm volatile("v_mfma_f32_32x32x16_" DTLR_H16_ASM_SUFFIX " %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
}
// LDS byte address of this lane's slice of the parameter tables, made opaque in every epilogue slice: computed outside, hipcc hoists one
// address register per table row out of the tile loop (~100 registers) and spills them
__device__ __forceinline__ unsigned f4_prm_addr(unsigned lds_base, int hh) {
    unsigned a = lds_base + (unsigned)F4_PRM_OFF + 16u * (unsigned)hh;
    asm volatile("" : "+v"(a));
    return a;
}
__device__ __forceinline__ float f4_relu(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_huge_valf()); }

__global__ __launch_bounds__(256, 1) void ffn4_bf16_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ W1p, const float* __restrict__ b1,
    const uint16_t* __restrict__ W2p, const float* __restrict__ b2, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, uint16_t* __restrict__ Y, int M, int d_ff, int lag)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char f4_smem[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)f4_smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31, hh = lane >> 5;
    const int nchunk = d_ff >> 5;
    const int NT = (M + 127) >> 7, NP = (NT + 1) >> 1, G = (int)gridDim.x, wg = (int)blockIdx.x;
    if (wg >= NP) return;
    const int np = (NP - wg + G - 1) / G;                       // tile PAIRS of this workgroup: pair i = tiles 2 (wg + i G) + {0, 1} -> slot 0, slot 1
    // a slot's period: nchunk + 1 chunk steps + 1 epilogue / seeding step.  Slot 1 runs `lag` steps behind slot 0: enough for the two slots'
    // epilogue steps never to coincide; every step of lag is a step with half the matrix pipe idle at either end of the workgroup's life
    // (first build: half a period, 33 steps -- 14 % of the encoder call's 232 steps -- and no faster than the kernels it replaces).  A kernel
    // ARGUMENT, not a constant: with a constant trip count hipcc unrolls the two single-slot loops into their neighbours and spills.
    const int PER = nchunk + 2, HALF = min(max(lag & 255, 1), PER - 1);
    const bool dbg_noepi = (lag >> 8) & 1, dbg_noinit = (lag >> 9) & 1;          // timing experiments only (results are garbage)

    // ---- weight DMA: wave w moves fragments 4 w .. 4 w + 3 of every chunk image ----------------------------------------------------
    const unsigned vlane = (unsigned)lane * 16u;
    const char* W1b = reinterpret_cast<const char*>(W1p) + wave * 4096;
    const char* W2b = reinterpret_cast<const char*>(W2p) + wave * 4096;
    const unsigned my1 = lds_base + (unsigned)wave * 4096u, my2 = my1 + F4_W2_OFF;
#define F4_PIECE1(CI, ST, U) f4_glds16so<(U) * 1024>(W1b + (long)(CI) * F4_RING, vlane, my1 + (unsigned)(ST) * F4_RING);
#define F4_PIECE2(CI, ST, U) f4_glds16so<(U) * 1024>(W2b + (long)(CI) * F4_RING, vlane, my2 + (unsigned)(ST) * F4_RING);
#define F4_ISSUE1(CI, ST) { F4_PIECE1(CI, ST, 0) F4_PIECE1(CI, ST, 1) F4_PIECE1(CI, ST, 2) F4_PIECE1(CI, ST, 3) }
#define F4_ISSUE2(CI, ST) { F4_PIECE2(CI, ST, 0) F4_PIECE2(CI, ST, 1) F4_PIECE2(CI, ST, 2) F4_PIECE2(CI, ST, 3) }
    // prologue: W1(0) W1(1) | W2(0) W2(1) W1(2); step s then issues W2(s + 2), W1(s + 3)   (indices = stream steps; chunk = step mod nchunk;
    // W1(i) is multiplied in step i, W2(i) in step i + 1)
    {
        const int c1 = 1 % nchunk, c2 = 2 % nchunk;
        F4_ISSUE1(0, 0)
        F4_ISSUE1(c1, 1)
        F4_ISSUE2(0, 0)
        F4_ISSUE2(c1, 1)
        F4_ISSUE1(c2, 2)
    }

    // X^T B-fragments: lane (j, hh) of slot tt holds X[row][16 s + 8 hh .. +7], row = tile * 128 + 32 wave + j; tile (tt, n) = position 2 n + tt
    uint4 xf[16][2];
#define F4_TOK(TT, I) ((long)(2 * (wg + (I) * G) + (TT)) * 128 + wave * 32 + j)
    // (compiler-visible loads: hipcc waits for them itself in front of their first use, the seeding MFMAs of the same straight-line step --
    // by then only the step's stores are younger.  Inside a loop with several step bodies that wait landed in front of every body's first
    // use of xf: a drain of the weight DMA per step.)
#define F4_LOADX(TT, I)                                                                            \
    {                                                                                              \
        const long tok_ = min(F4_TOK(TT, I), (long)M - 1);                                         \
        const uint4* src_ = reinterpret_cast<const uint4*>(X + tok_ * 256 + hh * 8);               \
        _Pragma("unroll") for (int s_ = 0; s_ < 16; ++s_) xf[s_][TT] = src_[2 * s_];              \
    }
    {   // b1 table (zero padded), epilogue parameters
        float* b1s = reinterpret_cast<float*>(f4_smem + F4_B1_OFF);
        for (int i = (int)threadIdx.x * 4; i < d_ff + 32; i += 256 * 4)
            *reinterpret_cast<float4*>(b1s + i) = i < d_ff ? *reinterpret_cast<const float4*>(b1 + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        float* prm = reinterpret_cast<float*>(f4_smem + F4_PRM_OFF);
        prm[threadIdx.x] = b2[threadIdx.x];
        prm[256 + threadIdx.x] = gamma[threadIdx.x];
        prm[512 + threadIdx.x] = beta[threadIdx.x];
    }
    f4_f32x16_t yacc[8][2];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int i = 0; i < 16; ++i) yacc[ct][tt][i] = 0.f;
    f4_f32x16_t he[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int i = 0; i < 16; ++i) he[tt][i] = 0.f;
    uint4 hb[2][2], hbn[2];                                  // H^T B-fragments [k-step][slot] of the chunk phase B multiplies; k-step 1 of the next one
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) { hb[s2][tt] = make_uint4(0u, 0u, 0u, 0u); hbn[tt] = make_uint4(0u, 0u, 0u, 0u); }
    uint4 w[F4_NW];
    float mean[2] = {0.f, 0.f}, rstd[2] = {0.f, 0.f};
#define F4_W1F(ST, Q) (*reinterpret_cast<const uint4*>(f4_smem + (ST) * F4_RING + (Q) * 1024 + lane * 16))
#define F4_W2F(ST, Q) (*reinterpret_cast<const uint4*>(f4_smem + F4_W2_OFF + (ST) * F4_RING + (Q) * 1024 + lane * 16))
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < F4_NW; ++q) w[q] = F4_W1F(0, q);

    // H epilogue slice P (0..15): pair p = P & 7 (accumulator registers 2 p, 2 p + 1) of slot P >> 3
#define F4_BIAS(P) (*reinterpret_cast<const float2*>(b1c_ + 8 * (((P) & 7) >> 1) + 2 * ((P) & 1)))
#define F4_HEPI(P, BB)                                                                             \
    {                                                                                              \
        const int tt_ = (P) >> 3, p_ = (P) & 7;                                                    \
        const uint32_t v_ = pack_bf16x2(f4_relu(he[tt_][2 * p_] + (BB).x), f4_relu(he[tt_][2 * p_ + 1] + (BB).y)); \
        uint4& d_ = (p_ < 4) ? hb[0][tt_] : hbn[tt_];                                              \
        if ((p_ & 3) == 0) d_.x = v_;                                                              \
        else if ((p_ & 3) == 1) d_.y = v_;                                                         \
        else if ((p_ & 3) == 2) d_.z = v_;                                                         \
        else d_.w = v_;                                                                            \
    }
    // both slots computing: slice under phase-B group Q as in ffn3 (k-step-1 pairs of slot 0, of slot 1 under groups 0..7, the k-step-0
    // pairs -- written in place -- under groups 8..15).  One slot computing: its k-step-1 pairs under groups 4..7, its k-step-0 pairs under
    // 12..15 (never in the first groups: `he` comes from asm MFMAs the hazard recogniser does not see)
#define F4_SLICE2(Q) ((Q) < 8 ? 8 * ((Q) >> 2) + 4 + ((Q) & 3) : 8 * (((Q) - 8) >> 2) + (((Q) - 8) & 3))
#define F4_SLICE1(Q, TT) (8 * (TT) + ((Q) < 8 ? (Q) : (Q) - 12))
#define F4_HAS1(Q) (((Q) >= 4 && (Q) < 8) || (Q) >= 12)
#define F4_NEXT1(Q) ((Q) == 7 ? 12 : (Q) + 1)
    // ---- the chunk step.  C0 / C1 (compile time): slot 0 / slot 1 computes.  Ring stages: st = W1(s), stp = W2(s - 1) and the W1 stage being
    // refilled, stn = W1(s + 1), st2 = the W2 stage being refilled; c2 / c3 = chunks of steps s + 2 / s + 3; cs = chunk of this step.
#define F4_STEP(C0, C1, PRE)                                                                       \
    {                                                                                              \
        const float* b1c_ = reinterpret_cast<const float*>(f4_smem + F4_B1_OFF) + cs * 32 + 4 * hh; \
        float2 bnx_ = make_float2(0.f, 0.f);                                                       \
        (void)b1c_; (void)bnx_;                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                           \
            if (q == 0) { if (C0) f4_mma_v0(w[0], xf[0][0], he[0]); if (C1) f4_mma_v0(w[0], xf[0][1], he[1]); } \
            else { if (C0) f4_mma_v(w[q % F4_NW], xf[q][0], he[0]); if (C1) f4_mma_v(w[q % F4_NW], xf[q][1], he[1]); } \
            if (!((C0) && (C1))) asm volatile("s_nop 1");       /* one slot: the same accumulator back to back */ \
            if (q < 16 - F4_NW) w[q % F4_NW] = F4_W1F(st, q + F4_NW); else w[q % F4_NW] = F4_W2F(stp, q - (16 - F4_NW)); \
            if (q == 1) F4_PIECE2(c2, st2, 0) else if (q == 5) F4_PIECE2(c2, st2, 1) else if (q == 9) F4_PIECE2(c2, st2, 2) else if (q == 13) F4_PIECE2(c2, st2, 3) \
            if (q == 15) bnx_ = ((C0) && (C1)) ? F4_BIAS(F4_SLICE2(0)) : F4_BIAS(F4_SLICE1(4, (C1) ? 1 : 0)); \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
        /* the H^T fragments of a slot's first chunk step are zeros the compiler materialises wherever it likes -- possibly with a VALU \
           instruction directly in front of the first MFMA that reads them (tools/ffn4_lint.py found that): group 0 uses the padded form */ \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                           \
            if (C0) { if (q == 0) f4_mma_a_pad(w[0], hb[0][0], yacc[0][0]); else f4_mma_a(w[q % F4_NW], hb[q >> 3][0], yacc[q & 7][0]); } \
            if (C1) { if (q == 0) f4_mma_a_pad(w[0], hb[0][1], yacc[0][1]); else f4_mma_a(w[q % F4_NW], hb[q >> 3][1], yacc[q & 7][1]); } \
            if (q < 16 - F4_NW) w[q % F4_NW] = F4_W2F(stp, q + F4_NW); else if (PRE) w[q % F4_NW] = F4_W1F(stn, q - (16 - F4_NW));      /* !PRE: another slot's epilogue follows; the first fragments of the next step are read after it */ \
            if (q == 3) F4_PIECE1(c3, stp, 0) else if (q == 7) F4_PIECE1(c3, stp, 1) else if (q == 11) F4_PIECE1(c3, stp, 2) else if (q == 15) F4_PIECE1(c3, stp, 3) \
            if ((C0) && (C1)) {                                                                    \
                const float2 bcur_ = bnx_;                                                         \
                if (q < 15) bnx_ = F4_BIAS(F4_SLICE2(q + 1));                                      \
                F4_HEPI(F4_SLICE2(q), bcur_)                                                       \
            } else if (F4_HAS1(q)) {                                                               \
                const float2 bcur_ = bnx_;                                                         \
                if (q < 15) bnx_ = F4_BIAS(F4_SLICE1(F4_NEXT1(q), (C1) ? 1 : 0));                  \
                F4_HEPI(F4_SLICE1(q, (C1) ? 1 : 0), bcur_)                                         \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
        if (C0) hb[1][0] = hbn[0];                                                                 \
        if (C1) hb[1][1] = hbn[1];                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }

    // ---- the parts of a slot's own step (run after the other slot's chunk step) ------------------------------------------------------------
    // KIND: 0 seeding, 1 row sum, 2 centred squares, 3 / 4 scale + round + store of channel tiles 0..3 / 4..7; `extra` += stores issued.
#define F4_SPECIAL(TT, KIND, NTILE, extra)                                                         \
    {                                                                                              \
        if ((KIND) == 0) {                                                                         \
            /* accumulators = X (the residual) through the matrix pipe; H fragments = 0: the slot's first chunk step adds W2 . 0 */ \
            uint32_t ia[4], ib[4];                                                                 \
            _Pragma("unroll") for (int e2 = 0; e2 < 4; ++e2) {                                     \
                const int r0 = j - 8 * hh - 2 * e2, r1 = r0 - 16;                                  \
                ia[e2] = (r0 == 0 ? H16_ONE : 0u) | (r0 == 1 ? (H16_ONE << 16) : 0u);              \
                ib[e2] = (r1 == 0 ? H16_ONE : 0u) | (r1 == 1 ? (H16_ONE << 16) : 0u);              \
            }                                                                                      \
            const uint4 Ia = make_uint4(ia[0], ia[1], ia[2], ia[3]), Ib = make_uint4(ib[0], ib[1], ib[2], ib[3]); \
            _Pragma("unroll") for (int ct = 0; ct < 8; ++ct) f4_mma_a0_pad(Ia, xf[2 * ct][TT], yacc[ct][TT]);       \
            _Pragma("unroll") for (int ct = 0; ct < 8; ++ct) f4_mma_a_pad(Ib, xf[2 * ct + 1][TT], yacc[ct][TT]);    \
            hb[0][TT] = make_uint4(0u, 0u, 0u, 0u); hb[1][TT] = make_uint4(0u, 0u, 0u, 0u); hbn[TT] = make_uint4(0u, 0u, 0u, 0u); \
        } else if ((KIND) == 1) {                                                                  \
            const unsigned pa_ = f4_prm_addr(lds_base, hh);                                        \
            f4_f32x2_t s2 = {0.f, 0.f};                                                            \
            _Pragma("unroll") for (int ct = 0; ct < 8; ++ct)                                       \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                    \
                    const f4_f32x4_t bb = *(f4_lds4_t)(size_t)(pa_ + 4u * (32 * ct + 8 * q));              \
                    const f4_f32x4_t y4 = f4_acc_read4(yacc[ct][TT], q);          \
                    s2 += (f4_f32x2_t{y4.x, y4.y} + f4_f32x2_t{bb.x, bb.y}) + (f4_f32x2_t{y4.z, y4.w} + f4_f32x2_t{bb.z, bb.w}); \
                }                                                                                  \
            float sum = s2[0] + s2[1];                                                             \
            sum += __shfl_xor(sum, 32, 64);                                                        \
            mean[TT] = sum * (1.0f / 256.0f);                                                      \
        } else if ((KIND) == 2) {                                                                  \
            const unsigned pa_ = f4_prm_addr(lds_base, hh);                                        \
            const f4_f32x2_t m2 = {mean[TT], mean[TT]};                                            \
            f4_f32x2_t q2 = {0.f, 0.f};                                                            \
            _Pragma("unroll") for (int ct = 0; ct < 8; ++ct)                                       \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                    \
                    const f4_f32x4_t bb = *(f4_lds4_t)(size_t)(pa_ + 4u * (32 * ct + 8 * q));              \
                    const f4_f32x4_t y4 = f4_acc_read4(yacc[ct][TT], q);          \
                    const f4_f32x2_t d0 = (f4_f32x2_t{y4.x, y4.y} + f4_f32x2_t{bb.x, bb.y}) - m2;  \
                    const f4_f32x2_t d1 = (f4_f32x2_t{y4.z, y4.w} + f4_f32x2_t{bb.z, bb.w}) - m2;  \
                    q2 += d0 * d0 + d1 * d1;                                                       \
                }                                                                                  \
            float sq = q2[0] + q2[1];                                                              \
            sq += __shfl_xor(sq, 32, 64);                                                          \
            rstd[TT] = rsqrtf(sq * (1.0f / 256.0f) + eps);                                         \
        } else if ((KIND) == 3 || (KIND) == 4) {                                                   \
            const unsigned pa_ = f4_prm_addr(lds_base, hh);                                        \
            const f4_f32x2_t m2 = {mean[TT], mean[TT]}, r2 = {rstd[TT], rstd[TT]};                 \
            const long tok = F4_TOK(TT, NTILE);                                                    \
            /* a wave whose 32 rows all lie past M (ragged last tile) issues no stores and counts none */ \
            const bool wave_rows = __builtin_amdgcn_readfirstlane((int)(tok - j < (long)M)) != 0;  \
            if (wave_rows) {                                                                       \
                extra += 8;                                                                        \
                _Pragma("unroll") for (int ct4 = 0; ct4 < 4; ++ct4) {                              \
                    _Pragma("unroll") for (int half_ = 0; half_ < 2; ++half_) {                    \
                        if (half_ != (KIND) - 3) continue;                                         \
                        const int ct = 4 * half_ + ct4;                                            \
                        _Pragma("unroll") for (int qp = 0; qp < 2; ++qp) {                         \
                            uint32_t pk[2][2];                                                     \
                            _Pragma("unroll") for (int qo = 0; qo < 2; ++qo) {                     \
                                const int q = 2 * qp + qo, ch = 32 * ct + 8 * q;                   \
                                const f4_f32x4_t y4 = f4_acc_read4(yacc[ct][TT], q); \
                                _Pragma("unroll") for (int e = 0; e < 2; ++e) {                    \
                                    /* one channel pair at a time (few live temporaries: every register of the file is in use here) */ \
                                    const f4_f32x2_t ga = *(f4_lds2_t)(size_t)(pa_ + 4u * (256 + ch + 2 * e)); \
                                    const f4_f32x2_t be = *(f4_lds2_t)(size_t)(pa_ + 4u * (512 + ch + 2 * e)); \
                                    const f4_f32x2_t bb = *(f4_lds2_t)(size_t)(pa_ + 4u * (ch + 2 * e));   \
                                    const f4_f32x2_t g2 = ga * r2;                                 \
                                    const f4_f32x2_t cc = be + (bb - m2) * g2;                     \
                                    const f4_f32x2_t o2 = (e ? f4_f32x2_t{y4.z, y4.w} : f4_f32x2_t{y4.x, y4.y}) * g2 + cc; \
                                    pk[qo][e] = pack_bf16x2(o2[0], o2[1]);                         \
                                    __builtin_amdgcn_sched_barrier(0);                             \
                                }                                                                  \
                            }                                                                      \
                            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false); \
                            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false); \
                            if (tok < M)                                                           \
                                *reinterpret_cast<uint4*>(Y + tok * 256 + 32 * ct + 16 * qp + 8 * hh) = make_uint4(s0[0], s1[0], s0[1], s1[1]); \
                            __builtin_amdgcn_sched_barrier(0);                                     \
                        }                                                                          \
                    }                                                                              \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
    }
    // ---- the schedule.  Stream step s: slot 0 has its epilogue / seeding step at s = i PER, slot 1 at s = HALF + i PER; in between both
    // run chunk steps (slot 1 idles before its first seeding step, slot 0 after its last epilogue).  Every loop below has ONE body and the
    // steps with an epilogue are straight-line code between the loops: with several step bodies or a switch over epilogue slices inside one
    // loop hipcc's register allocation (every register of the file is in use) ended in ~1000 spilled registers.
    int post1 = 0, tot1 = 0, post2 = 0;
    int s = 0, cs = 0;                                         // stream step; its chunk = s mod nchunk
    int st = 0, stp = 3, stn = 1, st2 = 2, c2 = 0, c3 = 0;
#define F4_TOP                                                                                     \
    {                                                                                              \
        if (s > 0) {   /* the pieces of step s - 2 have landed (mine); then everybody's */        \
            const int allow = post2 + tot1;                                                        \
            if (allow <= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");            \
            else if (allow <= 24) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");     \
            else asm volatile("s_waitcnt vmcnt(40) lgkmcnt(0)" ::: "memory");                      \
            __builtin_amdgcn_s_barrier();                                                          \
        }                                                                                          \
        st = s & 3; stp = (s + 3) & 3; stn = (s + 1) & 3; st2 = (s + 2) & 3;                       \
        c2 = cs + 2; c2 = c2 >= nchunk ? c2 - nchunk : c2; c2 = c2 >= nchunk ? c2 - nchunk : c2;   \
        c3 = cs + 3; c3 = c3 >= nchunk ? c3 - nchunk : c3; c3 = c3 >= nchunk ? c3 - nchunk : c3;   \
    }
#define F4_END(E) { post2 = post1; post1 = (E); tot1 = 8 + (E); ++s; cs = cs + 1 == nchunk ? 0 : cs + 1; }
#define F4_PRELOAD { _Pragma("unroll") for (int q = 0; q < F4_NW; ++q) w[q] = F4_W1F(stn, q); }
    // A slot's own step: epilogue of its tile I_EPI (if DO_EPI), then the rows of tile I_NEXT are loaded and seeded.  Loads and seeding are
    // UNCONDITIONAL (past the last tile: clamped rows, accumulators nobody reads): hipcc's wait insertion follows control flow, not
    // conditions -- with the loads under one `if` and the seeding under another it assumed a path with the loads issued and the seeding
    // skipped and put vmcnt waits for them into the chunk loops.  The loads go between the statistics passes and the store slices: the
    // slot's X^T registers are dead since its last chunk step, but the statistics passes need them as temporaries.  e_ = vector-memory
    // operations issued here (all after the step's DMA pieces).
#define F4_SPECIAL_STEP(TT, DO_EPI, I_EPI, I_NEXT, e_)                                             \
    {                                                                                              \
        int stc_ = 0;                                                                              \
        if (DO_EPI) {                                                                              \
            F4_SPECIAL(TT, 1, I_EPI, stc_)                                                         \
            F4_SPECIAL(TT, 2, I_EPI, stc_)                                                         \
        }                                                                                          \
        if (!dbg_noinit) {                                                                         \
        F4_LOADX(TT, I_NEXT)                                                                       \
        if (DO_EPI) {                                                                              \
            F4_SPECIAL(TT, 3, I_EPI, stc_)                                                         \
            F4_SPECIAL(TT, 4, I_EPI, stc_)                                                         \
        }                                                                                          \
        /* the rows have landed: only the stores issued after them may stay in flight */          \
        if (stc_) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        F4_SPECIAL(TT, 0, I_NEXT, stc_)                                                            \
        e_ = 16 + stc_;                                                                            \
        } else e_ = 0;                                                                             \
    }

    // The schedule as straight-line steps and single-body loops (a loop with several step bodies, or a step body under a condition, costs
    // hipcc's register allocation hundreds of spills: every register of the file is in use).  The store slices exist three times (slot 1 in
    // the tile loop, slot 0 in the tile loop, slot 1's last tile): 46 KB of code in all.  (First build: five copies, 61 KB for a 64 KB
    // instruction cache shared by two compute units -- every epilogue step ran 20 KB of straight-line code from L2.)
    {   // s = 0: slot 0 loads and seeds its first tile; no chunk step yet
        F4_TOP
        F4_ISSUE2(c2, st2)
        F4_ISSUE1(c3, stp)
        int e = 0;
        F4_SPECIAL_STEP(0, false, 0, 0, e)
        F4_PRELOAD
        F4_END(e)
    }
    for (int k = 0; k < HALF - 1; ++k) {                        // slot 1 has not started
        F4_TOP
        F4_STEP(true, false, true)
        F4_END(0)
    }
    for (int i = 0; i < np; ++i) {
        {   // s = lag + i PER: slot 0 computes; slot 1 stores tile i - 1, loads and seeds tile i
            F4_TOP
            F4_STEP(true, false, false)
            int e = 0;
            F4_SPECIAL_STEP(1, i > 0 && !dbg_noepi, i - 1, i, e)
            F4_PRELOAD
            F4_END(e)
        }
        for (int k = 0; k < PER - HALF - 1; ++k) {
            F4_TOP
            F4_STEP(true, true, true)
            F4_END(0)
        }
        {   // s = (i + 1) PER: slot 1 computes; slot 0 stores tile i, loads and seeds tile i + 1 (clamped rows past the last one)
            F4_TOP
            F4_STEP(false, true, false)
            int e = 0;
            F4_SPECIAL_STEP(0, !dbg_noepi, i, i + 1, e)
            F4_PRELOAD
            F4_END(e)
        }
        if (i + 1 < np) {
            for (int k = 0; k < HALF - 1; ++k) {
                F4_TOP
                F4_STEP(true, true, true)
                F4_END(0)
            }
        }
    }
    for (int k = 0; k < HALF - 1; ++k) {                        // slot 0 has finished
        F4_TOP
        F4_STEP(false, true, true)
        F4_END(0)
    }
    {   // the last step: slot 1 stores its last tile
        F4_TOP
        F4_ISSUE2(c2, st2)
        F4_ISSUE1(c3, stp)
        int e = 0;
        F4_SPECIAL_STEP(1, true, np - 1, np, e)
        (void)e;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef F4_SPECIAL_STEP
#undef F4_PRELOAD
#undef F4_END
#undef F4_TOP
#undef F4_SPECIAL
#undef F4_STEP
#undef F4_NEXT1
#undef F4_HAS1
#undef F4_SLICE1
#undef F4_SLICE2
#undef F4_HEPI
#undef F4_BIAS
#undef F4_W1F
#undef F4_W2F
#undef F4_LOADX
#undef F4_TOK
#undef F4_ISSUE1
#undef F4_ISSUE2
#undef F4_PIECE1
#undef F4_PIECE2
}

// X, Y [M, 256] 16-bit; W1p / W2p = device copies of the images of dtlr_ffn32_pack_weights; b1 [d_ff], b2 / gamma / beta [256] fp32.
extern "C" int dtlr_ffn4_bf16(const void* X, const void* W1p, const float* b1, const void* W2p, const float* b2,
                              const float* gamma, const float* beta, float eps, void* Y, long M, int d_ff, void* stream)
{
    clear_stale_error();
    if (!X || !W1p || !b1 || !W2p || !b2 || !gamma || !beta || !Y) return DTLR_EINVAL;
    if (M <= 0 || M > 0x7fffffffL) return DTLR_EINVAL;
    if (d_ff < 64 || d_ff > F4_MAX_DFF || (d_ff & 31)) return DTLR_ESHAPE;
    static DevOnce once;
    if (once.first()) { (void)hipFuncSetAttribute((const void*)ffn4_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F4_LDS); (void)hipGetLastError(); }
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    static int cu_cache[64];
    if (cu_cache[dev & 63] == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cu_cache[dev & 63] = cus;
    }
    cus = cu_cache[dev & 63];
    const long NT = (M + 127) / 128;
    static const int lag = exp_env_int("DTLR_FFN4_LAG", 2);     // experiment builds only
    const long NPAIR = (NT + 1) / 2;
    const long G = NPAIR >= cus ? cus : NPAIR;
    hipLaunchKernelGGL(ffn4_bf16_kernel, dim3((unsigned)G), dim3(256), F4_LDS, (hipStream_t)stream,
                       (const uint16_t*)X, (const uint16_t*)W1p, b1, (const uint16_t*)W2p, b2, gamma, beta, eps, (uint16_t*)Y, (int)M, d_ff, lag);
    return check_launch();
}

}  // namespace dtlr
