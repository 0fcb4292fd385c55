// Shared helpers for the gfx950 kernels of libdtlr_hip.so.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dtlr_hip.h"

namespace dtlr {

constexpr int kWave = 64;

extern thread_local int g_last_hip_error;

// Call at the top of every C-ABI entry point: the HIP "last error" is sticky per thread, so an error left by an
// unrelated earlier call (e.g. a failed attribute query) must not be blamed on this launch.
inline void clear_stale_error() { (void)hipGetLastError(); }

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = (int)e; return DTLR_ELAUNCH; }
    return DTLR_OK;
}

// One-time per-DEVICE setup (hipFuncSetAttribute is a per-device property: a process that drives several GPUs must repeat it
// on each): `static DevOnce once; if (once.first()) { ... }`.
struct DevOnce {
    unsigned long long mask = 0;
    bool first() {
        int d = 0;
        (void)hipGetDevice(&d);
        const unsigned long long bit = 1ull << (d & 63);
        if (mask & bit) return false;
        mask |= bit;
        return true;
    }
};

// ---- the 16-bit storage / MFMA-operand format of this library ------------------------------------------------------------
// libdtlr_hip.so is compiled with bf16 as "the" 16-bit format; the same sources compiled with -DDTLR_HALF_IS_F16 give
// libdtlr_hip_f16.so, whose 16-bit format is IEEE fp16 (unit roundoff 2^-11 against bf16's 2^-8: 8x finer, same MFMA rate, same
// bytes -- the parity build; range 65504, which every post-normalisation activation of this network fits).  Everything that
// depends on the format goes through the helpers below: the dtype code the entry points accept (DTLR_H16), the conversions
// (hardware on gfx950: v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, round-to-nearest-even; NaN stays a quiet NaN), the unpacking
// of the halves of a 32-bit word, and the MFMA instructions.  The first bf16 version rounded in software (~10 VALU
// instructions per pair); in the K = 256 GEMMs that made the epilogue's VALU time equal to the tile's MFMA time.
typedef float f32x2_hw_t __attribute__((ext_vector_type(2)));
#ifdef DTLR_HALF_IS_F16
typedef _Float16 h16_hw_t;
constexpr int DTLR_H16 = DTLR_F16;
#define DTLR_MFMA_16x16x32_H16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define DTLR_MFMA_32x32x16_H16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define DTLR_H16_ASM_SUFFIX "f16"
constexpr uint32_t H16_ONE = 0x3c00u;        // 1.0 as a 16-bit pattern
__host__ __device__ __forceinline__ float h16_lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
__host__ __device__ __forceinline__ float h16_hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
inline uint16_t f32_to_h16_host(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }      // RNE (compiler-rt / F16C)
#else
typedef __bf16 h16_hw_t;
constexpr int DTLR_H16 = DTLR_BF16;
#define DTLR_MFMA_16x16x32_H16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define DTLR_MFMA_32x32x16_H16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define DTLR_H16_ASM_SUFFIX "bf16"
constexpr uint32_t H16_ONE = 0x3f80u;
__host__ __device__ __forceinline__ float h16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__host__ __device__ __forceinline__ float h16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
inline uint16_t f32_to_h16_host(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);      // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                                // round to nearest even
    return (uint16_t)(u >> 16);
}
#endif
typedef h16_hw_t h16x2_hw_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return h16_lo((uint32_t)h); }        // (historic names: "bf16" = the library's 16-bit format)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2_hw_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, h16x2_hw_t));
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return (uint16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

// ---- A/B switches of the measurement builds.  The product libraries are compiled WITHOUT -DDTLR_EXPERIMENT: every switch below is then
// its default, a compile-time constant, and no environment variable can change what a kernel launch does.  `python -m dtlr_amd.build
// --instr` (never loaded by the product) defines DTLR_EXPERIMENT and reads them from the environment, once per process.
#ifdef DTLR_EXPERIMENT
#include <stdlib.h>
inline int exp_env_int(const char* name, int dflt) { const char* e = getenv(name); return (e && e[0]) ? atoi(e) : dflt; }
#else
constexpr int exp_env_int(const char*, int dflt) { return dflt; }
#endif

// Scratch buffer of at least `bytes` owned by (current device, stream): defined in gemm.hip (the split-K workspace).  Contents are only
// meaningful between the launches of ONE operator on that stream; nullptr when no buffer can be had (capture without a warm slot, OOM).
float* stream_workspace(size_t bytes, hipStream_t st);

// ---- wave reductions (all 64 lanes participate) ----------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace dtlr
