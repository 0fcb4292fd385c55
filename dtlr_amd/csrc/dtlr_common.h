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

// ---- bf16 <-> f32 (round-to-nearest-even; NaN kept quiet) -------------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

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
