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

// ---- bf16 <-> f32: gfx950 converts in hardware (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays a quiet NaN).
// The first version rounded in software (~10 VALU instructions per pair); in the K = 256 GEMMs that made the
// epilogue's VALU time equal to the tile's MFMA time.
typedef float f32x2_hw_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_hw_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2_hw_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_hw_t));
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return (uint16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

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
