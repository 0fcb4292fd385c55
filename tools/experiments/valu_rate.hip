// Issue cost of the VALU instructions the encoder sampler (msda_enc.hip) is made of, on gfx950: ns per wave-instruction with 1, 2 and 4
// waves per SIMD (256 / 512 / 1024 threads per workgroup, one workgroup per CU), 16 independent destination registers per stream.
// Question it answers: is v_pk_fma_f16 (the sampling itself: 256 per lane-iteration) a one-pass instruction like v_fma_f32, and what do the
// other instructions of the lane-iteration (cvt_pk, med3, floor, exp, rcp, mul_lo, DPP moves, ds_read_b128) cost beside it?
//     hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Op { FMA_F32, PK_FMA_F16, PK_FMA_F16_OPSEL, PK_FMA_F32, FMA_MIX, CVT_PK_F16, MED3_F32, FLOOR, CVT_I32, EXP, RCP, MUL_LO, DPP_MOV, ADD_U32,
          LSHL_ADD_U64, CNDMASK, PK_MUL_F32, MIX_PKFMA_DS, DS_B128 };

template <int OP>
__global__ void k(float* out, int iters, long long* cyc)
{
    __shared__ uint4 sm[2048];
    uint32_t r[16];
    for (int i = 0; i < 16; ++i) r[i] = 0x3c003c00u + threadIdx.x + i;
    uint32_t a = 0x38003800u + threadIdx.x, b = 0x3c003c01u;
    uint64_t r64[16];
    for (int i = 0; i < 16; ++i) r64[i] = threadIdx.x + i;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) sm[i] = make_uint4(i, 2, 3, 4);
    __syncthreads();
    const unsigned ldsaddr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#define X(i) \
            if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b)); \
            else if constexpr (OP == PK_FMA_F16) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b)); \
            else if constexpr (OP == PK_FMA_F16_OPSEL) asm volatile("v_pk_fma_f16 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(r[i]) : "v"(a), "v"(b)); \
            else if constexpr (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r64[i]) : "v"(r64[(i + 1) & 15]), "v"(r64[(i + 2) & 15])); \
            else if constexpr (OP == FMA_MIX) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(r[i]) : "v"(a), "v"(b)); \
            else if constexpr (OP == CVT_PK_F16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r[i]) : "v"(a), "v"(b)); \
            else if constexpr (OP == MED3_F32) asm volatile("v_med3_f32 %0, %1, -1.0, %2" : "=v"(r[i]) : "v"(a), "v"(b)); \
            else if constexpr (OP == FLOOR) asm volatile("v_floor_f32 %0, %1" : "=v"(r[i]) : "v"(a)); \
            else if constexpr (OP == CVT_I32) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(r[i]) : "v"(a)); \
            else if constexpr (OP == EXP) asm volatile("v_exp_f32 %0, %1" : "=v"(r[i]) : "v"(a)); \
            else if constexpr (OP == RCP) asm volatile("v_rcp_f32 %0, %1" : "=v"(r[i]) : "v"(a)); \
            else if constexpr (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r[i]) : "v"(a), "v"(b)); \
            else if constexpr (OP == DPP_MOV) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(a)); \
            else if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %1, %2" : "=v"(r[i]) : "v"(a), "v"(b)); \
            else if constexpr (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %1, 2, %2" : "=v"(r64[i]) : "v"(r64[(i + 1) & 15]), "v"(r64[(i + 2) & 15])); \
            else if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r[i]) : "v"(a), "v"(b)); \
            else if constexpr (OP == PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r64[i]) : "v"(r64[(i + 1) & 15]), "v"(r64[(i + 2) & 15])); \
            else if constexpr (OP == DS_B128) { uint4 d; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ldsaddr), "n"((i & 3) * 4096)); r[i] ^= d.x; } \
            else if constexpr (OP == MIX_PKFMA_DS) { \
                if ((i & 3) == 0) { uint4 d; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ldsaddr), "n"((i & 12) * 1024)); a ^= d.x & 1u; } \
                asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b)); \
            }
            REP16(X)
#undef X
        }
    }
    const long long t1 = clock64();
    uint32_t s = 0;
    for (int i = 0; i < 16; ++i) s += r[i] + (uint32_t)r64[i] + (uint32_t)(r64[i] >> 32);
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(s);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP> void run(const char* name, float* out, long long* cyc)
{
    const int iters = 4000;
    printf("%-34s", name);
    for (int nt = 256; nt <= 1024; nt *= 2) {
        hipLaunchKernelGGL((k<OP>), dim3(256), dim3(nt), 0, 0, out, 10, cyc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP>), dim3(256), dim3(nt), 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double n = iters * 64.0;                      // instructions per wave
        const int wps = nt / 256;
        // ns of SIMD time per wave-instruction = wall / (instructions per wave x waves per SIMD)
        printf("  %dw/SIMD: %6.2f ns/instr/SIMD (%5.1f ticks/instr/wave)", wps, ms * 1e6 / (n * wps), (double)c / n);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("\n");
}

int main()
{
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    run<FMA_F32>("v_fma_f32", out, cyc);
    run<PK_FMA_F16>("v_pk_fma_f16", out, cyc);
    run<PK_FMA_F16_OPSEL>("v_pk_fma_f16 op_sel (broadcast hi)", out, cyc);
    run<PK_FMA_F32>("v_pk_fma_f32", out, cyc);
    run<PK_MUL_F32>("v_pk_mul_f32", out, cyc);
    run<FMA_MIX>("v_fma_mix_f32", out, cyc);
    run<CVT_PK_F16>("v_cvt_pk_f16_f32", out, cyc);
    run<MED3_F32>("v_med3_f32", out, cyc);
    run<FLOOR>("v_floor_f32", out, cyc);
    run<CVT_I32>("v_cvt_i32_f32", out, cyc);
    run<EXP>("v_exp_f32", out, cyc);
    run<RCP>("v_rcp_f32", out, cyc);
    run<MUL_LO>("v_mul_lo_u32", out, cyc);
    run<ADD_U32>("v_add_u32", out, cyc);
    run<LSHL_ADD_U64>("v_lshl_add_u64", out, cyc);
    run<CNDMASK>("v_cndmask_b32", out, cyc);
    run<DPP_MOV>("v_mov_b32_dpp quad_perm", out, cyc);
    run<DS_B128>("ds_read_b128 (conflict-free)", out, cyc);
    run<MIX_PKFMA_DS>("4 v_pk_fma_f16 + 1 ds_read_b128", out, cyc);
    return 0;
}
