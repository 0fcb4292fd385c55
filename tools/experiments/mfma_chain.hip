// cycles per v_mfma_f32_32x32x16_bf16 for one wave per SIMD as a function of how many independent accumulators the stream rotates
// through, and of one LDS read between groups.   hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int NACC, bool LDS, bool VFORM>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, long long* cyc)
{
    __shared__ uint4 sm[4096];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = threadIdx.x * 0.001f + i;
    u32x4 a = {threadIdx.x, 1, 2, 3}, b = {4, 5, threadIdx.x, 7};
    sm[threadIdx.x] = make_uint4(1, 2, 3, 4);
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if constexpr (VFORM) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u % NACC]) : "v"(a), "v"(b));
            else acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[u % NACC], 0, 0, 0);
            if (LDS && (u & 1)) { const uint4 v = sm[(threadIdx.x + u * 64 + it) & 4095]; a[0] ^= v.x; }
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC, bool LDS, bool VFORM> void run(const char* name, float* out, long long* cyc)
{
    const int iters = 2000;
    hipLaunchKernelGGL((k<NACC, LDS, VFORM>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<NACC, LDS, VFORM>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-40s %7.3f ms  clock64 ticks per MFMA %.1f   ns per MFMA %.2f\n", name, ms, (double)c / (iters * 32.0), ms * 1e6 / (iters * 32.0));
}
int main()
{
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<1, false, false>("1 acc, builtin", out, cyc);
    run<2, false, false>("2 acc, builtin", out, cyc);
    run<4, false, false>("4 acc, builtin", out, cyc);
    run<8, false, false>("8 acc, builtin", out, cyc);
    run<2, false, true>("2 acc, asm VGPR form", out, cyc);
    run<4, false, true>("4 acc, asm VGPR form", out, cyc);
    run<2, true, false>("2 acc + ds_read per pair, builtin", out, cyc);
    run<4, true, false>("4 acc + ds_read per pair, builtin", out, cyc);
    run<2, true, true>("2 acc + ds_read per pair, asm VGPR", out, cyc);
    run<4, true, true>("4 acc + ds_read per pair, asm VGPR", out, cyc);
    return 0;
}
