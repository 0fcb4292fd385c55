// How many cycles does one wave-wide global load/store cost in the vector L1 as a function of its lane -> address pattern?
// Buffer is small (L1-resident for loads).  One workgroup of 8 waves per CU, every wave loops the same pattern.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l1_pattern tools/experiments/l1_pattern.hip && /tmp/l1_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int BYTES, int ROW_LANES>      // BYTES per lane; ROW_LANES consecutive lanes share a row (row pitch 768 B)
__global__ __launch_bounds__(512) void ld_kernel(const uint8_t* __restrict__ buf, float* out, int iters, long long* cyc)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane / ROW_LANES, col = (lane % ROW_LANES) * BYTES + wave * 96;      // like the 48-channel strip of wave `wave`
    const uint8_t* p = buf + (size_t)row * 768 + (col % 768);
    float acc = 0.f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint8_t* q = p + (size_t)((i * 8 + u) & 3) * (64 / ROW_LANES) * 768;
            if constexpr (BYTES == 16) { uint4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(q) : "memory"); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); acc += __uint_as_float(v.x); }
            else if constexpr (BYTES == 8) { uint2 v; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(q) : "memory"); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); acc += __uint_as_float(v.x); }
            else { uint32_t v; asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(q) : "memory"); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); acc += __uint_as_float(v); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 123.456f) out[threadIdx.x] = acc;
}

template <int BYTES, int ROW_LANES>
__global__ __launch_bounds__(512) void st_kernel(uint8_t* __restrict__ buf, int iters, long long* cyc)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane / ROW_LANES, col = (lane % ROW_LANES) * BYTES + wave * 96;
    uint8_t* p = buf + (size_t)blockIdx.x * (1 << 20) + (size_t)row * 768 + (col % 768);
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint8_t* q = p + (size_t)((i * 8 + u) & 63) * (64 / ROW_LANES) * 768;
            if constexpr (BYTES == 16) *reinterpret_cast<uint4*>(q) = make_uint4(i, u, 0, 0);
            else if constexpr (BYTES == 8) *reinterpret_cast<uint2*>(q) = make_uint2(i, u);
            else *reinterpret_cast<uint32_t*>(q) = i;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename F> static void run(const char* name, F launch, int iters, long long* dcyc, int nblk)
{
    launch();
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c[4]; hipMemcpy(c, dcyc, sizeof(c), hipMemcpyDeviceToHost);
    const double instr = (double)iters * 8 * 8;      // wave-instructions per CU (8 waves)
    printf("%-44s %8.3f ms  %7.1f ns per wave-instr per CU  (clock64 ticks/instr/CU: %.1f)\n", name, ms, ms * 1e6 / instr, (double)c[0] / instr);
}

int main()
{
    const int nblk = 256, iters = 2000;
    uint8_t* buf; float* out; long long* cyc;
    hipMalloc(&buf, (size_t)nblk << 20); hipMalloc(&out, 4096); hipMalloc(&cyc, nblk * 8);
    hipMemset(buf, 0, (size_t)nblk << 20);
#define LD(B, R) run("load  " #B " B/lane, " #R " lanes per 768-B row", [&] { hipLaunchKernelGGL((ld_kernel<B, R>), dim3(nblk), dim3(512), 0, 0, buf, out, iters, cyc); }, iters, cyc, nblk);
#define ST(B, R) run("store " #B " B/lane, " #R " lanes per 768-B row", [&] { hipLaunchKernelGGL((st_kernel<B, R>), dim3(nblk), dim3(512), 0, 0, buf, iters, cyc); }, iters, cyc, nblk);
    LD(16, 64) LD(16, 8) LD(16, 4) LD(8, 4) LD(8, 8) LD(4, 4) LD(16, 2)
    ST(16, 64) ST(16, 8) ST(16, 4) ST(8, 4) ST(4, 4)
    return 0;
}
