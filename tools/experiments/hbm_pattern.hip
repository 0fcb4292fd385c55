// HBM access-pattern probe: read a [rows x 512 B] matrix (a) slab-wise, 128 rows x 128 B at a time (the GEMM loader's pattern for
// K = 256 bf16), (b) as whole 512-byte rows.  Prints GB/s.  hipcc --offload-arch=gfx950 -O3 hbm_pattern.hip -o hbm_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)

// block = 256 threads, owns 128-row tiles in a chain; per tile 4 slabs of 128 B columns; lane (row = tid>>3 (+32 i), kc = tid&7)
__global__ __launch_bounds__(256) void slab_kernel(const uint4* __restrict__ x, uint4* __restrict__ sink, int ntiles, int per)
{
    const int tid = threadIdx.x, srow = tid >> 3, kc = tid & 7;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int t = blockIdx.x * per; t < min((int)(blockIdx.x + 1) * per, ntiles); ++t)
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long row = (long)t * 128 + srow + 32 * i;
                const uint4 v = x[row * 32 + s * 8 + kc];
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
    if (acc.x == 0x12345678u) sink[blockIdx.x * 256 + tid] = acc;
}
// same tiles, but each wave-instruction reads 2 whole rows (32 lanes x 16 B = 512 B)
__global__ __launch_bounds__(256) void row_kernel(const uint4* __restrict__ x, uint4* __restrict__ sink, int ntiles, int per)
{
    const int tid = threadIdx.x, r8 = tid >> 5, c = tid & 31;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int t = blockIdx.x * per; t < min((int)(blockIdx.x + 1) * per, ntiles); ++t)
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const long row = (long)t * 128 + r8 + 8 * i;
            const uint4 v = x[row * 32 + c];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    if (acc.x == 0x12345678u) sink[blockIdx.x * 256 + tid] = acc;
}

int main()
{
    const long rows = 174080L * 4;                      // 356 MB: beyond L2 + MALL
    const int ntiles = (int)(rows / 128);
    uint4 *x, *sink;
    CK(hipMalloc(&x, rows * 512));
    CK(hipMalloc(&sink, 4096 * 256 * 16));
    CK(hipMemset(x, 1, rows * 512));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int nblk : {512, 1024, 2048, 4096}) {
        const int per = (ntiles + nblk - 1) / nblk;
        for (int which = 0; which < 2; ++which) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(a));
                if (which == 0) hipLaunchKernelGGL(slab_kernel, dim3(nblk), dim3(256), 0, 0, x, sink, ntiles, per);
                else hipLaunchKernelGGL(row_kernel, dim3(nblk), dim3(256), 0, 0, x, sink, ntiles, per);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            printf("%s blocks=%d  %.3f ms  %.1f GB/s\n", which == 0 ? "slab(128x128B)" : "rows(2x512B) ", nblk, best, rows * 512.0 / best / 1e6);
        }
    }
    return 0;
}
