// Discrete selections of the path, one workgroup per line, everything in LDS:
//   dtlr_topk_rows     two-stage query selection: indices of the k largest scores per row, descending
//                      (torch.topk(enc_outputs_class.max(-1)[0], 900, dim=1)[1], deformable_transformer.py:345)
//   dtlr_decode_blank  the blank/argmax decoder (evaluation.py:116-158 == dino.py:466-502 + engine.py:511-530):
//                      sort queries by box cx, sigmoid, blank-channel construction, argmax, drop blanks
// Both sorts are bitonic networks over 64-bit keys (value bits | index) so the order is total and
// deterministic: equal scores keep the LOWER index first (torch leaves ties unspecified).
#include "dtlr_common.h"

namespace dtlr {

// monotone map float -> uint32 (ascending)
__device__ __forceinline__ uint32_t f32_sortable(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// in-LDS bitonic sort of n = power of two 64-bit keys, ascending
__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* keys, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void topk_rows_kernel(const float* __restrict__ scores, long* __restrict__ idx_out,
                                                         int S, int k, int npow2)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int b = blockIdx.x;
    const float* row = scores + (long)b * S;
    // ascending sort of key = (~sortable(score) << 32) | index  ==  descending score, ascending index
    for (int i = threadIdx.x; i < npow2; i += blockDim.x)
        keys[i] = i < S ? (((unsigned long long)(~f32_sortable(row[i]))) << 32) | (unsigned)i : ~0ull;
    bitonic_sort_u64(keys, npow2);
    for (int i = threadIdx.x; i < k; i += blockDim.x) idx_out[(long)b * k + i] = (long)(keys[i] & 0xffffffffull);
}

// logits [B,nq,C] fp32, boxes [B,nq,4] fp32 -> labels [B,nq] int32 (left-packed, -1 padded), lengths [B]
__global__ __launch_bounds__(1024) void decode_blank_kernel(const float* __restrict__ logits, const float* __restrict__ boxes,
                                                            int* __restrict__ labels, int* __restrict__ lengths,
                                                            int nq, int C, float eps, int npow2)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];      // [npow2] then int lab[npow2]
    int* lab = reinterpret_cast<int*>(keys + npow2);
    __shared__ int wave_tot[16];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (int i = threadIdx.x; i < npow2; i += blockDim.x)
        keys[i] = i < nq ? (((unsigned long long)f32_sortable(boxes[((long)b * nq + i) * 4])) << 32) | (unsigned)i : ~0ull;
    bitonic_sort_u64(keys, npow2);                                             // ascending cx, ties: lower index first
    // one wave per sorted position
    for (int p = wave; p < nq; p += nwave) {
        const int q = (int)(keys[p] & 0xffffffffull);
        const float* lr = logits + ((long)b * nq + q) * C;
        float sum = 0.f, best = -1.f;
        int arg = 0x7fffffff;
        for (int c = lane; c < C; c += 64) {
            const float pr = 1.f / (1.f + expf(-lr[c]));
            sum += pr;
            if (pr > best) { best = pr; arg = c; }                             // first maximum within the lane's stride
        }
        sum = wave_sum(sum);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {                                     // max with lowest-index tie break
            const float ob = __shfl_xor(best, o, 64);
            const int oa = __shfl_xor(arg, o, 64);
            if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
        // blank channel (dino.py:489-502): sum < 1-eps -> blank = 1-sum ; else blank = eps, p <- (1-eps) p / sum
        float blank, top;
        if (sum < 1.f - eps) { blank = 1.f - sum; top = best; }
        else { blank = eps; top = (1.f - eps) * best / sum; }
        if (lane == 0) lab[p] = (blank >= top) ? -1 : arg;                     // argmax over [blank | classes]: blank wins ties
    }
    __syncthreads();
    // stable compaction of the non-blank labels: block-wide exclusive scan of keep flags
    int running = 0;
    for (int base = 0; base < nq; base += blockDim.x) {
        const int p = base + threadIdx.x;
        const int v = p < nq ? lab[p] : -1;
        const int keep = v >= 0;
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int off = running;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        int tot = 0;
        for (int w = 0; w < nwave; ++w) tot += wave_tot[w];
        if (keep) labels[(long)b * nq + off + before] = v;
        running += tot;
        __syncthreads();
    }
    for (int i = running + threadIdx.x; i < nq; i += blockDim.x) labels[(long)b * nq + i] = -1;
    if (threadIdx.x == 0) lengths[b] = running;
}

static inline int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

}  // namespace dtlr

using namespace dtlr;

extern "C" int dtlr_topk_rows(const float* scores, long* idx_out, int B, int S, int k, void* stream)
{
    clear_stale_error();
    if (!scores || !idx_out) return DTLR_EINVAL;
    if (B <= 0 || S <= 0 || k <= 0 || k > S) return DTLR_EINVAL;
    const int np = next_pow2(S);
    const size_t lds = (size_t)np * 8;
    if (lds > 160 * 1024) return DTLR_ESHAPE;
    (void)hipGetLastError();                                   // do not inherit a stale error from an earlier API call
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)topk_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    hipLaunchKernelGGL(topk_rows_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, scores, idx_out, S, k, np);
    return check_launch();
}

extern "C" int dtlr_decode_blank(const float* logits, const float* boxes, int* labels, int* lengths,
                                 int B, int nq, int C, float eps, void* stream)
{
    clear_stale_error();
    if (!logits || !boxes || !labels || !lengths) return DTLR_EINVAL;
    if (B <= 0 || nq <= 0 || C <= 0) return DTLR_EINVAL;
    const int np = next_pow2(nq);
    const size_t lds = (size_t)np * 12;
    if (lds > 150 * 1024) return DTLR_ESHAPE;
    (void)hipGetLastError();
    if (lds > 60 * 1024) (void)hipFuncSetAttribute((const void*)decode_blank_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    hipLaunchKernelGGL(decode_blank_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, logits, boxes, labels, lengths, nq, C, eps, np);
    return check_launch();
}
