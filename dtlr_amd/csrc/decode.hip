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

// in-LDS bitonic sort of n = power of two 64-bit keys, ascending.  A thread owns compare-exchange PAIRS (pair t of stage j is
// i = the index with bit j cleared, i | j), four at a time, and reads all eight keys before it writes any: one LDS round trip
// per stage instead of one per element (the element-wise form serialised 8 dependent read->write trips per stage at n = 8192
// and made the two selection kernels ~100 us each).
__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* keys, int n) {
    const int half = n >> 1;
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t0 = threadIdx.x; t0 < half; t0 += 4 * blockDim.x) {
                unsigned long long a[4], b[4];
                int ia[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + u * blockDim.x;
                    ia[u] = t < half ? (((t & ~(j - 1)) << 1) | (t & (j - 1))) : -1;
                    if (ia[u] >= 0) { a[u] = keys[ia[u]]; b[u] = keys[ia[u] | j]; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (ia[u] < 0) continue;
                    const bool up = (ia[u] & k) == 0;
                    if ((a[u] > b[u]) == up) { keys[ia[u]] = b[u]; keys[ia[u] | j] = a[u]; }
                }
            }
        }
    }
    __syncthreads();
}

// k largest scores of a row, descending (ties: lower index first).  key = (~sortable(score) << 32) | index is unique, so the
// k-th smallest key K* is found EXACTLY by a radix select (six 8-bit histogram passes over the row in LDS: the four score bytes
// and the two live index bytes), the k keys <= K* are compacted, and only those are sorted (a 1024-key network, 55 stages,
// instead of the 8192-key one, 91 stages at 8x the traffic: the full sort was LDS-bandwidth bound at ~80 us).
// KP2 = next_pow2(k) ; when KP2 >= npow2 the row is simply sorted whole.
__global__ __launch_bounds__(1024) void topk_rows_kernel(const float* __restrict__ scores, long* __restrict__ idx_out,
                                                         int S, int k, int npow2, int kp2)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];      // [npow2] row keys, then [kp2] candidates
    __shared__ int hist[256];
    __shared__ int s_rem, s_digit, s_cnt;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* row = scores + (long)b * S;
    for (int i = threadIdx.x; i < npow2; i += blockDim.x)
        keys[i] = i < S ? (((unsigned long long)(~f32_sortable(row[i]))) << 32) | (unsigned)i : ~0ull;
    if (kp2 >= npow2) {
        bitonic_sort_u64(keys, npow2);
        for (int i = threadIdx.x; i < k; i += blockDim.x) idx_out[(long)b * k + i] = (long)(keys[i] & 0xffffffffull);
        return;
    }
    unsigned long long* cand = keys + npow2;
    unsigned long long pref = 0ull, mask = 0ull;
    int rem = k;
    for (int pass = 0; pass < 6; ++pass) {
        const int byte = pass < 4 ? 7 - pass : 5 - pass;                          // 7, 6, 5, 4, then 1, 0 (index bytes 3, 2 are zero)
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
            const unsigned long long key = keys[i];
            if ((key & mask) == pref) atomicAdd(&hist[(int)(key >> (8 * byte)) & 255], 1);
        }
        __syncthreads();
        if (wave == 0) {                                                          // which digit holds the rem-th smallest key?
            const int4 h = reinterpret_cast<const int4*>(hist)[lane];
            const int loc = h.x + h.y + h.z + h.w;
            int inc = loc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
            const int exc = inc - loc;
            if (exc < rem && rem <= inc) {                                        // exactly one lane
                int r = rem - exc, d = 0;
                if (r > h.x) { r -= h.x; d = 1; if (r > h.y) { r -= h.y; d = 2; if (r > h.z) { r -= h.z; d = 3; } } }
                s_rem = r; s_digit = 4 * lane + d;
            }
        }
        __syncthreads();
        rem = s_rem;
        pref |= (unsigned long long)s_digit << (8 * byte);
        mask |= 0xffull << (8 * byte);
        if (byte == 4) mask |= 0xffff0000ull;                                     // index < 2^16: those bytes match as zero
    }
    // pref == K*.  Compact the k keys <= K* (any order: they are sorted next), one LDS atomic per wave
    if (threadIdx.x == 0) s_cnt = 0;
    for (int i = k + threadIdx.x; i < kp2; i += blockDim.x) cand[i] = ~0ull;
    __syncthreads();
    for (int i0 = 0; i0 < npow2; i0 += blockDim.x) {
        const int i = i0 + threadIdx.x;
        const unsigned long long key = i < npow2 ? keys[i] : ~0ull;
        const bool sel = key <= pref;
        const unsigned long long m = __ballot(sel);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&s_cnt, __popcll(m));
        base = __shfl(base, 0, 64);
        if (sel) cand[base + __popcll(m & ((1ull << lane) - 1ull))] = key;
    }
    bitonic_sort_u64(cand, kp2);
    for (int i = threadIdx.x; i < k; i += blockDim.x) idx_out[(long)b * k + i] = (long)(cand[i] & 0xffffffffull);
}

// k largest of a LONG row that stays in global memory (PostProcess: top `num_select` of the nq x C flattened class scores,
// models/dino/dino.py:1000-1006; 149400 per line for the Latin model, 6.6 M for the Chinese one).  Same exact radix select as
// topk_rows on key = (~sortable(x) << 32) | index, but every pass re-reads the row (L2-resident) instead of an LDS copy: four score
// bytes, then as many index bytes as n needs.  The k survivors are compacted into LDS, sorted, and written as (value, index) with
// value = sigmoid(x) when `apply_sigmoid` (the selection runs on the logits: sigmoid is monotone, so the result is a valid top-k of
// the probabilities, descending, with ties ordered by logit and then by lower index).  k <= 8192 (the survivors' sort lives in LDS).
// Also the large-S path of dtlr_topk_rows (rows too long for an LDS copy), with values == nullptr.
__global__ __launch_bounds__(1024) void topk_flat_kernel(const float* __restrict__ x, float* __restrict__ values, long* __restrict__ idx_out,
                                                         long n, int k, int kp2, int index_bytes, int apply_sigmoid)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long cand[];      // [kp2]
    __shared__ int hist[256];
    __shared__ int s_rem, s_digit, s_cnt;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* row = x + (long)b * n;
    unsigned long long pref = 0ull, mask = 0ull;
    int rem = k;
    const int npass = 4 + index_bytes;
    for (int pass = 0; pass < npass; ++pass) {
        const int byte = pass < 4 ? 7 - pass : index_bytes - 1 - (pass - 4);
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        for (long i = threadIdx.x; i < n; i += blockDim.x) {
            const unsigned long long key = (((unsigned long long)(~f32_sortable(row[i]))) << 32) | (unsigned long long)i;
            if ((key & mask) == pref) atomicAdd(&hist[(int)(key >> (8 * byte)) & 255], 1);
        }
        __syncthreads();
        if (wave == 0) {
            const int4 h = reinterpret_cast<const int4*>(hist)[lane];
            const int loc = h.x + h.y + h.z + h.w;
            int inc = loc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
            const int exc = inc - loc;
            if (exc < rem && rem <= inc) {
                int r = rem - exc, d = 0;
                if (r > h.x) { r -= h.x; d = 1; if (r > h.y) { r -= h.y; d = 2; if (r > h.z) { r -= h.z; d = 3; } } }
                s_rem = r; s_digit = 4 * lane + d;
            }
        }
        __syncthreads();
        rem = s_rem;
        pref |= (unsigned long long)s_digit << (8 * byte);
        mask |= 0xffull << (8 * byte);
        if (byte == 4 && index_bytes < 4) mask |= 0xffffffffull & ~((1ull << (8 * index_bytes)) - 1ull);   // unused high index bytes are zero
    }
    if (threadIdx.x == 0) s_cnt = 0;
    for (int i = k + threadIdx.x; i < kp2; i += blockDim.x) cand[i] = ~0ull;
    __syncthreads();
    for (long i0 = 0; i0 < n; i0 += blockDim.x) {
        const long i = i0 + threadIdx.x;
        const unsigned long long key = i < n ? (((unsigned long long)(~f32_sortable(row[i]))) << 32) | (unsigned long long)i : ~0ull;
        const bool sel = key <= pref;
        const unsigned long long m = __ballot(sel);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&s_cnt, __popcll(m));
        base = __shfl(base, 0, 64);
        if (sel) cand[base + __popcll(m & ((1ull << lane) - 1ull))] = key;
    }
    bitonic_sort_u64(cand, kp2);
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const long id = (long)(cand[i] & 0xffffffffull);
        const float v = row[id];
        idx_out[(long)b * k + i] = id;
        if (values) values[(long)b * k + i] = apply_sigmoid ? 1.f / (1.f + expf(-v)) : v;
    }
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

// Step 1 of the blank decoder, chip-wide: the label of every query (class index, or -1 = blank), 16 lanes per query.
// logits [B*nq, C] fp32 -> raw [B*nq] int32.  The 16-lane reductions are DPP row operations (xor 1, xor 2, half-mirror, mirror):
// no LDS traffic, no 64-lane butterflies.
__global__ __launch_bounds__(256) void query_label_kernel(const float* __restrict__ logits, int* __restrict__ raw, long nrows, int C, float eps)
{
    const int l16 = threadIdx.x & 15;
    const long q = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = q < nrows;
    const float* lr = logits + (live ? q : 0) * C;
    float sum = 0.f, best = -1.f, nonfin = 0.f;
    int arg = 0x7fffffff;
    for (int c0 = 0; c0 < C; c0 += 64) {                            // four classes per lane in flight
        float x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int c = c0 + 16 * u + l16; x[u] = (live && c < C) ? lr[c] : -INFINITY; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + 16 * u + l16;
            const float pq = 1.f / (1.f + expf(-x[u]));             // sigmoid(-inf) = 0 for the padding
            sum += pq;
            if (live && c < C) nonfin += x[u] - x[u];               // 0 for a finite logit, NaN for +-inf / NaN (sigmoid(+-inf) is finite: `sum` alone misses it)
            if (c < C && pq > best) { best = pq; arg = c; }         // ascending c: the first maximum of the lane
        }
    }
#define QL_STEP(CTRL)                                                                          \
    {                                                                                          \
        sum += dpp_f<CTRL>(sum);                                                               \
        nonfin += dpp_f<CTRL>(nonfin);                                                         \
        const float ob = dpp_f<CTRL>(best);                                                    \
        const int oa = dpp_i<CTRL>(arg);                                                       \
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }                    \
    }
    QL_STEP(0xB1) QL_STEP(0x4E) QL_STEP(0x141) QL_STEP(0x140)       // quad xor 1, quad xor 2, row_half_mirror, row_mirror
#undef QL_STEP
    // blank channel (dino.py:489-502): sum < 1-eps -> blank = 1-sum ; else blank = eps, p <- (1-eps) p / sum
    float blank, top;
    if (sum < 1.f - eps) { blank = 1.f - sum; top = best; }
    else { blank = eps; top = (1.f - eps) * best / sum; }
    // A query with ANY non-finite logit (NaN, +inf or -inf) gets the label -2 and its LINE the length -1 (decode_blank_kernel): the
    // fp16 / split engines turn an activation beyond 65504 into inf and from there into NaN everywhere (DTLREngine checks the range on
    // its first forward only) -- the caller sees it in the record, without a host synchronisation on the step, instead of reading
    // garbage labels.  `nonfin` sums x - x over the query's logits (0 for finite x, NaN otherwise; round 5 tested `sum` only, which a
    // +-inf logit leaves finite: sigmoid(+inf) = 1, sigmoid(-inf) = 0).
    const bool bad = !(nonfin == 0.f) || !(sum - sum == 0.f);
    if (live && l16 == 0) raw[q] = bad ? -2 : ((blank >= top) ? -1 : arg);       // argmax over [blank | classes]: blank wins ties
}

// Step 2, one workgroup per line: sort the queries by box cx, read their labels in that order, drop the blanks.
// boxes [B,nq,4] fp32 ; labels [B,nq] int32 holds the per-query labels of step 1 on entry and the left-packed, -1 padded
// reading-order labels on exit ; lengths [B]
__global__ __launch_bounds__(1024) void decode_blank_kernel(const float* __restrict__ boxes,
                                                            int* __restrict__ labels, int* __restrict__ lengths,
                                                            int nq, int npow2)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];      // [npow2] then int lab[npow2], int rawl[npow2]
    int* lab = reinterpret_cast<int*>(keys + npow2);
    int* rawl = lab + npow2;
    __shared__ int wave_tot[16];
    __shared__ int s_bad;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
        keys[i] = i < nq ? (((unsigned long long)f32_sortable(boxes[((long)b * nq + i) * 4])) << 32) | (unsigned)i : ~0ull;
        rawl[i] = i < nq ? labels[(long)b * nq + i] : -1;
        if (rawl[i] == -2) s_bad = 1;                                          // a query with non-finite logits (query_label_kernel): benign race, same value
    }
    bitonic_sort_u64(keys, npow2);                                             // ascending cx, ties: lower index first
    for (int p = threadIdx.x; p < nq; p += blockDim.x) lab[p] = rawl[(int)(keys[p] & 0xffffffffull)];
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
    if (threadIdx.x == 0) lengths[b] = s_bad ? -1 : running;                     // -1: this line's logits were not finite
}

// ---- CTC-style emissions of the n-gram re-scoring path (ngram/prediction_helpers.py:5-46, get_new_pred_logits; with scale = 1 the
// blank construction of SetCriterion.loss_CTC, models/dino/dino.py:466-502) -------------------------------------------------------
// out[b, r, :] = [blank | classes] of the r-th query of line b IN READING ORDER (sorted by box cx): p = scale * sigmoid(logit),
// s = sum_c p; s < 1 - eps: blank = 1 - s, classes = p; else blank = eps, classes = (1 - eps) p / s.
// Step 1 (query_sum_kernel above): s / scale for every query, chip-wide.  Step 2, one workgroup per line: the reading order
// (bitonic sort of (cx, index) keys in LDS, the decoders' own sort) -> order[b, r].  Step 3, chip-wide: one wave per output row.
__global__ __launch_bounds__(256) void query_sum_kernel(const float* __restrict__ logits, float* __restrict__ sums, long nrows, int C);
__global__ __launch_bounds__(1024) void reading_order_kernel(const float* __restrict__ boxes, int* __restrict__ order, int nq, int npow2)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < npow2; i += blockDim.x)
        keys[i] = i < nq ? (((unsigned long long)f32_sortable(boxes[((long)b * nq + i) * 4])) << 32) | (unsigned)i : ~0ull;
    bitonic_sort_u64(keys, npow2);                                             // ascending cx, ties: lower index first
    for (int p = threadIdx.x; p < nq; p += blockDim.x) order[(long)b * nq + p] = (int)(keys[p] & 0xffffffffull);
}

__global__ __launch_bounds__(256) void blank_emissions_kernel(const float* __restrict__ logits, const float* __restrict__ sums,
                                                               const int* __restrict__ order, float* __restrict__ out,
                                                               long nrows, int nq, int C, float scale, float eps)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);               // output row = (line, rank)
    if (row >= nrows) return;
    const long b = row / nq;
    const long q = b * nq + order[row];
    const float s = sums[q] * scale;
    const bool low = s < 1.f - eps;
    const float mul = low ? scale : (1.f - eps) * scale / s;
    const float* lr = logits + q * C;
    float* o = out + row * (long)(C + 1);
    if (lane == 0) o[0] = low ? 1.f - s : eps;
    for (int c = lane; c < C; c += 64) o[1 + c] = mul / (1.f + expf(-lr[c]));
}

// ---- evaluation-time CTC loss value (models/dino/dino.py:457-551, SetCriterion.loss_CTC) ------------------------------------
// Step 1, chip-wide: sum over classes of sigmoid(logit) for every query (16 lanes per query, DPP reductions).
__global__ __launch_bounds__(256) void query_sum_kernel(const float* __restrict__ logits, float* __restrict__ sums, long nrows, int C)
{
    const int l16 = threadIdx.x & 15;
    const long q = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = q < nrows;
    const float* lr = logits + (live ? q : 0) * C;
    float sum = 0.f;
    for (int c0 = 0; c0 < C; c0 += 64) {
        float x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int c = c0 + 16 * u + l16; x[u] = (live && c < C) ? lr[c] : -INFINITY; }
#pragma unroll
        for (int u = 0; u < 4; ++u) sum += 1.f / (1.f + expf(-x[u]));
    }
    sum += dpp_f<0xB1>(sum); sum += dpp_f<0x4E>(sum); sum += dpp_f<0x141>(sum); sum += dpp_f<0x140>(sum);
    if (live && l16 == 0) sums[q] = sum;
}

// Step 2, one workgroup per line, one thread per state of the blank-extended label sequence l' (S = 2 L + 1 <= blockDim):
// the CTC forward recursion over T = 2 nq steps -- step 2 i is query i of the reading order (sorted by box cx) with the
// blank-channel probabilities of dino.py:474-502, step 2 i + 1 the filler row [1, filler, filler, ...] of :505-519 -- in log
// space with the exact update of torch's CTCLoss (max-shifted three-term log-sum-exp, -inf handling).  Alphas are double
// buffered in LDS, one barrier per step; the logit a state needs at query i is prefetched CTC_PF queries ahead (a dependent
// global load per step would cost an HBM round trip 900 times).
constexpr int CTC_PF = 8;

__global__ __launch_bounds__(1024) void ctc_interleaved_kernel(const float* __restrict__ logits, const float* __restrict__ boxes,
                                                               const float* __restrict__ sums, const int* __restrict__ targets,
                                                               const int* __restrict__ target_lengths, float* __restrict__ nll,
                                                               int nq, int C, int Lmax, float eps, float filler, int npow2)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];      // [npow2] | float ssum[npow2] | float alpha[2][blockDim + 2]
    float* ssum = reinterpret_cast<float*>(keys + npow2);
    float* alpha = ssum + npow2;
    const int b = blockIdx.x, s = threadIdx.x, AP = blockDim.x + 2;
    const int L = target_lengths[b], S = 2 * L + 1;
    for (int i = threadIdx.x; i < npow2; i += blockDim.x)
        keys[i] = i < nq ? (((unsigned long long)f32_sortable(boxes[((long)b * nq + i) * 4])) << 32) | (unsigned)i : ~0ull;
    bitonic_sort_u64(keys, npow2);                                             // ascending cx, ties: lower index first
    for (int i = threadIdx.x; i < nq; i += blockDim.x) ssum[i] = sums[(long)b * nq + (int)(keys[i] & 0xffffffffull)];
    // this thread's state: label (0 = blank), and whether the skip transition s-2 -> s exists
    const bool live = s < S;
    const int lab = (live && (s & 1)) ? targets[(long)b * Lmax + (s >> 1)] : 0;
    const bool skip = live && (s & 1) && s >= 3 && lab != targets[(long)b * Lmax + (s >> 1) - 1];
    const float one_m_eps = (float)(1.0 - (double)eps), thr = one_m_eps;
    const float lfill = lab == 0 ? 0.f : logf(filler);
    const float* lrow = logits + (long)b * nq * C + (lab > 0 ? lab - 1 : 0);
    __syncthreads();
    // alpha buffers carry two -inf guard slots in front (states -2, -1)
    float* a0 = alpha + 2;
    float* a1 = alpha + AP + 2;
    if (threadIdx.x < 2) { alpha[threadIdx.x] = -INFINITY; alpha[AP + threadIdx.x] = -INFINITY; }

    float pf[CTC_PF];                                                           // logits of queries i0 .. i0 + CTC_PF - 1 for this state
#pragma unroll
    for (int u = 0; u < CTC_PF; ++u) pf[u] = (lab > 0 && u < nq) ? lrow[(long)(int)(keys[u] & 0xffffffffull) * C] : 0.f;

    auto logp = [&](float x, float sum) -> float {                              // log of the blank-augmented probability
        float p;
        if (sum < thr) p = lab == 0 ? 1.f - sum : 1.f / (1.f + expf(-x));
        else p = lab == 0 ? eps : one_m_eps * (1.f / (1.f + expf(-x))) / sum;
        return logf(p);
    };
    auto step = [&](const float* prev, float* cur, float lp) {
        if (live) {
            const float la1 = prev[s], la2 = prev[s - 1], la3 = skip ? prev[s - 2] : -INFINITY;
            float m = fmaxf(la1, fmaxf(la2, la3));
            if (m == -INFINITY) m = 0.f;
            cur[s] = logf(expf(la1 - m) + expf(la2 - m) + expf(la3 - m)) + m + lp;
        }
        __syncthreads();
    };
    for (int i0 = 0; i0 < nq; i0 += CTC_PF) {
        float nx[CTC_PF];
#pragma unroll
        for (int u = 0; u < CTC_PF; ++u) {
            const int i = i0 + CTC_PF + u;
            nx[u] = (lab > 0 && i < nq) ? lrow[(long)(int)(keys[i] & 0xffffffffull) * C] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < CTC_PF; ++u) {
            const int i = i0 + u;
            if (i < nq) {                                                       // block-uniform
                const float lp = logp(pf[u], ssum[i]);
                if (i == 0) {                                                   // t = 0: only states 0 and 1 are reachable
                    if (live) a0[s] = s < 2 ? lp : -INFINITY;
                    __syncthreads();
                } else step(a1, a0, lp);                                        // t = 2 i   : a1 -> a0
                step(a0, a1, lfill);                                            // t = 2 i + 1: a0 -> a1
            }
        }
#pragma unroll
        for (int u = 0; u < CTC_PF; ++u) pf[u] = nx[u];
    }
    if (threadIdx.x == 0) {                                                     // after t = T - 1 the alphas are in a1
        const float l1 = a1[S - 1], l2 = L > 0 ? a1[S - 2] : -INFINITY;
        float m = fmaxf(l1, l2);
        if (m == -INFINITY) m = 0.f;
        const float v = -(logf(expf(l1 - m) + expf(l2 - m)) + m);
        nll[b] = isinf(v) ? 0.f : v;                                            // zero_infinity=True
    }
}

// ---- greedy non-maximum suppression, one workgroup per image (torchvision.ops.nms semantics as models/dino/dino.py:1029-1033 uses
// it: descending score, a box is dropped when its IoU with an already kept box is > threshold; equal scores keep the lower
// index first).  Stable score sort (bitonic, LDS) -> boxes in sorted order -> the upper-triangular suppression bit-matrix
// (n x n/64 words, LDS, all threads) -> ONE wave walks the rows in order, OR-ing a kept row's word into its lane's "removed"
// word (lane w owns word w; no barriers) -> kept original indices, in descending score order.  n <= 1024.
__global__ __launch_bounds__(1024) void nms_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, float thr,
                                                   long* __restrict__ keep, int* __restrict__ counts, int n, int npow2)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long nms_lds[];   // matrix [n][W] (its head doubles as the sort keys) | float4 sb[npow2] | int order[npow2]
    const int W = (n + 63) >> 6;
    unsigned long long* mat = nms_lds;
    float4* sb = reinterpret_cast<float4*>(nms_lds + (size_t)max(n * W, npow2));
    int* order = reinterpret_cast<int*>(sb + npow2);
    const int b = blockIdx.x;
    const float* bx = boxes + (long)b * n * 4;
    unsigned long long* keys = nms_lds;
    for (int i = threadIdx.x; i < npow2; i += blockDim.x)
        keys[i] = i < n ? (((unsigned long long)(~f32_sortable(scores[(long)b * n + i]))) << 32) | (unsigned)i : ~0ull;
    bitonic_sort_u64(keys, npow2);                                             // descending score, ties: lower index first
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int o = (int)(keys[i] & 0xffffffffull);
        order[i] = o;
        sb[i] = *reinterpret_cast<const float4*>(bx + (long)o * 4);
    }
    __syncthreads();                                                           // keys are dead from here: the matrix overwrites them
    for (int t = threadIdx.x; t < n * W; t += blockDim.x) {
        const int i = t / W, w = t - i * W;
        unsigned long long bits = 0ull;
        if (64 * w + 63 > i) {                                                 // only columns j > i
            const float4 a = sb[i];
            const float area_a = (a.z - a.x) * (a.w - a.y);
            for (int jj = 0; jj < 64; ++jj) {
                const int j = 64 * w + jj;
                if (j > i && j < n) {
                    const float4 c = sb[j];
                    const float iw = fmaxf(fminf(a.z, c.z) - fmaxf(a.x, c.x), 0.f), ih = fmaxf(fminf(a.w, c.w) - fmaxf(a.y, c.y), 0.f);
                    const float inter = iw * ih;
                    const float area_c = (c.z - c.x) * (c.w - c.y);
                    if (inter / (area_a + area_c - inter) > thr) bits |= 1ull << jj;
                }
            }
        }
        mat[t] = bits;
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                                    // the sequential sweep, one wave, no barriers
        const int lane = threadIdx.x;
        unsigned long long removed = 0ull;                                     // lane w: bits of boxes 64 w .. 64 w + 63
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            const unsigned long long r = __shfl(removed, i >> 6, 64);
            if (!((r >> (i & 63)) & 1ull)) {                                   // wave-uniform
                if (lane == 0) keep[(long)b * n + cnt] = order[i];
                ++cnt;
                if (lane < W) removed |= mat[i * W + lane];
            }
        }
        for (int i = cnt + lane; i < n; i += 64) keep[(long)b * n + i] = -1;
        if (lane == 0) counts[b] = cnt;
    }
}

static inline int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

}  // namespace dtlr

using namespace dtlr;

extern "C" int dtlr_topk_rows(const float* scores, long* idx_out, int B, int S, int k, void* stream)
{
    clear_stale_error();
    if (!scores || !idx_out) return DTLR_EINVAL;
    if (B <= 0 || S <= 0 || k <= 0 || k > S) return DTLR_EINVAL;
    const int np = next_pow2(S), kp = next_pow2(k);
    const size_t lds = (size_t)np * 8 + (kp < np ? (size_t)kp * 8 : 0);
    if (np > 65536 || lds > 156 * 1024) {
        // Rows too long for an LDS copy (tall canvases: S > 16384 tokens): the same exact radix select with the row left in
        // global memory (topk_flat_kernel, values not wanted) -- identical key order, so identical indices.
        if (kp > 8192) return DTLR_ESHAPE;
        int index_bytes = 1;
        while (index_bytes < 4 && ((long)S - 1) >> (8 * index_bytes)) ++index_bytes;
        const size_t fl = (size_t)kp * 8;
        if (fl > 48 * 1024) { (void)hipFuncSetAttribute((const void*)topk_flat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl); (void)hipGetLastError(); }
        hipLaunchKernelGGL(topk_flat_kernel, dim3(B), dim3(1024), fl, (hipStream_t)stream, scores, (float*)nullptr, idx_out, (long)S, k, kp, index_bytes, 0);
        return check_launch();
    }
    (void)hipGetLastError();                                   // do not inherit a stale error from an earlier API call
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)topk_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    hipLaunchKernelGGL(topk_rows_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, scores, idx_out, S, k, np, kp);
    return check_launch();
}

extern "C" int dtlr_decode_blank(const float* logits, const float* boxes, int* labels, int* lengths,
                                 int B, int nq, int C, float eps, void* stream)
{
    clear_stale_error();
    if (!logits || !boxes || !labels || !lengths) return DTLR_EINVAL;
    if (B <= 0 || nq <= 0 || C <= 0) return DTLR_EINVAL;
    const int np = next_pow2(nq);
    const size_t lds = (size_t)np * 16;
    if (lds > 150 * 1024) return DTLR_ESHAPE;
    (void)hipGetLastError();
    if (lds > 60 * 1024) (void)hipFuncSetAttribute((const void*)decode_blank_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    const long nrows = (long)B * nq;
    hipLaunchKernelGGL(query_label_kernel, dim3((unsigned)((nrows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, logits, labels, nrows, C, eps);
    hipLaunchKernelGGL(decode_blank_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, boxes, labels, lengths, nq, np);
    return check_launch();
}

extern "C" int dtlr_ctc_loss_interleaved(const float* logits, const float* boxes, const int* targets, const int* target_lengths,
                                         float* nll, float* workspace, int B, int nq, int C, int Lmax, int max_target_length,
                                         float eps, float filler, void* stream)
{
    clear_stale_error();
    if (!logits || !boxes || !target_lengths || !nll || !workspace) return DTLR_EINVAL;
    if (B <= 0 || nq <= 0 || C <= 0 || Lmax < 0 || max_target_length < 0 || max_target_length > Lmax) return DTLR_EINVAL;
    if (Lmax > 0 && !targets) return DTLR_EINVAL;
    const int S = 2 * max_target_length + 1;
    if (S > 1024) return DTLR_ESHAPE;                          // one thread per state
    const int threads = S <= 64 ? 64 : ((S + 63) / 64) * 64;
    const int np = next_pow2(nq);
    const size_t lds = (size_t)np * 12 + (size_t)2 * (threads + 2) * 4;
    if (lds > 150 * 1024) return DTLR_ESHAPE;
    const long nrows = (long)B * nq;
    if (lds > 60 * 1024) (void)hipFuncSetAttribute((const void*)ctc_interleaved_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    hipLaunchKernelGGL(query_sum_kernel, dim3((unsigned)((nrows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, logits, workspace, nrows, C);
    hipLaunchKernelGGL(ctc_interleaved_kernel, dim3(B), dim3(threads), lds, (hipStream_t)stream, logits, boxes, workspace, targets,
                       target_lengths, nll, nq, C, Lmax, eps, filler, np);
    return check_launch();
}

extern "C" int dtlr_blank_emissions(const float* logits, const float* boxes, float* out, float* workspace,
                                    int B, int nq, int C, float scale, float eps, void* stream)
{
    clear_stale_error();
    if (!logits || !boxes || !out || !workspace) return DTLR_EINVAL;
    if (B <= 0 || nq <= 0 || C <= 0) return DTLR_EINVAL;
    const int np = next_pow2(nq);
    const size_t lds = (size_t)np * 8;
    if (lds > 150 * 1024) return DTLR_ESHAPE;
    if (lds > 60 * 1024) (void)hipFuncSetAttribute((const void*)reading_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    const long nrows = (long)B * nq;
    float* sums = workspace;                                  // [B * nq] fp32, then [B * nq] int32 (dtlr_blank_emissions_workspace_bytes)
    int* order = reinterpret_cast<int*>(workspace + nrows);
    hipLaunchKernelGGL(query_sum_kernel, dim3((unsigned)((nrows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, logits, sums, nrows, C);
    hipLaunchKernelGGL(reading_order_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, boxes, order, nq, np);
    hipLaunchKernelGGL(blank_emissions_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, sums, order, out,
                       nrows, nq, C, scale, eps);
    return check_launch();
}

extern "C" long dtlr_blank_emissions_workspace_bytes(int B, int nq) { return (long)B * nq * 8; }


extern "C" int dtlr_nms(const float* boxes, const float* scores, float iou_threshold, long* keep, int* counts, int B, int n, void* stream)
{
    clear_stale_error();
    if (!boxes || !scores || !keep || !counts) return DTLR_EINVAL;
    if (B <= 0 || n <= 0) return DTLR_EINVAL;
    if (n > 1024) return DTLR_ESHAPE;                          // the bit-matrix lives in LDS
    const int np = next_pow2(n), W = (n + 63) / 64;
    const size_t words = (size_t)(n * W > np ? n * W : np);
    const size_t lds = words * 8 + (size_t)np * 16 + (size_t)np * 4;
    if (lds > 160 * 1024) return DTLR_ESHAPE;
    (void)hipFuncSetAttribute((const void*)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, boxes, scores, iou_threshold, keep, counts, n, np);
    return check_launch();
}

extern "C" int dtlr_topk_flat(const float* x, float* values, long* idx_out, int B, long n, int k, int apply_sigmoid, void* stream)
{
    clear_stale_error();
    if (!x || !values || !idx_out) return DTLR_EINVAL;
    if (B <= 0 || n <= 0 || k <= 0 || (long)k > n) return DTLR_EINVAL;
    if (k > 8192 || n > 0xffffffffl) return DTLR_ESHAPE;
    int index_bytes = 1;
    while (index_bytes < 4 && (n - 1) >> (8 * index_bytes)) ++index_bytes;
    const size_t fl = (size_t)next_pow2(k) * 8;
    if (fl > 48 * 1024) { (void)hipFuncSetAttribute((const void*)topk_flat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl); (void)hipGetLastError(); }
    hipLaunchKernelGGL(topk_flat_kernel, dim3(B), dim3(1024), fl, (hipStream_t)stream, x, values, idx_out, n, k, next_pow2(k), index_bytes, apply_sigmoid);
    return check_launch();
}
