/* CPU ORACLE (plain C) for the multi-scale deformable attention forward.  TEST INFRASTRUCTURE ONLY:
 * linked by nothing under dtlr_amd/; loaded through ctypes by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg.
 *
 * Restates the arithmetic of the reference CUDA kernel, one output element at a time:
 *   ms_deformable_im2col_gpu_kernel   models/dino/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299
 *   ms_deform_attn_im2col_bilinear    models/dino/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-84
 * (the reference's CPU branch only throws, src/cpu/ms_deform_attn_cpu.cpp:17-40, and the CUDA
 * sources cannot be built here -- no nvcc, THC headers gone -- so there is no oracle/_ref build;
 * this restatement is pinned instead against ms_deform_attn_core_pytorch through the committed
 * fixtures tests/golden/g1_msda.npz, see tests/test_oracle_golden.py).
 *
 * Layouts (all contiguous, as ms_deform_attn_cuda.cu:28-38 asserts):
 *   value [N,S,M,D]  shapes int64 [L,2] (H,W)  lsi int64 [L]  loc [N,Lq,M,L,P,2] (x,y)
 *   attn [N,Lq,M,L,P]  out [N,Lq,M*D]
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define DEFINE_MSDA(NAME, T, FLOOR)                                                                  \
static T NAME##_bilinear(const T *data, int H, int W, int M, int D, T h, T w, int m, int c)         \
{                                                                                                    \
    const int h_low = (int)FLOOR(h), w_low = (int)FLOOR(w);                                          \
    const int h_high = h_low + 1, w_high = w_low + 1;                                                \
    const T lh = h - (T)h_low, lw = w - (T)w_low, hh = 1 - lh, hw = 1 - lw;                          \
    const int w_stride = M * D, h_stride = W * w_stride;                                             \
    const int base = m * D + c;                                                                      \
    T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                                                \
    if (h_low >= 0 && w_low >= 0) v1 = data[h_low * h_stride + w_low * w_stride + base];             \
    if (h_low >= 0 && w_high <= W - 1) v2 = data[h_low * h_stride + w_high * w_stride + base];       \
    if (h_high <= H - 1 && w_low >= 0) v3 = data[h_high * h_stride + w_low * w_stride + base];       \
    if (h_high <= H - 1 && w_high <= W - 1) v4 = data[h_high * h_stride + w_high * w_stride + base]; \
    const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                                  \
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;                                                    \
}                                                                                                    \
int NAME(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc, const T *attn,    \
         int N, int S, int M, int D, int L, int Lq, int P, T *out)                                   \
{                                                                                                    \
    if (!value || !shapes || !lsi || !loc || !attn || !out) return -1;                               \
    for (int b = 0; b < N; ++b)                                                                      \
        for (int q = 0; q < Lq; ++q)                                                                 \
            for (int m = 0; m < M; ++m) {                                                            \
                const size_t si = ((size_t)b * Lq + q) * M + m;                                      \
                for (int c = 0; c < D; ++c) {                                                        \
                    size_t wp = si * L * P, lp = wp * 2;                                             \
                    T col = 0;                                                                       \
                    for (int l = 0; l < L; ++l) {                                                    \
                        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                \
                        const T *v = value + ((size_t)b * S + (size_t)lsi[l]) * M * D;               \
                        for (int p = 0; p < P; ++p) {                                                \
                            const T lw = loc[lp], lh = loc[lp + 1], a = attn[wp];                    \
                            const T h_im = lh * H - (T)0.5, w_im = lw * W - (T)0.5;                  \
                            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)                      \
                                col += NAME##_bilinear(v, H, W, M, D, h_im, w_im, m, c) * a;         \
                            wp += 1; lp += 2;                                                        \
                        }                                                                            \
                    }                                                                                \
                    out[si * D + c] = col;                                                           \
                }                                                                                    \
            }                                                                                        \
    return 0;                                                                                        \
}

DEFINE_MSDA(msda_ref_forward_f32, float, floorf)
DEFINE_MSDA(msda_ref_forward_f64, double, floor)
