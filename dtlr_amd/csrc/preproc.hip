// Eval-time preprocessing of text-line images on the device:
//     PIL RGB uint8 [h, w, 3]  ->  resize (short side `size`, long side capped)  ->  /255  ->  (x - mean) / std  ->  zero-padded
//     batch canvas [B, 3, Hc, Wc] fp32 + padding mask [B, Hc, Wc]
// == datasets/transforms.py:78-109 (`resize` -> torchvision F.resize on a PIL image == Image.resize(BILINEAR)), :247-249 (ToTensor),
// :552-559 (Normalize), composed by datasets/IAM.py:110-112, 225-230, and the collate of util/misc.py:375-397.
//
// The resize is Pillow's fixed-point two-pass resample (third-party arithmetic, restated in oracle/dtlr_oracle.py and pinned there
// against Pillow itself): triangle filter of support max(scale, 1), per-output normalised weights rounded to 22 fractional bits,
// horizontal pass rounded to uint8, then the vertical pass.  The coefficient arithmetic is IEEE double with contraction OFF so the
// (int) truncations land where the C code's do; bit-exact output is the test bar.
//
// One thread per canvas pixel.  A thread's horizontal weights live in its own LDS column (computed once, used for every source
// row it touches); the few vertical weights are recomputed per thread (a handful of double operations).  The source is uint8:
// a 128 x 2048 line is 0.8 MB against the 1.3 MB fp32 tensor it produces, so the kernel is bound by its fp32 canvas writes.
#include "dtlr_common.h"

namespace dtlr {

constexpr int PP_MAXT = 24;                 // horizontal taps: ceil(scale) * 2 + 1 <= 24  <=>  down-scaling up to 11x
constexpr int PP_BITS = 32 - 8 - 2;         // Pillow PRECISION_BITS

__device__ __forceinline__ void pp_bounds(int in_size, int out_size, int xx, double& center, double& ss, int& first, int& count) {
#pragma clang fp contract(off)
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    ss = 1.0 / filterscale;
    center = 0.0 + ((double)xx + 0.5) * scale;
    first = (int)(center - support + 0.5);
    if (first < 0) first = 0;
    int last = (int)(center + support + 0.5);
    if (last > in_size) last = in_size;
    count = last - first;
}

__device__ __forceinline__ double pp_weight(int x, int first, double center, double ss) {
#pragma clang fp contract(off)
    double a = ((double)(x + first) - center + 0.5) * ss;
    if (a < 0.0) a = -a;
    return a < 1.0 ? 1.0 - a : 0.0;
}

__device__ __forceinline__ int pp_fixed(double w, double ww) {
#pragma clang fp contract(off)
    if (ww != 0.0) w = w / ww;
    return w < 0.0 ? (int)(-0.5 + w * (double)(1 << PP_BITS)) : (int)(0.5 + w * (double)(1 << PP_BITS));
}

__device__ __forceinline__ int pp_clip8(int v) { v >>= PP_BITS; return v < 0 ? 0 : (v > 255 ? 255 : v); }

struct PPNorm { float mean[3], std[3]; };

// src: the images back to back, each [h, w, 3] uint8 ; offsets [B] bytes ; dims [B][4] = (h, w, oh, ow)
__global__ __launch_bounds__(256) void preprocess_lines_kernel(const unsigned char* __restrict__ src, const long* __restrict__ offsets,
                                                               const int* __restrict__ dims, float* __restrict__ canvas,
                                                               unsigned char* __restrict__ mask, int Hc, int Wc, PPNorm nrm)
{
    __shared__ int kh[PP_MAXT][256];
    const int b = blockIdx.z, yy = blockIdx.y, xx = blockIdx.x * 256 + threadIdx.x;
    if (xx >= Wc) return;
    const int h = dims[b * 4], w = dims[b * 4 + 1], oh = dims[b * 4 + 2], ow = dims[b * 4 + 3];
    const long plane = (long)Hc * Wc;
    float* out = canvas + (long)b * 3 * plane + (long)yy * Wc + xx;
    if (yy >= oh || xx >= ow) {                                     // padding: zeros, mask = True (util/misc.py:388-396)
        out[0] = 0.f; out[plane] = 0.f; out[2 * plane] = 0.f;
        mask[(long)b * plane + (long)yy * Wc + xx] = 1;
        return;
    }
    // horizontal weights of output column xx
    double cx, sx;
    int x0, xn;
    pp_bounds(w, ow, xx, cx, sx, x0, xn);
    {
        double ww = 0.0;
        for (int i = 0; i < xn; ++i) ww += pp_weight(i, x0, cx, sx);
        for (int i = 0; i < xn; ++i) kh[i][threadIdx.x] = pp_fixed(pp_weight(i, x0, cx, sx), ww);
    }
    // vertical weights of output row yy
    double cy, sy;
    int y0, yn;
    pp_bounds(h, oh, yy, cy, sy, y0, yn);
    double wwv = 0.0;
    for (int i = 0; i < yn; ++i) wwv += pp_weight(i, y0, cy, sy);

    const unsigned char* img = src + offsets[b];
    int v0 = 1 << (PP_BITS - 1), v1 = v0, v2 = v0;
    for (int yi = 0; yi < yn; ++yi) {
        const int kv = pp_fixed(pp_weight(yi, y0, cy, sy), wwv);
        const unsigned char* row = img + ((long)(y0 + yi) * w + x0) * 3;
        int a0 = 1 << (PP_BITS - 1), a1 = a0, a2 = a0;
        for (int xi = 0; xi < xn; ++xi) {
            const int k = kh[xi][threadIdx.x];
            a0 += (int)row[3 * xi] * k; a1 += (int)row[3 * xi + 1] * k; a2 += (int)row[3 * xi + 2] * k;
        }
        v0 += pp_clip8(a0) * kv; v1 += pp_clip8(a1) * kv; v2 += pp_clip8(a2) * kv;     // the horizontal pass is stored as uint8
    }
    // ToTensor (/255) and Normalize, fp32, true divisions
    out[0] = ((float)pp_clip8(v0) / 255.0f - nrm.mean[0]) / nrm.std[0];
    out[plane] = ((float)pp_clip8(v1) / 255.0f - nrm.mean[1]) / nrm.std[1];
    out[2 * plane] = ((float)pp_clip8(v2) / 255.0f - nrm.mean[2]) / nrm.std[2];
    mask[(long)b * plane + (long)yy * Wc + xx] = 0;
}

}  // namespace dtlr

using namespace dtlr;

extern "C" int dtlr_preprocess_lines(const unsigned char* src, const long* offsets, const int* dims, int B, int Hc, int Wc,
                                     float max_downscale, const float* mean3, const float* std3,
                                     float* canvas, unsigned char* mask, void* stream)
{
    clear_stale_error();
    if (!src || !offsets || !dims || !canvas || !mask || !mean3 || !std3) return DTLR_EINVAL;
    if (B <= 0 || Hc <= 0 || Wc <= 0) return DTLR_EINVAL;
    // horizontal taps = ceil(max(scale, 1)) * 2 + 1 must fit the per-thread LDS column
    if (!(max_downscale >= 0.f) || (int)ceilf(fmaxf(max_downscale, 1.0f)) * 2 + 1 > PP_MAXT) return DTLR_ESHAPE;
    if (Hc > 65535 || B > 65535) return DTLR_ESHAPE;
    PPNorm n;
    for (int c = 0; c < 3; ++c) { n.mean[c] = mean3[c]; n.std[c] = std3[c]; }
    const dim3 grid((Wc + 255) / 256, Hc, B);
    hipLaunchKernelGGL(preprocess_lines_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, offsets, dims, canvas, mask, Hc, Wc, n);
    return check_launch();
}
