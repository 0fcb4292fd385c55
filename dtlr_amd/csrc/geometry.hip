// Everything of the forward that depends only on the padding mask, in ONE launch:
//   per-level masks         F.interpolate(mask[None].float(), size=(h, w)).bool()       backbone.py:103, dino.py:304-307
//   valid ratios            get_valid_ratio                                             deformable_transformer.py:239-246
//   position embedding      PositionEmbeddingSineHW + level_embed                       position_encoding.py:79-108,
//                                                                                       deformable_transformer.py:281-285
//   encoder reference pts   get_reference_points                                        deformable_transformer.py:479-492
//   proposals / keep        gen_encoder_output_proposals (logit, +inf masking, validity) models/dino/utils.py:31-62
// The first version of the engine did this with ~60 ATen launches (cumsum, meshgrid, stack, cat ...) re-run on every padded
// forward.  Here a workgroup owns one ROW of one level of one image:
//   * the valid width/height of all four levels (needed by every reference point) are counted by the whole workgroup;
//   * x_embed (the cumulative count of unpadded pixels along the row) is a ballot/popcount scan with a carry across
//     256-token chunks; y_embed and its column total are short loops over the level's rows (h <= 16 for text lines);
//   * a token's scalars (mask, keep, proposal logits, 4 x 2 reference points) are written by the thread that owns the token;
//   * the 256-channel embedding is written through a small LDS transpose so that a wave stores 1 KiB contiguous: thread =
//     (token, 8-channel group), one sincosf per channel PAIR (channels 2m, 2m+1 are sin / cos of the same angle).
// HBM-bound on the embedding write (S x 256 x e bytes per image); everything else is L2-resident mask bytes.
#include "dtlr_common.h"

namespace dtlr {

struct GeoLevels { int H[4], W[4], start[4]; };

__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
    // ATen nearest (upsample_nearest2d): min(int(floorf(dst * scale)), in - 1), scale = float(in) / out
    const float scale = (float)in / (float)out;
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

template <typename OutT> struct GeoOut;
template <> struct GeoOut<float> {
    static __device__ __forceinline__ void st8(float* p, const float* v) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct GeoOut<uint16_t> {
    static __device__ __forceinline__ void st8(uint16_t* p, const float* v) {
        *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    }
};

// mask [B,H,W] uint8 (1 = padding) ; level_embed [4,256] fp32 ; dim_t [128] fp32 (temperature^(2 (i/2) / 128): y table then x table
// are passed separately) ; outputs: see dtlr_geometry in include/dtlr_hip.h
template <typename OutT>
__global__ __launch_bounds__(256) void geometry_kernel(const uint8_t* __restrict__ mask, int H, int W, GeoLevels lv, int S,
                                                       const float* __restrict__ level_embed, const float* __restrict__ dim_ty,
                                                       const float* __restrict__ dim_tx,
                                                       uint8_t* __restrict__ mask_flat, uint8_t* __restrict__ keep, OutT* __restrict__ pos,
                                                       float* __restrict__ valid_ratios, float* __restrict__ enc_ref,
                                                       float* __restrict__ proposals)
{
    __shared__ int s_cnt[8];
    __shared__ int s_wtot[4];
    __shared__ float s_xe[256], s_ye[256];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t* mb = mask + (long)b * H * W;

    // which (level, row) is this workgroup?
    int l = 0, row = blockIdx.x;
    while (l < 3 && row >= lv.H[l]) { row -= lv.H[l]; ++l; }
    const int h = lv.H[l], w = lv.W[l];

    // ---- valid width (row 0) / height (column 0) of every level ----------------------------------------------------
    if (threadIdx.x < 8) s_cnt[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int hq = lv.H[q], wq = lv.W[q];
        int cw = 0, ch = 0;
        const int sy0 = nearest_src(0, H, hq), sx0 = nearest_src(0, W, wq);
        for (int j = threadIdx.x; j < wq; j += 256) cw += mb[(long)sy0 * W + nearest_src(j, W, wq)] ? 0 : 1;
        for (int i = threadIdx.x; i < hq; i += 256) ch += mb[(long)nearest_src(i, H, hq) * W + sx0] ? 0 : 1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { cw += __shfl_xor(cw, o, 64); ch += __shfl_xor(ch, o, 64); }
        if (lane == 0) { if (cw) atomicAdd(&s_cnt[2 * q], cw); if (ch) atomicAdd(&s_cnt[2 * q + 1], ch); }
    }
    __syncthreads();
    float vrw[4], vrh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        vrw[q] = (float)s_cnt[2 * q] / (float)lv.W[q];          // deformable_transformer.py:243-245: valid / size -> (w, h)
        vrh[q] = (float)s_cnt[2 * q + 1] / (float)lv.H[q];
    }
    if (blockIdx.x == 0 && threadIdx.x < 4) {
        valid_ratios[((long)b * 4 + threadIdx.x) * 2 + 0] = (float)s_cnt[2 * threadIdx.x] / (float)lv.W[threadIdx.x];
        valid_ratios[((long)b * 4 + threadIdx.x) * 2 + 1] = (float)s_cnt[2 * threadIdx.x + 1] / (float)lv.H[threadIdx.x];
    }
    const float valid_w = (float)s_cnt[2 * l], valid_h = (float)s_cnt[2 * l + 1];
    const float vrw_l = valid_w / (float)w, vrh_l = valid_h / (float)h;       // this level's own ratios (no dynamic register index)

    // the thread's channel group for the embedding pass: channels 8 cg .. 8 cg + 7 (cg < 16: pos_y, else pos_x)
    const int cg = threadIdx.x & 31, tsub = threadIdx.x >> 5;
    const float* dtab = cg < 16 ? dim_ty : dim_tx;
    float dt[4], le[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) dt[e] = dtab[(cg & 15) * 8 + 2 * e];
#pragma unroll
    for (int e = 0; e < 8; ++e) le[e] = level_embed[l * 256 + cg * 8 + e];

    const int sy = nearest_src(row, H, h);
    const float scale = 6.283185307179586f;                      // 2 * math.pi as the fp32 scalar torch multiplies with
    const float wh_l = 0.05f * (float)(1 << l);
    int carry = 0;
    for (int c0 = 0; c0 < w; c0 += 256) {
        const int j = c0 + threadIdx.x;
        const bool live = j < w;
        const int sx = live ? nearest_src(j, W, w) : 0;
        const bool pad = live ? mb[(long)sy * W + sx] != 0 : true;
        // x_embed: inclusive count of unpadded pixels of this row up to j
        const unsigned long long bal = __ballot(live && !pad);
        const int within = __popcll(bal & ((2ull << lane) - 1ull));
        if (lane == 0) s_wtot[wave] = __popcll(bal);
        // y_embed: unpadded pixels of column j in rows <= row, and in all rows
        int ycum = 0, ytot = 0;
        if (live)
            for (int i = 0; i < h; ++i) {
                const int v = mb[(long)nearest_src(i, H, h) * W + sx] ? 0 : 1;
                ytot += v;
                if (i <= row) ycum += v;
            }
        __syncthreads();
        int before = carry, total = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int t = s_wtot[q]; if (q < wave) before += t; total += t; }
        const int xcum = before + within;
        carry += total;
        // the row total is only known after the last chunk: x_embed's normaliser is computed separately below
        if (live) {
            const long tok = (long)b * S + lv.start[l] + (long)row * w + j;
            mask_flat[tok] = pad ? 1 : 0;
            // proposals (utils.py:31-62)
            const float px = ((float)j + 0.5f) / valid_w, py = ((float)row + 0.5f) / valid_h;
            const bool ok = px > 0.01f && px < 0.99f && py > 0.01f && py < 0.99f && wh_l > 0.01f && wh_l < 0.99f;
            const float inf = __builtin_huge_valf();
            float4 pr;
            if (pad || !ok) pr = make_float4(inf, inf, inf, inf);
            else {
                const float lw = logf(wh_l / (1.f - wh_l));
                pr = make_float4(logf(px / (1.f - px)), logf(py / (1.f - py)), lw, lw);
            }
            *reinterpret_cast<float4*>(proposals + tok * 4) = pr;
            keep[tok] = (!pad && ok) ? 1 : 0;
            // encoder reference points (deformable_transformer.py:479-492): own-level normalised centre, times every level's ratio
            const float ry = ((float)row + 0.5f) / (vrh_l * (float)h);
            const float rx = ((float)j + 0.5f) / (vrw_l * (float)w);
            float4* er = reinterpret_cast<float4*>(enc_ref + tok * 8);
            er[0] = make_float4(rx * vrw[0], ry * vrh[0], rx * vrw[1], ry * vrh[1]);
            er[1] = make_float4(rx * vrw[2], ry * vrh[2], rx * vrw[3], ry * vrh[3]);
        }
        s_xe[threadIdx.x] = (float)xcum;                          // normalised once the row total is known
        s_ye[threadIdx.x] = live ? ((float)ycum / ((float)ytot + 1e-6f)) * scale : 0.f;
        __syncthreads();
        // row total = count over the WHOLE row: finish the scan of the remaining chunks (counts only) on the first chunk
        // (rows wider than 256 tokens: w = 320 at 128 x 2560) -- cheap: one ballot per remaining chunk
        int rowtot = carry;
        for (int c1 = c0 + 256; c1 < w; c1 += 256) {
            const int j1 = c1 + threadIdx.x;
            const bool v1 = j1 < w && mb[(long)sy * W + nearest_src(j1, W, w)] == 0;
            const unsigned long long b1 = __ballot(v1);
            if (lane == 0) s_wtot[wave] = __popcll(b1);
            __syncthreads();
            rowtot += s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
            __syncthreads();
        }
        const float xden = (float)rowtot + 1e-6f;
        // ---- embedding: thread = (token tsub + 8 k, channel group cg) ---------------------------------------------------
        const int ntok = min(256, w - c0);
        for (int k = tsub; k < ntok; k += 8) {
            const float v = cg < 16 ? s_ye[k] : (s_xe[k] / xden) * scale;
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sn, cs;
                sincosf(v / dt[e], &sn, &cs);
                o[2 * e] = sn + le[2 * e];
                o[2 * e + 1] = cs + le[2 * e + 1];
            }
            const long tok = (long)b * S + lv.start[l] + (long)row * w + c0 + k;
            GeoOut<OutT>::st8(pos + tok * 256 + cg * 8, o);
        }
        __syncthreads();
    }
}

}  // namespace dtlr

using namespace dtlr;

extern "C" int dtlr_geometry(const unsigned char* mask, int B, int H, int W, const int* level_hw,
                             const float* level_embed, const float* dim_ty, const float* dim_tx, int pos_dtype,
                             unsigned char* mask_flat, unsigned char* keep, void* pos, float* valid_ratios,
                             float* enc_ref, float* proposals, void* stream)
{
    clear_stale_error();
    if (!mask || !level_hw || !level_embed || !dim_ty || !dim_tx || !mask_flat || !keep || !pos || !valid_ratios || !enc_ref || !proposals)
        return DTLR_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0) return DTLR_EINVAL;
    GeoLevels lv;
    int S = 0, rows = 0;
    for (int l = 0; l < 4; ++l) {
        lv.H[l] = level_hw[2 * l]; lv.W[l] = level_hw[2 * l + 1]; lv.start[l] = S;
        if (lv.H[l] <= 0 || lv.W[l] <= 0) return DTLR_EINVAL;
        S += lv.H[l] * lv.W[l];
        rows += lv.H[l];
    }
    if (B > 65535) return DTLR_ESHAPE;
    const dim3 grid(rows, B);
    if (pos_dtype == DTLR_H16)
        hipLaunchKernelGGL(geometry_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, mask, H, W, lv, S, level_embed, dim_ty, dim_tx,
                           mask_flat, keep, (uint16_t*)pos, valid_ratios, enc_ref, proposals);
    else if (pos_dtype == DTLR_F32)
        hipLaunchKernelGGL(geometry_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, mask, H, W, lv, S, level_embed, dim_ty, dim_tx,
                           mask_flat, keep, (float*)pos, valid_ratios, enc_ref, proposals);
    else
        return DTLR_EDTYPE;
    return check_launch();
}
