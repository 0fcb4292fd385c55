/* dtlr_hip.h -- C ABI of libdtlr_hip.so, the MI355X (gfx950) implementation of the DTLR
 * inference hot path.  Plain pointers and sizes only; no torch types.  Every entry point:
 *   - takes DEVICE pointers to contiguous buffers owned by the caller,
 *   - enqueues its kernels on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *     and returns without synchronising,
 *   - returns DTLR_OK (0) or a negative DTLR_E* code and never throws; dtlr_strerror() maps codes
 *     to text.  Kernel-launch failures are returned (the reference only printf's them,
 *     models/dino/ops/src/cuda/ms_deform_im2col_cuda.cuh:948-952).
 *
 * Each declaration cites the reference interface it replaces (paths under /root/reference).
 */
#ifndef DTLR_HIP_H
#define DTLR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DTLR_OK 0
#define DTLR_EINVAL (-1)   /* null pointer / non-positive size */
#define DTLR_EDTYPE (-2)   /* unsupported dtype code */
#define DTLR_ESHAPE (-3)   /* shape outside what the kernels support */
#define DTLR_ELAUNCH (-4)  /* hipLaunch / runtime error, see dtlr_last_hip_error() */

/* dtype codes (the arithmetic/storage type of the floating-point operands) */
#define DTLR_F32 0
#define DTLR_F64 1
#define DTLR_BF16 2
#define DTLR_F16 3    /* IEEE fp16: accepted by libdtlr_hip_f16.so (the same sources built with -DDTLR_HALF_IS_F16) wherever libdtlr_hip.so accepts DTLR_BF16 */
#define DTLR_F32S 4   /* "split fp32": fp32 activations / results in memory, products as three fp16 MFMAs on hi + lo halves (per term <= 2^-21 |a w| +
                         2^-24 (|a| + |w|): 22-bit operands down to |x| = 2^-3, an absolute floor of 2^-25 below -- ~18 bits at |w| = 1e-2;
                         16/3 x the rate of the exact-fp32 MFMA).  Accepted as the input dtype of dtlr_gemm_nt / _a2bcast / _rowmax(_lda)
                         and dtlr_conv2d_nhwc; the weight argument is then the image written by dtlr_split_pack_weights (same size as the
                         fp32 weight).  |activation| must stay below 65504. */

const char *dtlr_strerror(int code);
int dtlr_last_hip_error(void);          /* last hipError_t seen by this library (thread-local) */
int dtlr_abi_version(void);

/* The library's only internal scratch memory: one workspace per (device, stream), used by the split-K convolutions / GEMMs
 * (dtlr_conv2d_nhwc, dtlr_gemm_nt at grids that cannot fill the chip) and by the hidden-dimension split of dtlr_ffn_fused_bf16 /
 * dtlr_ffn_split at small M.  Contract:
 *   - a buffer that has been handed to a launch is never freed or moved (its address may live in a captured HIP graph); a larger
 *     request allocates a new buffer and keeps the old one (dtlr_workspace_retired_bytes(): how much is kept that way);
 *   - under stream capture nothing is allocated: a launch borrows the device's largest existing buffer, and falls back to its
 *     unsplit kernel (another summation order, same tolerance) when none is large enough.  A caller that captures should therefore
 *     call dtlr_workspace_reserve(bytes, stream) once before capturing (128 MiB covers every shape of a 32-line batch);
 *   - two streams never share a buffer outside capture.
 * No reference counterpart (torch's caching allocator plays this role for the reference's temporaries). */
int dtlr_workspace_reserve(long bytes, void *stream);
long dtlr_workspace_retired_bytes(void);

/* ---------------------------------------------------------------------------------------------
 * Multi-scale deformable attention forward.
 * Replaces: MultiScaleDeformableAttention.ms_deform_attn_forward
 *           (models/dino/ops/src/vision.cpp:13-16 -> ms_deform_attn.h:20-39 ->
 *            cuda/ms_deform_attn_cuda.cu:20-80 -> cuda/ms_deform_im2col_cuda.cuh:237-299),
 *           called by MSDeformAttnFunction.forward (ops/functions/ms_deform_attn_func.py:23-29).
 *   value  [N,S,M,D]          dtype
 *   shapes [L,2] int64 (H,W)  level_start_index [L] int64      (device, as the reference passes)
 *   loc    [N,Lq,M,L,P,2]     (x,y) in units of the level's width/height;  dtype (f32 for BF16)
 *   attn   [N,Lq,M,L,P]       dtype (f32 for BF16)
 *   out    [N,Lq,M*D]         dtype; every element is written
 * out[b,q,m,:] = sum_l sum_p attn * bilinear(value_l[b,:,m,:], (x*W-0.5, y*H-0.5)), zero padding.
 * No im2col_step: the whole batch is one launch (the reference's chunk loop, cu:50-75, is a CUDA
 * grid-size workaround; its divisibility check is kept in the Python binding).
 * `value` must be finite: the fused forms (Lq x L x P = 4 x 4 front ends) give a corner outside the map a ZERO
 * WEIGHT on the clamped border pixel where the reference skips the read (cuh:49-70), so an inf / NaN stored in a
 * border pixel turns outputs the reference keeps finite into NaN.
 */
int dtlr_msda_forward(const void *value, const int64_t *shapes, const int64_t *level_start_index,
                      const void *loc, const void *attn,
                      int N, int S, int M, int D, int L, int Lq, int P,
                      int dtype, void *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * MSDeformAttn.forward front end fused into the sampling kernel (L = 4 levels, P = 4 points).
 * Replaces: ops/modules/ms_deform_attn.py:97-124 -- `sampling_offsets`/`attention_weights` views,
 *           F.softmax over the 16 (level,point) logits per head, the sampling-location arithmetic
 *           (102-105 for 2-d reference points, 106-108 for 4-d boxes) and MSDeformAttnFunction.apply.
 *   ow   [N,Lq, M*L*P*2 + M*L*P]  the fused projection row: offsets (M,L,P,2) then logits (M,L*P);
 *                                 ow_dtype F32 or BF16
 *   ref  [N,Lq,L,ref_dim] fp32    ref_dim 2: (x,y) per level;  4: (cx,cy,w,h) per level
 *   value/out as dtlr_msda_forward (dtype F32 or BF16).
 */
int dtlr_msda_fused_forward(const void *value, const int64_t *shapes, const int64_t *level_start_index,
                            const void *ow, const float *ref, int ref_dim,
                            int N, int S, int M, int D, int L, int Lq, int P,
                            int dtype, int ow_dtype, void *out, void *stream);
/* Same, with `value` a column slice of a wider [N, S, value_row_stride] tensor (value_row_stride elements between
 * consecutive spatial positions, >= M*D, multiple of 8; 0 = contiguous): the value projections of all decoder layers
 * (`value_proj(memory)` of every layer, ms_deform_attn.py:94, read the same `memory`) are produced by ONE GEMM. */
int dtlr_msda_fused_forward_strided(const void *value, int value_row_stride, const int64_t *shapes,
                                    const int64_t *level_start_index, const void *ow, const float *ref, int ref_dim,
                                    int N, int S, int M, int D, int L, int Lq, int P,
                                    int dtype, int ow_dtype, void *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Encoder self-attention form of the above (queries ARE the pixels of the L = 4 levels, Lq = S), with
 * the value windows of each (x-tile, head) staged in LDS and a global-memory path for samples that
 * leave the window.  Same inputs/outputs/arithmetic as dtlr_msda_fused_forward with ref_dim = 2.
 * Replaces: MSDeformAttn.forward lines 97-124 as called from DeformableTransformerEncoderLayer
 *           (models/dino/deformable_transformer.py:810-813).
 *   level_hw  HOST array [L*2] = (H_0, W_0, ..., H_3, W_3) -- the tiling is planned on the host
 *   halo      extra window columns on each side, in pixels of every level (offsets beyond it still
 *             give exact results through the global path)
 *   D must be 32, L = P = 4.  Returns DTLR_ESHAPE if no window plan fits the 160 KB LDS.
 */
int dtlr_msda_encoder_forward(const void *value, const void *ow, const float *ref, const int *level_hw,
                              int N, int M, int D, int L, int P, int halo,
                              int dtype, int ow_dtype, void *out, void *stream);
/* 1 if the LDS window plan of dtlr_msda_encoder_forward fits these level shapes, 0 if not (canvases taller than ~270 px in
 * fp32 / ~550 px in bf16: use dtlr_msda_fused_forward, which has no size limit), negative DTLR_E* on bad arguments. */
int dtlr_msda_encoder_plan_ok(const int *level_hw /* host, 8 ints */, int dtype, int halo);
/* Far-sample probe for the choice between dtlr_msda_encoder_forward (LDS windows) and dtlr_msda_fused_forward (gather) on a given
 * layer: counts[0] += sampling points (ops/modules/ms_deform_attn.py:102-105: reference point + offset / (W_l, H_l)) that lie inside
 * their map but outside the column window the LDS kernel stages for the query's tile -- each costs that kernel a dependent global
 * gather; counts[1] += points inside the map.  counts: 2 x uint64 in DEVICE memory, accumulated.  ow / ref / level_hw / halo as for
 * dtlr_msda_encoder_forward; dtype = the value dtype (it sets the window plan), ow_dtype = the projection row's dtype. */
int dtlr_msda_encoder_far_samples(const void *ow, const float *ref, const int *level_hw, int N, int M, int halo,
                                  int dtype, int ow_dtype, unsigned long long *counts, void *stream);

/* ---------------------------------------------------------------------------------------------
 * y = LayerNorm(x [+ residual]) * gamma + beta over rows of C channels (C multiple of 256).
 * Replaces: the `src = norm(src + dropout(src2))` post-norm pattern of
 *           DeformableTransformerEncoderLayer / DecoderLayer (deformable_transformer.py:804-823,
 *           876-959; nn.LayerNorm eps 1e-5, dropout 0) and TransformerDecoder.norm (:758).
 *   x, residual (may be NULL), y: [rows, C] dtype (F32 or BF16); gamma, beta: [C] fp32.
 */
int dtlr_layernorm(const void *x, const void *residual, const float *gamma, const float *beta,
                   void *y, long rows, int C, float eps, int dtype, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The same block for the SPLIT-fp32 engine (round 4): X, Y [M, 256] fp32, every product as three fp16 MFMAs on hi + lo halves
 * (fp32-grade), fp32 residual and LayerNorm, the [M, d_ff] intermediate on chip.
 * Replaces: forward_ffn + norm2 / norm3 (models/dino/deformable_transformer.py:804-823, 876-880).
 *   Wp: even(d_ff / 32) + dtlr_ffn_split_pad_chunks() blocks of 64 KB, block c = [W1_hi | W1_lo | W2_hi | W2_lo] of hidden units 32 c .. 32 c + 31,
 *       each part 16 fragments of 1 KB in the fragment order of dtlr_ffn32_pack_weights; hi = fp16(w), lo = fp16(w - hi); zero blocks
 *       behind the last real chunk (dtlr_amd.ops.ffn_split_pack builds it).  b1 [d_ff], b2 / gamma / beta [256] fp32.
 *   d_ff a multiple of 32, 32 <= d_ff <= 2048 (DTLR_ESHAPE otherwise).
 */
int dtlr_ffn_split(const void *X, const void *Wp, const float *b1, const float *b2, const float *gamma, const float *beta,
                   float eps, void *Y, long M, int d_ff, void *stream);
int dtlr_ffn_split_pad_chunks(void);

/* ---------------------------------------------------------------------------------------------
 * Token-stationary class head for large charsets (round 5; 16-bit engines: DTLR_BF16 in libdtlr_hip.so, DTLR_F16 in libdtlr_hip_f16.so):
 *   nprod 3:  Y = A Whi^T + B Whi^T + A Wlo^T + bias      nprod 2:  Y = A Whi^T + A Wlo^T + bias
 *   A = X[:, a_off : a_off + 256], B = X[:, b_off : b_off + 256] of the 16-bit rows X [M, ldx]; Whi = 16-bit(W), Wlo = 16-bit(W - Whi), W [N, 256] fp32.
 *   mode 0: out [M] fp32 = max over the N classes of Y.
 *           Replaces: topk_logits = enc_outputs_class_unselected.max(-1)[0] of the two-stage selection
 *           (models/dino/deformable_transformer.py:341-345) on the [hi | lo | hi] image of output_memory (a_off 0, b_off 256, nprod 3).
 *   mode 1: out [M, N] fp32 = Y (N % 4 == 0).
 *           Replaces: class_embed on the decoder states (models/dino/dino.py:349-352; nprod 2 on the 16-bit state) and the interm_outputs
 *           logits of the selected two-stage rows (dino.py:382-385; nprod 3).
 *   Wp: ceil(N / 32) + dtlr_head_ts_pad_chunks() blocks of 32 KB, block c = [Whi fragments | Wlo fragments] of classes 32 c .. 32 c + 31,
 *       16 fragments of 1 KB each: lane l of k-step s <- W[32 c + (l & 31)][16 s + 8 (l >> 5) .. + 7]; zero blocks behind the last chunk
 *       (dtlr_amd.ops.head_ts_pack builds it).  bias: 32 ceil(N / 32) floats, classes >= N at -3e38.
 *   ldx, a_off, b_off multiples of 8; N <= 24576 (mode 0) / 15360 (mode 1).  Same three (two) terms as the tiled GEMM on [hi | lo | hi] . [Whi | Whi | Wlo], summed in
 *   another order: equal to fp32 rounding.  The tokens are stationary in registers, the weights stream through LDS once per 256 tokens.
 */
int dtlr_head_ts(const void *X, int ldx, int a_off, int b_off, const void *Wp, const float *bias, int N, int nprod, int mode,
                 void *out, long M, void *stream);
int dtlr_head_ts_pad_chunks(void);

/* ---------------------------------------------------------------------------------------------
 * SPLIT-fp32 engine, K = N = 256: the weight-resident streaming projection (round 4).  A, C (and R) [M, 256] fp32.
 *   R == NULL, gamma == NULL: C = A W^T + bias, rows with row_mask[m] != 0 (may be NULL) written as zeros.
 *              Replaces: value = self.value_proj(input_flatten); value.masked_fill(padding_mask, 0) (ops/modules/ms_deform_attn.py:94-96).
 *   R != NULL: C = LayerNorm(R + A W^T + bias), gamma / beta [256] fp32 (row_mask ignored).
 *              Replaces: output_proj (ms_deform_attn.py:124) + src = norm1(src + dropout1(src2)) (deformable_transformer.py:810-815).
 *   R == NULL, gamma != NULL: C = LayerNorm(bias + A W^T), the product of rows with row_mask[m] != 0 (may be NULL) taken as zero.
 *              Replaces: output_memory.masked_fill(..) -> enc_output -> enc_output_norm of the two-stage front end
 *              (deformable_transformer.py:320-330; models/dino/utils.py:58-62).
 *   Wp = dtlr_k256s_pack_weights(W [256, 256] fp32): 256 KB, fp16 hi then lo halves in MFMA fragment order
 *        ([8 waves][2 row tiles][8 k-steps][64 lanes][8]: lane (m, g) <- W[32 wave + 16 rt + m][32 ks + 8 g + e]).
 */
int dtlr_k256s_pack_weights(const float *w, void *out, void *stream);
int dtlr_gemm_k256s(const float *A, const void *Wp, const float *bias, const float *R, const unsigned char *row_mask,
                    const float *gamma, const float *beta, float eps, float *C, long M, void *stream);

/* The same kernel structure with SEVERAL output slices per launch (round 6): ONE pass over A [M, 256] fp32 for `nslices` (1..8)
 * projections of it,
 *     C_y[m, 0:n_valid_y] = act( A[m, :] W_y^T + bias_y + R_y[m or m % res_rows, 0:n_valid_y] ),   rows with row_mask[m] != 0 written as zeros.
 * Replaces (split-fp32 engine):
 *   - decoder: value_proj(memory) of all six cross-attention layers (ms_deform_attn.py:94-96) as six slices of one [M, 1536] buffer;
 *   - (built and tested, not used by the engine: measured slower than its parts -- the slices' token tiles do not meet in L2)
 *     encoder layer, unpadded batch: value = value_proj(src) AND [sampling_offsets | attention_weights](src + pos)
 *     (ops/modules/ms_deform_attn.py:94-98) as three slices (256 | 256 | 128 channels), the position term pos W^T + b as the
 *     row-broadcast residual (res_rows = tokens per image).
 * slices: HOST array of dtlr_k256s_slice.  Wp = dtlr_k256s_pack_weights of the slice's weight zero-padded to [256, 256]; bias [n_valid] fp32 or
 * NULL; R fp32 or NULL, row stride ldr: [M, ldr] when res_rows == 0, ONE [res_rows, ldr] matrix shared by the M / res_rows images otherwise
 * (res_rows % 32 == 0, M % res_rows == 0; launch-wide); C = the slice's first output column, row stride ldc; n_valid a multiple of 32, <= 256;
 * relu != 0: ReLU after the residual; row_mask [M] or NULL applies to every slice.  C and R 16-byte aligned, ldc / ldr multiples of 4. */
typedef struct dtlr_k256s_slice {
    const void *Wp;
    const float *bias;
    const float *R;
    float *C;
    int ldc, ldr, n_valid, relu;
} dtlr_k256s_slice;
int dtlr_gemm_k256s_multi(const float *A, long M, const dtlr_k256s_slice *slices, int nslices, const unsigned char *row_mask,
                          int res_rows, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused position-wise feed-forward block + residual + LayerNorm, bf16 (fp32 accumulate / statistics):
 *     Y = LayerNorm( X + relu(X W1^T + b1) W2^T + b2 )
 * Replaces: DeformableTransformerEncoderLayer.forward_ffn + norm2
 *           (models/dino/deformable_transformer.py:804-823: linear1 -> relu -> linear2, src + src2, norm2)
 *           and the decoder layer's ffn + norm3 (:876-880), dropout = identity at inference.
 *           The [M, d_ff] intermediate stays on chip (never written to HBM).
 *   X, Y [M, d_model] bf16 ; W1 [d_ff, d_model] bf16 ; b1 [d_ff], b2/gamma/beta [d_model] fp32
 *   W2p: linear2.weight [d_model, d_ff] packed chunk-major, W2p[c][o][k] = W2[o][32 c + k] (shape [d_ff/32][d_model][32], bf16)
 *   d_model must be 256, d_ff a multiple of 32 and <= 2048 (DTLR_ESHAPE otherwise).
 */
int dtlr_ffn_fused_bf16(const void *X, const void *W1, const float *b1, const void *W2p, const float *b2,
                        const float *gamma, const float *beta, float eps, void *Y,
                        int M, int d_model, int d_ff, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Weight-resident streaming GEMM with a streamed residual, bf16:   C = relu?( A W^T + b + R )
 * Replaces: the last 1x1 convolution of a torchvision ResNet bottleneck with its folded FrozenBN, the identity add and the ReLU
 *           (`out = relu(bn3(conv3(out)) + identity)`; models/dino/backbone.py:62-72,97-106 wrapping torchvision resnet50) and the
 *           stride-1 1x1 `downsample` convolution of layer1 -- the HBM-streaming shapes K = 64 / 128 / 256, N = 256 / 512 / 1024.
 *   dtlr_gemm_kres_pack_weights: HOST-side packer: W [N, K] bf16 row-major -> fragment order; N a multiple of 256 (N * K elements out),
 *                   or N = 64 / 128 / 192 without residual (the bottleneck's first 1x1 convolution): one zero-padded 256-channel column,
 *                   256 * K elements out.
 *   dtlr_gemm_kres: A [M, K] bf16 ; Wp = device copy of the packed weight ; bias [N] fp32 or NULL ; R [M, N] bf16 or NULL ;
 *                   C [M, N] bf16 ; relu != 0 applies ReLU after the residual add.
 */
int dtlr_gemm_kres_pack_weights(const unsigned short *w_host, unsigned short *wp_host, int N, int K);
int dtlr_gemm_kres(const void *A, const void *Wp, const float *bias, const void *R, void *C, int M, int N, int K, int relu, void *stream);
/* Layer1's bottleneck tails chained with their neighbours (torchvision resnet50 `layer1`, models/dino/backbone.py:97-106), 64-channel
 * inputs, 256 output channels:
 *     C  = relu?( [A | A2] W^T + bias (+ R) )     exactly one of A2 [M, 64] (the block input x of the FIRST bottleneck: its 1x1 `downsample`
 *                                                 shortcut becomes K columns 64..127 of the same GEMM, W = [W3 | Wd], bias = b3 + bd, and
 *                                                 no shortcut map is written or read) and R [M, 256] (identity shortcut)
 *     C2 = relu( C W2^T + bias2 )                 the NEXT bottleneck's first 1x1 convolution (N2 = 64 inside layer1, 128 for layer2.0),
 *                                                 computed while the 64-row tile of C is on chip; Wp2 NULL: not computed (A2 form only)
 *   Wp / Wp2: device copies of dtlr_gemm_kres_pack_weights(W [256, 64 or 128]) / (W2 [N2, 256]).  C2 is bit-identical to dtlr_gemm_kres
 *   on the stored C; with R, C is bit-identical to dtlr_gemm_kres. */
int dtlr_gemm_kres_chain(const void *A, const void *A2, const void *Wp, const float *bias, const void *R, void *C, int M, int relu,
                         const void *Wp2, const float *bias2, void *C2, int N2, void *stream);
/* layer2's first bottleneck tail (torchvision resnet50 `layer2[0]`, stride on the 3x3: `out = relu(bn3(conv3(t)) + downsample(x))` with
 * downsample = conv1x1 stride 2 + FrozenBN) as ONE GEMM: the strided shortcut convolution is K columns 128..383,
 *     C[(b, i, j), :] = relu?( [A[(b, i, j), :] | X[b, 2 i, 2 j, :]] W^T + bias ),   W = [W3 | Wd] [512, 384], bias = b3 + bd
 *   A [B Hout Wout, 128], X [B, Hin, Win, 256] NHWC, Hout = (Hin - 1) / 2 + 1, Wout likewise, C [B Hout Wout, 512];
 *   Wp = device copy of dtlr_gemm_kres_pack_weights(W, 512, 384). */
int dtlr_gemm_kres_cat_s2(const void *A, const void *X, const void *Wp, const float *bias, void *C, int B, int Hin, int Win, int relu,
                          void *stream);
/* The encoder's [sampling offsets | attention logits] projection (ops/modules/ms_deform_attn.py:97-98 on query = src + pos) for an
 * unpadded batch: (src + pos) W^T + b = src W^T + (pos W^T + b), the second term ONE [res_rows, 384] bf16 matrix shared by all images:
 *     C[m, :] = A[m, :] W^T + R[m % res_rows, :]        A [M, 256], C [M, 384] bf16
 *   Wp = device copy of dtlr_gemm_kres_pack_weights_bcast384(W [384, 256]) (512 x 256 elements).  res_rows a multiple of 64 and M a
 *   multiple of res_rows (DTLR_ESHAPE otherwise; dtlr_gemm_k256 takes any shape). */
int dtlr_gemm_kres_pack_weights_bcast384(const unsigned short *w_host, unsigned short *wp_host);
int dtlr_gemm_kres_bcast384(const void *A, const void *Wp, const void *R, int res_rows, void *C, int M, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The same fused feed-forward block for MANY rows (the encoder call, M = batch x 5440 tokens), built on the 32x32x16 MFMA with
 * both weights pre-packed in fragment order:
 *     Y = LayerNorm( X + relu(X W1^T + b1) W2^T + b2 )            X, Y [M, 256] bf16
 * Replaces: DeformableTransformerEncoderLayer.forward_ffn + norm2 (models/dino/deformable_transformer.py:804-823).
 *   dtlr_ffn32_pack_weights: HOST-side packer (all four pointers host memory): linear1.weight [d_ff, 256] and linear2.weight
 *     [256, d_ff] (bf16) -> two images of (d_ff/32 + dtlr_ffn32_pad_chunks()) x 16 KB; the trailing chunks are zero (they are
 *     streamed by the kernel's steady-state loop and never multiplied).  d_ff a multiple of 32, 64 <= d_ff <= 2048.
 *   dtlr_ffn32_bf16: W1p / W2p = device copies of those images; b1 [d_ff], b2 / gamma / beta [256] fp32.
 */
int dtlr_ffn32_pack_weights(const void *w1, const void *w2, void *w1p, void *w2p, int d_ff);
int dtlr_ffn32_pad_chunks(void);
int dtlr_ffn32_bf16(const void *X, const void *W1p, const float *b1, const void *W2p, const float *b2,
                    const float *gamma, const float *beta, float eps, void *Y, long M, int d_ff, void *stream);
/*   dtlr_ffn4_bf16 (round 6): the same block, same packed images, as ONE persistent launch for any M >= 1 (ffn4.hip): every workgroup streams
 *     the weights cyclically and carries two 128-row tiles half a tile period apart, so that a tile's epilogue (LayerNorm, stores) and the
 *     load of its successor run under the other tile's MFMAs; no tail kernel.  Same result as dtlr_ffn32_bf16 up to the fp32 summation order
 *     over the hidden chunks (a tile starts at the chunk the stream happens to be at).
 */
int dtlr_ffn4_bf16(const void *X, const void *W1p, const float *b1, const void *W2p, const float *b2,
                   const float *gamma, const float *beta, float eps, void *Y, long M, int d_ff, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Output projection + residual + LayerNorm of an attention block, bf16 (fp32 accumulate / statistics):
 *     Y = LayerNorm( R + A W^T + b )            A, R, Y [M, 256] bf16 ; b, gamma, beta [256] fp32
 *     W: the [256, 256] bf16 weight re-ordered by dtlr_proj_pack_weights (a HOST-side helper, both pointers host memory)
 * Replaces: MSDeformAttn.output_proj (ops/modules/ms_deform_attn.py:124) / nn.MultiheadAttention.out_proj followed by
 *           `src = norm1(src + dropout1(src2))` (models/dino/deformable_transformer.py:810-815) and the decoder's
 *           norm2 / norm1 after self- and cross-attention (:847-870).  d_model must be 256.
 */
int dtlr_proj_pack_weights(const unsigned short *w_host /* [256*256] bf16, row-major */, unsigned short *wp_host /* [256*256] */);
int dtlr_proj_ln_bf16(const void *A, const void *W, const float *bias, const void *R,
                      const float *gamma, const float *beta, float eps, void *Y, int M, int d_model, void *stream);

/* Two-stage query selection front end, bf16 engine: output_memory = LayerNorm(Linear(memory with masked tokens zeroed)),
 * written as three bf16 images per token, Y3[tok] = [hi | lo | hi] (hi = bf16(y), lo = bf16(y - hi)), so that a class head
 * stored as [W_hi | W_hi | W_lo] ([C, 768] bf16) yields y W^T to ~2^-16 relative on the bf16 matrix cores.
 * Replaces: `output_memory = output_memory.masked_fill(...)` + `enc_output_norm(enc_output(output_memory))`
 *           (models/dino/utils.py:60-62 ; models/dino/deformable_transformer.py:325-340) ahead of the class head / top-k (:341-345).
 *   A [M,256] bf16 ; W packed by dtlr_proj_pack_weights ; bias, gamma, beta [256] fp32 ;
 *   keep [M] uint8 (0 = zero the token's features before the projection) or NULL ; Y3 [M,768] bf16. */
int dtlr_proj_ln_split_bf16(const void *A, const void *W_packed, const float *bias, const unsigned char *keep,
                            const float *gamma, const float *beta, float eps, void *Y3, int M, int d_model, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-head self-attention over the decoder queries, fused (scores/softmax/PV stay on chip).
 * Replaces: nn.MultiheadAttention(256, 8, dropout=0)(q, k, v)[0] minus its in/out projections, as
 *           called from DeformableTransformerDecoderLayer.forward_sa
 *           (models/dino/deformable_transformer.py:847,904-907): softmax(q k^T / sqrt(head_dim)) v
 *           per head, no masks (eval).
 *   qk  [B, L, 2*H*head_dim]  projected q (first half of the row) and k (second half);  dtype
 *   v   [B, L, H*head_dim]    projected v;  dtype
 *   vt_workspace              >= dtlr_mha_workspace_bytes(B, L, H, head_dim) bytes of scratch
 *   out [B, L, H*head_dim]    dtype.   head_dim must be 32; dtype BF16 (fp32 accumulate/softmax), F32
 *                             (exact-fp32 MFMA) or F32S (fp32 tensors, every product as three fp16 MFMAs on hi + lo halves: fp32-grade;
 *                             the workspace is not used).
 */
int dtlr_mha_forward(const void *qk, const void *v, void *vt_workspace, void *out,
                     int B, int L, int H, int head_dim, int dtype, void *stream);
long dtlr_mha_workspace_bytes(int B, int L, int H, int head_dim);

/* ---------------------------------------------------------------------------------------------
 * C[M,N] = epilogue( (A [+ A2])[M,K] . W[N,K]^T )   -- every nn.Linear and 1x1 convolution on the path.
 * Replaces: F.linear / nn.Linear.forward call sites of the hot path: MSDeformAttn's value_proj,
 *           sampling_offsets|attention_weights, output_proj (ops/modules/ms_deform_attn.py:94-98,125);
 *           FFN linear1/linear2 (deformable_transformer.py:804-808,876-880); MultiheadAttention in/out
 *           projections (:847); enc_output, class/bbox heads (:338-342, dino.py:339-354); ref_point_head;
 *           1x1 Conv2d of input_proj (dino.py:121-123) and of the ResNet bottlenecks on NHWC tokens.
 *   A, A2 (may be NULL): [M,K] in_dtype; the prologue adds them (query = src + pos).
 *   W: [N,K] in_dtype (nn.Linear layout).  bias (may be NULL): [N] fp32.
 *   epilogue order: + bias -> ReLU (if relu == 1) -> zero rows where row_mask[m] != 0 (may be NULL,
 *   [M] bytes; value.masked_fill of ms_deform_attn.py:95-96) -> + residual (may be NULL, [M,N] out_dtype)
 *   -> ReLU (if relu == 2: the bottleneck tail relu(conv3 + identity)) -> store.
 *   in_dtype BF16 (K % 64 == 0; fp32 accumulate; out BF16 or F32), F32 (K % 32 == 0; exact fp32 MFMA) or F32S (K % 32 == 0; A / A2 /
 *   residual / C fp32, W the dtlr_split_pack_weights image: three fp16 MFMAs per product, fp32-grade).
 */
/* W [rows, K] fp32 (nn.Linear layout; a [Cout, KH, KW, Cin] convolution weight is rows = Cout, K = KH*KW*Cin) -> `out`, rows*K*4 bytes:
 * per 32-k slab of a row, 16-byte chunk c < 4 holds fp16(w) of k 8c..8c+7 and chunk 4 + c holds fp16(w - fp16(w)) of the same k -- the
 * slab image the DTLR_F32S kernels copy to LDS.  Row slices of the image are images of the row slices.  K % 32 == 0.  (round 4) */
int dtlr_split_pack_weights(const float *w, void *out, long rows, int K, void *stream);
/* C = A . W^T [+ bias] + residual[m % res_rows]  (dtype BF16 / F16 / F32 / F32S as dtlr_gemm_nt; residual and C in the output dtype, i.e. the
 * 16-bit format for 16-bit operands, fp32 otherwise): the encoder's [offsets | logits] projection of an unpadded batch,
 * (src + pos) W^T + b = src W^T + (pos W^T + b), where the second term is ONE [S, N] matrix for every image
 * (ops/modules/ms_deform_attn.py:97-98 with query = src + pos, deformable_transformer.py:812).  M % res_rows == 0.  (round 4) */
int dtlr_gemm_nt_resbcast(const void *A, const void *W, const float *bias, const void *residual, int res_rows, void *C,
                          int M, int N, int K, int dtype, void *stream);
int dtlr_gemm_nt(const void *A, const void *A2, const void *W, const float *bias,
                 const void *residual, const unsigned char *row_mask, void *C,
                 int M, int N, int K, int relu, int in_dtype, int out_dtype, void *stream);
/* scores[m] = max over n of (A[m,:] . W[n,:] + bias[n]): dtlr_gemm_nt with a row-max epilogue -- the [M,N] product is never
 * written.  Replaces: enc_outputs_class_unselected.max(-1)[0] (models/dino/deformable_transformer.py:341-345: only the
 * per-token maximum of the two-stage class head feeds torch.topk).  A [M,K], W [N,K] in_dtype (F32: K % 32 == 0, BF16:
 * K % 64 == 0), bias [N] fp32 or NULL, rowmax [M] fp32 (every element written; a row whose products are all -inf stays -inf). */
/* Weight-resident streaming GEMM for K = 256, bf16 in / bf16 out (gemm_k256.hip): C[m, 0:N] = A[m,:] . W^T + bias
 *   [+ resid[m % res_rows, :]] ; rows with row_mask[m] != 0 are written as zeros.  N = 256 or 384.
 * Replaces: value_proj / sampling_offsets+attention_weights of MSDeformAttn on the encoder's T = B*S tokens
 *           (models/dino/ops/modules/ms_deform_attn.py:94-98) and the decoder layers' value_proj(memory) (same file, :94).
 *   A [M,256] bf16 ; Wp = dtlr_k256_pack_weights(W [N,256]) (device copy of the fragment-order image) ; bias [N] fp32 or NULL ;
 *   resid [res_rows, N] bf16 or NULL (M need not be a multiple of res_rows) ; row_mask [M] uint8 or NULL ;
 *   C bf16 with row stride ldc elements (ldc >= N, multiple of 8). */
int dtlr_k256_pack_weights(const unsigned short *w_host, unsigned short *wp_host, int N);
int dtlr_gemm_k256(const void *A, const void *Wp, const float *bias, const void *resid, int res_rows,
                   const unsigned char *row_mask, void *C, int ldc, int M, int N, void *stream);

/* Y = LayerNorm(R + A W^T + bias) * gamma + beta over [M, 256] bf16 rows, in the weight-resident streaming form (gemm_k256.hip):
 * the large-M variant of dtlr_proj_ln_bf16 (same reference lines: deformable_transformer.py:810-815, ms_deform_attn.py:124).
 *   Wp = dtlr_proj_ln_k256_pack_weights(W [256,256]) (device copy). */
int dtlr_proj_ln_k256_pack_weights(const unsigned short *w_host, unsigned short *wp_host);
int dtlr_proj_ln_k256(const void *A, const void *Wp, const float *bias, const void *R, const float *gamma, const float *beta,
                      float eps, void *Y, int M, void *stream);

/* dtlr_gemm_nt with a row-broadcast A2 prologue: C = (A + A2[m % a2_rows]) . W^T + bias.  A2 [a2_rows, K]; M % a2_rows == 0.
 * Replaces: `with_pos_embed(src, pos)` feeding sampling_offsets / attention_weights (models/dino/deformable_transformer.py:
 * 797-812, ops/modules/ms_deform_attn.py:97-98) when the batch is unpadded: the position embedding is then the same [S,256]
 * matrix for every image and stays L2-resident.  dtype F32 or BF16 (operands and result). */
int dtlr_gemm_nt_a2bcast(const void *A, const void *A2, int a2_rows, const void *W, const float *bias, void *C,
                         int M, int N, int K, int dtype, void *stream);
int dtlr_gemm_nt_rowmax(const void *A, const void *W, const float *bias, float *rowmax,
                        int M, int N, int K, int in_dtype, void *stream);
/* the same with A = K columns of a wider row-major matrix (lda elements between rows, lda >= K, rows 16-byte aligned): the first
 * pass of the two-stage selection scores the `hi` image of the [hi|lo|hi] activation only (see DTLREngine.two_stage) */
int dtlr_gemm_nt_rowmax_lda(const void *A, int lda, const void *W, const float *bias, float *rowmax,
                            int M, int N, int K, int in_dtype, void *stream);

/* ---------------------------------------------------------------------------------------------
 * NHWC convolution as an implicit GEMM on the matrix cores, with the FrozenBN-folded bias, optional
 * residual and ReLU fused (same epilogue as dtlr_gemm_nt).
 * Replaces: the 3x3 convolutions of torchvision's resnet50 bottlenecks + FrozenBatchNorm2d + ReLU as
 *           run by BackboneBase.forward (models/dino/backbone.py:36-72,97-106), and the stride-2 3x3
 *           Conv2d of input_proj level 3 (models/dino/dino.py:126-133, used at :299-301).
 *   X [B,H,W,Cin]  W [Cout,KH,KW,Cin] (filter taps outermost, channels innermost)  bias [Cout] fp32 or NULL
 *   residual / Y [B,Ho,Wo,Cout],  Ho = (H + 2 pad - KH)/stride + 1.   dtype BF16 or F32 (X, W, Y alike).
 *   relu: 0 none, 1 after bias, 2 after the residual add.   Needs Cin*sizeof(elem) % 128 == 0.
 */
int dtlr_conv2d_nhwc(const void *X, const void *W, const float *bias, const void *residual, void *Y,
                     int B, int H, int Wd, int Cin, int Cout, int KH, int KW, int stride, int pad,
                     int relu, int dtype, void *stream);
/* The 3x3 / stride 1 / pad 1 bf16 case of dtlr_conv2d_nhwc without residual (the middle convolution of a ResNet bottleneck,
 * models/dino/backbone.py:62-72,97-106) as its own entry: the (8+2) x (16+2) input patch of a workgroup's pixel tile stays resident in
 * LDS and only the weights stream.  Cin in {64, 128, 256}, Cout a multiple of 64 (dtlr_conv3x3_patch_supported); dtlr_conv2d_nhwc
 * routes here by itself.  X [B,H,W,Cin], Wt [Cout,3,3,Cin], Y [B,H,W,Cout] bf16; bias [Cout] fp32 or NULL; relu != 0: ReLU after the bias. */
int dtlr_conv3x3_patch_supported(int Cin, int Cout);
int dtlr_conv3x3_patch_bf16(const void *X, const void *Wt, const float *bias, void *Y, int B, int H, int W, int Cin, int Cout,
                            int relu, void *stream);
/* The same case for the split-fp32 engine (DTLR_F32S; round 6): X, Y fp32 NHWC, Wt = dtlr_split_pack_weights of [Cout,3,3,Cin] (rows = Cout,
 * K = 9 Cin).  The fp32 patch is split into fp16 hi + lo LDS planes ONCE per workgroup (the implicit-GEMM form re-gathers and re-splits it
 * for each of the nine taps); three fp16 MFMAs per product, fp32 accumulation.  Cin = 64 (Cout a multiple of 64) or Cin = 128 (Cout a
 * multiple of 128) (dtlr_conv3x3_patch_f32s_supported); dtlr_conv2d_nhwc routes here by itself.  Same reference lines as above. */
int dtlr_conv3x3_patch_f32s_supported(int Cin, int Cout);
int dtlr_conv3x3_patch_f32s(const float *X, const void *Wt, const float *bias, float *Y, int B, int H, int W, int Cin, int Cout,
                            int relu, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The query stage of one decoder layer in one launch (16-bit engines; csrc/dec_query.hip).
 * Replaces: TransformerDecoder.forward's per-layer preparation -- reference_points * valid_ratios, gen_sineembed_for_position,
 *           ref_point_head (models/dino/deformable_transformer.py:684-692, models/dino/utils.py:141-167) -- and the q / k / v
 *           input projections of DeformableTransformerDecoderLayer's self-attention (nn.MultiheadAttention in_proj on
 *           q = k = tgt + query_pos, v = tgt; deformable_transformer.py:904-907).
 * ref [B*nq,4] fp32 (sigmoided), valid_ratios [B,L,2] fp32, dim_t [128] fp32 (10000^(2(i//2)/128)), tgt [B*nq,256] 16-bit;
 * W0 [256,512] b0, W1 [256,256] b1 (ref_point_head), Wqk [512,256] bqk, Wv [256,256] bv (in_proj split); weights 16-bit, biases fp32.
 * The four weights are handed over in the kernel's fragment order (dtlr_dq_pack_weights, once per model).
 * Outputs: ref_in [B*nq,L,4] fp32, qpos [B*nq,256], qk [B*nq,512], v [B*nq,256] (16-bit).  dtype = the library's 16-bit code. */
int dtlr_dq_pack_weights(const unsigned short *w_host, unsigned short *out_host, int N, int K);   /* [N,K] row-major -> fragment order (host) */
int dtlr_dec_query_stage(const float *ref, const float *valid_ratios, const float *dim_t, const void *tgt,
                         const void *W0, const float *b0, const void *W1, const float *b1,
                         const void *Wqk, const float *bqk, const void *Wv, const float *bv,
                         float *ref_in, void *qpos, void *qk, void *v, int B, int nq, int L, int dtype, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Per-decoder-layer query preparation, fused.
 * Replaces: TransformerDecoder.forward lines 684-690 (reference_points[:, :, None] * cat(valid_ratios,
 *           valid_ratios)) + gen_sineembed_for_position (models/dino/utils.py:141-167).
 *   ref [B*nq,4] fp32 (cx,cy,w,h)   valid_ratios [B,L,2] fp32 (w,h)   dim_t [128] fp32 = 10000**(2*(i//2)/128)
 *   ref_in [B*nq,L,4] fp32          sine [B*nq,512] sine_dtype: [emb(y)|emb(x)|emb(w)|emb(h)] of level 0
 */
int dtlr_decoder_query_prep(const float *ref, const float *valid_ratios, const float *dim_t,
                            float *ref_in, void *sine, int B, int nq, int L, int sine_dtype, void *stream);

/* out = sigmoid(delta + inverse_sigmoid(ref)), elementwise over n fp32 values.
 * Replaces: the box refinement of TransformerDecoder.forward (deformable_transformer.py:734-739) and of
 *           DINO.forward (models/dino/dino.py:343-346) with inverse_sigmoid of util/misc.py:575-579 (eps 1e-3). */
int dtlr_box_refine(const float *delta, const float *ref, float *out, long n, void *stream);

/* Output layer of the box MLP (256 -> 4) fused with its consumer.
 * Replaces: bbox_embed[-1] / enc_out_bbox_embed[-1] (the last nn.Linear of MLP(256,256,4,3), models/dino/dino.py) +
 *           mode 0: `sigmoid(delta + inverse_sigmoid(reference))` (deformable_transformer.py:734-756)
 *           mode 1: `delta + topk proposals` (unsigmoided two-stage boxes, :352-356)
 *   h [rows,256] fp32 (second hidden layer, after ReLU); W [4,256], bias [4], ref [rows,4], out [rows,4] fp32. */
int dtlr_box_head_refine(const float *h, const float *W, const float *bias, const float *ref, float *out,
                         long rows, int hidden, int mode, void *stream);

/* The whole 3-layer box MLP + its consumer in one launch (bf16 engine): hidden layers on the bf16 MFMA path with fp32
 * accumulation, output layer and box arithmetic in fp32.
 * Replaces: `bbox_embed[i](hs)` / `enc_out_bbox_embed(output_memory)` = MLP(256, 256, 4, 3) (models/dino/dino.py) followed by
 *           mode 0 `sigmoid(. + inverse_sigmoid(reference))` (deformable_transformer.py:734-756) or mode 1 `. + proposals` (:352-356).
 *   X [M,256] bf16 ; W1 [256,256] bf16 ; W2p = layer-2 weight packed chunk-major as for dtlr_ffn_fused_bf16 ([8][256][32] bf16) ;
 *   b1, b2 [256] fp32 ; W3 [4,256], b3 [4] fp32 ; ref, out [M,4] fp32. */
int dtlr_box_mlp_refine_bf16(const void *X, const void *W1, const float *b1, const void *W2p, const float *b2,
                             const float *W3, const float *b3, const float *ref, float *out, int M, int mode, void *stream);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm(32, 256) over the tokens of one feature level (statistics per sample and group over
 * T positions x 8 channels, eps, affine).
 * Replaces: the nn.GroupNorm(32, hidden_dim) of every input_proj level (models/dino/dino.py:121-134,
 *           applied at :293,299-301) on the NHWC/token layout.
 *   x, y [B, T, 256] dtype (F32/BF16); gamma, beta [256] fp32; workspace >= dtlr_groupnorm_workspace_bytes.
 */
int dtlr_groupnorm_tokens(const void *x, const float *gamma, const float *beta, void *y, void *workspace,
                          int B, int T_tokens, int C, int groups, float eps, int dtype, void *stream);
/* Same, writing level l's tokens straight into the concatenated [B, S, 256] token matrix the encoder consumes
 * (`torch.cat(src_flatten, 1)`, deformable_transformer.py:278-285): y points at the level's first token of image 0 and
 * y_batch_stride (elements, >= T*256; 0 = T*256) is the distance between images. */
int dtlr_groupnorm_tokens_strided(const void *x, const float *gamma, const float *beta, void *y, long y_batch_stride,
                                  void *workspace, int B, int T_tokens, int C, int groups, float eps, int dtype, void *stream);
long dtlr_groupnorm_workspace_bytes(int B, int T_tokens);

/* ---------------------------------------------------------------------------------------------
 * Eval-time preprocessing of a batch of RGB uint8 line images, on the device: resize (Pillow's fixed-point bilinear
 * resample, bit-exact) -> /255 -> (x - mean) / std -> zero-padded canvas + padding mask.
 * Replaces: datasets/transforms.py `resize` (:78-109, F.resize on a PIL image), `ToTensor` (:247-249), `Normalize`
 *           (:552-559) as composed for evaluation by datasets/IAM.py:110-112,225-230, and the collate
 *           `nested_tensor_from_tensor_list` (util/misc.py:375-397).
 *   src      device: the images back to back, image b = [h_b, w_b, 3] uint8 at byte offset offsets[b]
 *   offsets  device [B] int64 ; dims device [B][4] int32 = (h, w, oh, ow), (oh, ow) = the resized size the HOST derived
 *            (transforms.py:81-99 `get_size_with_aspect_ratio`; dtlr_amd/transforms.py mirrors it)
 *   Hc, Wc   canvas size = max oh, max ow over the batch ; max_downscale = max over images and axes of in/out (sizes the
 *            filter footprint; > 11 -> DTLR_ESHAPE)
 *   mean3, std3  HOST pointers to 3 floats each (copied into the launch)
 *   canvas   device [B,3,Hc,Wc] fp32, fully written ; mask device [B,Hc,Wc] uint8, 1 = padding, fully written */
int dtlr_preprocess_lines(const unsigned char *src, const long *offsets, const int *dims, int B, int Hc, int Wc,
                          float max_downscale, const float *mean3, const float *std3,
                          float *canvas, unsigned char *mask, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Swin Transformer backbone (selected by `backbone = 'swin_*'`, models/dino/backbone.py:172-205).  Activations are token-major
 * [B,H,W,C]; head_dim is 32 in every reference variant (swin_transformer.py:686-717).
 * dtlr_swin_patch_embed  replaces PatchEmbed.forward (models/dino/swin_transformer.py:416-432): zero padding to a multiple of 4,
 *     4x4/stride-4 convolution, LayerNorm(E).  x [B,3,H,W] fp32 ; w_kE [48,E] fp32 (k = (c*4 + dy)*4 + dx) ; bias/gamma/beta [E] ;
 *     out [B,ceil(H/4),ceil(W/4),E] (F32 / BF16) ; E in {32, 64, 96, 128, 192}.
 * dtlr_swin_window_attn  replaces the attention core of SwinTransformerBlock.forward + WindowAttention.forward (:116-147,
 *     191-236): F.pad to a multiple of the window, torch.roll(-shift), window_partition, q k^T * scale + relative position bias
 *     (+ the 0/-100 shifted-window mask of BasicLayer.forward :357-376), softmax, p v, window_reverse, roll back, un-pad.
 *     qkv [B,H,W,3C] = the qkv projection of norm1(x) ; qkv_bias [3C] fp32 (a padded position projects to the bare biases) ;
 *     rpb [n_heads, ceil16(N), ceil32(N)] fp32 = relative_position_bias_table[relative_position_index], zero padded, N = window^2 ;
 *     out [B,H,W,C] (the input of `proj`).  dtype F32 (exact-fp32 MFMA) or BF16.  window <= 13, C == 32 * n_heads.
 * dtlr_swin_patch_merge  replaces PatchMerging.forward up to its LayerNorm (:262-286): y [B,ceil(H/2),ceil(W/2),4C] =
 *     LayerNorm(cat(x[2i,2j], x[2i+1,2j], x[2i,2j+1], x[2i+1,2j+1])), zeros beyond H/W ; 4C <= 3072. */
int dtlr_swin_patch_embed(const float *x, const float *w_kE, const float *bias, const float *gamma, const float *beta,
                          void *out, int B, int H, int W, int E, float eps, int out_dtype, void *stream);
int dtlr_swin_window_attn(const void *qkv, const float *qkv_bias, const float *rpb, void *out,
                          int B, int H, int W, int C, int n_heads, int window, int shift, int dtype, void *stream);
int dtlr_swin_patch_merge(const void *x, const float *gamma, const float *beta, void *y, int B, int H, int W, int C,
                          float eps, int dtype, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Mask-derived geometry of one forward, ONE launch (a workgroup per row of a level of an image).
 * Replaces: the per-level masks F.interpolate(mask[None].float(), size).bool() (models/dino/backbone.py:103, dino.py:304-307);
 *           get_valid_ratio (models/dino/deformable_transformer.py:239-246); PositionEmbeddingSineHW.forward
 *           (models/dino/position_encoding.py:79-108) + level_embed (deformable_transformer.py:281-285);
 *           TransformerEncoder.get_reference_points (deformable_transformer.py:479-492); the proposal / validity part of
 *           gen_encoder_output_proposals (models/dino/utils.py:31-62).
 *   mask [B,H,W] uint8 (1 = padding) device ; level_hw HOST 8 ints (H_l, W_l) ; level_embed [4,256] fp32 device ;
 *   dim_ty / dim_tx [128] fp32 device = temperature{H,W} ** (2 * (i // 2) / 128) (the table position_encoding.py:95-98 builds)
 *   mask_flat [B,S] uint8, keep [B,S] uint8 (unpadded AND proposal valid), pos [B,S,256] pos_dtype (F32 / BF16, level_embed
 *   added), valid_ratios [B,4,2] fp32 (w,h), enc_ref [B,S,4,2] fp32 (x,y), proposals [B,S,4] fp32 (logits; +inf where padded
 *   or invalid).  S = sum H_l W_l; every output element is written. */
int dtlr_geometry(const unsigned char *mask, int B, int H, int W, const int *level_hw,
                  const float *level_embed, const float *dim_ty, const float *dim_tx, int pos_dtype,
                  unsigned char *mask_flat, unsigned char *keep, void *pos, float *valid_ratios,
                  float *enc_ref, float *proposals, void *stream);

/* ---------------------------------------------------------------------------------------------
 * ResNet stem convolution: 7x7 / stride 2 / pad 3, 3 -> 64 channels, on the bf16 matrix cores, reading the
 * NCHW fp32 image directly (bf16-rounded operands, fp32 accumulate) and writing NHWC bf16.
 * Replaces: torchvision resnet50 `conv1` (+ the FrozenBN scale folded into the weights, backbone.py:62-72) as run by
 *           IntermediateLayerGetter (backbone.py:94-106); the FrozenBN shift + ReLU + max-pool follow in
 *           dtlr_maxpool3x3s2_nhwc.
 *   x [B,3,H,W] fp32 device ; y [B, (H-1)/2+1, (W-1)/2+1, 64] bf16 device
 *   wfrag: device copy of the 24 KB fragment-major weight image built by dtlr_stem_pack_weights (a HOST-side helper:
 *          both of its pointers are host memory; w_oihw = conv1.weight * bn_scale, [64,3,7,7] fp32). */
int dtlr_stem_pack_weights(const float *w_oihw_host, unsigned short *wfrag_host /* [4*6*64*8] */);
int dtlr_stem_conv7x7(const float *x, const void *wfrag, void *y, int B, int H, int W, int out_dtype, void *stream);
/* conv1 + bn1 (folded: weights carry the scale, `bias` [64] fp32 the shift) + ReLU + 3x3 / stride-2 / pad-1 max-pool in one kernel
 * (torchvision resnet50 stem as run by models/dino/backbone.py:97-106): x [B,3,H,W] fp32 NCHW -> y [B,Hp,Wp,64] 16-bit NHWC with
 * Hp = floor((Ho - 1) / 2) + 1, Ho = floor((H - 1) / 2) + 1 (same for W).  Bit-identical to dtlr_stem_conv7x7 + dtlr_maxpool3x3s2_nhwc. */
int dtlr_stem_conv7x7_pool(const float *x, const void *wfrag, const float *bias, void *y, int B, int H, int W, int out_dtype, void *stream);
/* The same convolution in exact fp32 (the parity engine; direct convolution on the vector ALUs, patch + weights in LDS).
 *   wk: device [147][64] fp32, k-major image of conv1.weight * bn_scale (k = (ci*7 + kh)*7 + kw) ; y [B,Ho,Wo,64] fp32 NHWC. */
int dtlr_stem_conv7x7_f32(const float *x, const float *wk, float *y, int B, int H, int W, void *stream);
/* The same convolution for the split-fp32 engine (round 4): fp32 image and output, the image and the weights as fp16 hi + lo halves, three fp16
 * MFMAs per product (stem_conv7x7's MFMA formulation; 4x faster than the direct fp32 kernel above).  wfrag_hi / wfrag_lo: the fragment
 * images dtlr_stem_pack_weights of the fp16 build (libdtlr_hip_f16.so) writes for fp16(w) and for w - fp16(w). */
int dtlr_stem_conv7x7_f32s(const float *x, const void *wfrag_hi, const void *wfrag_lo, float *y, int B, int H, int W, void *stream);

/* 3x3 / stride 2 / pad 1 max pooling on NHWC, optionally preceded by a per-channel bias and ReLU:
 *     y = maxpool(relu(x + bias))      (bias NULL: no bias; relu 0: no ReLU)
 * Replaces: torchvision resnet50 `maxpool` as run through IntermediateLayerGetter (backbone.py:94,98), and with
 *           bias/relu also the stem's FrozenBN shift (backbone.py:62-72, folded) + `relu` between conv1 and maxpool. */
int dtlr_maxpool3x3s2_nhwc(const void *x, void *y, const float *bias, int relu, int B, int H, int W, int C, int dtype, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Indices of the k largest scores of every row, in descending score order; equal scores keep the lower
 * index first (deterministic; torch.topk leaves ties unspecified).
 * Replaces: torch.topk(enc_outputs_class_unselected.max(-1)[0], num_queries, dim=1)[1]
 *           (models/dino/deformable_transformer.py:345).   scores [B,S] fp32 -> idx_out [B,k] int64.
 */
int dtlr_topk_rows(const float *scores, long *idx_out, int B, int S, int k, void *stream);

/* The gathers that follow the two-stage top-k, one launch.
 * Replaces: torch.gather of output_memory / output_proposals at topk_proposals and the sigmoid of the gathered proposals
 *           (models/dino/deformable_transformer.py:347-356).
 *   om        [B,S,768] bf16 = the [hi|lo|hi] image of dtlr_proj_ln_split_bf16 (dtype BF16) or [B,S,256] fp32 (dtype F32)
 *   proposals [B,S,4] fp32 ; idx [B,k] int64
 *   sel_raw   [B,k,768] bf16 / [B,k,256] fp32 : the gathered rows ; sel_x [B,k,256] bf16 = bf16(hi + lo) (BF16 only, else NULL)
 *   prop_sel  [B,k,4] fp32 gathered proposal logits ; init_box [B,k,4] fp32 = sigmoid(prop_sel) */
int dtlr_two_stage_gather(const void *om, const float *proposals, const long *idx, void *sel_raw, void *sel_x,
                          float *prop_sel, float *init_box, int B, int S, int k, int dtype, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Blank/argmax decoder, one workgroup per line.
 * Replaces: evaluation.convert_output_to_pred, blank branch (evaluation.py:116-158) == the blank
 *           construction of SetCriterion.loss_CTC (models/dino/dino.py:466-502) followed by
 *           engine.convert_output_to_pred (engine.py:511-530): sort queries by box cx; p = sigmoid(logits);
 *           s = sum_c p; blank = 1 - s if s < 1 - eps else eps (then p <- (1-eps) p / s); argmax over
 *           [blank | p]; drop blanks; no repeat collapse.
 *   logits [B,nq,C] fp32, boxes [B,nq,4] fp32 (cx first) -> labels [B,nq] int32 left-packed, -1 padded;
 *   lengths [B] int32; -1 for a line with a query that has ANY non-finite logit (NaN, +inf or -inf: on the fp16 / split engines an
 *   activation beyond 65504) -- flagged on the device, no host synchronisation.   eps = 0.03/C (evaluation.py:141) or 0.003 (dino.py:491).
 */
int dtlr_decode_blank(const float *logits, const float *boxes, int *labels, int *lengths,
                      int B, int nq, int C, float eps, void *stream);

/* Evaluation-time CTC loss value, per line (negative log-likelihood; inf -> 0 as `zero_infinity=True`).
 * Replaces: the forward of `SetCriterion.loss_CTC` (models/dino/dino.py:457-551) as engine.evaluate_CTC calls it
 *           (engine.py:381): queries sorted by box cx, sigmoid, blank channel with eps (0.003 there), a filler step
 *           [1, filler, ...] after every query (T = 2 nq), nn.CTCLoss(blank=0) against labels + 1.  The caller applies the
 *           'mean' reduction: mean_b( nll[b] / max(target_lengths[b], 1) ).
 *   logits [B,nq,C] fp32 ; boxes [B,nq,4] fp32 ; targets [B,Lmax] int32 = label + 1, rows padded arbitrarily ;
 *   target_lengths [B] int32 ; nll [B] fp32 out ; workspace >= B*nq floats ; all device pointers.
 *   max_target_length = max(target_lengths) as known to the HOST (sizes the workgroup: one thread per state of the
 *   blank-extended sequence, 2 L + 1 <= 1024, else DTLR_ESHAPE). */
int dtlr_ctc_loss_interleaved(const float *logits, const float *boxes, const int *targets, const int *target_lengths,
                              float *nll, float *workspace, int B, int nq, int C, int Lmax, int max_target_length,
                              float eps, float filler, void *stream);

/* ---------------------------------------------------------------------------------------------
 * CTC-style emissions for the n-gram re-scoring path.
 * Replaces: get_new_pred_logits (ngram/prediction_helpers.py:5-46): queries sorted by box cx, p = scale * sigmoid(logits),
 *           blank channel first with the rule of SetCriterion.loss_CTC (models/dino/dino.py:466-502; eps = 0.003 there).
 * logits [B,nq,C] fp32, boxes [B,nq,4] fp32 -> out [B,nq,C+1] fp32 (row r = the r-th query in reading order).
 * workspace: dtlr_blank_emissions_workspace_bytes(B, nq) bytes of device memory.  Asynchronous on `stream`. */
int dtlr_blank_emissions(const float *logits, const float *boxes, float *out, float *workspace,
                         int B, int nq, int C, float scale, float eps, void *stream);
long dtlr_blank_emissions_workspace_bytes(int B, int nq);

/* Greedy non-maximum suppression, batched: image b keeps, in descending score order (equal scores: lower index first), every
 * box whose IoU with an already kept box is <= iou_threshold.
 * Replaces: `torchvision.ops.nms(b, s, iou_threshold)` as PostProcess calls it per image (models/dino/dino.py:1029-1033), i.e.
 *           the NMS decoder of evaluation.py:94-115 (IAM / READ / RIMES scripts: --NMS 0.5 --TH 0.3).
 *   boxes [B,n,4] fp32 (x0,y0,x1,y1) ; scores [B,n] fp32 ; keep [B,n] int64: kept ORIGINAL indices, -1 padded ; counts [B] int32.
 *   n <= 1024 (DTLR_ESHAPE otherwise: the suppression bit-matrix lives in LDS). */
int dtlr_nms(const float *boxes, const float *scores, float iou_threshold, long *keep, int *counts, int B, int n, void *stream);

/* k largest of each row of a [B, n] fp32 matrix that is too long for LDS, descending, equal values: lower index first.
 * Replaces: `torch.topk(prob.view(B, -1), num_select, dim=1)` of PostProcess (models/dino/dino.py:1000-1006), with the sigmoid
 *           folded in (apply_sigmoid: the selection runs on the logits, values are returned as sigmoid(logit)).
 *   x [B,n] fp32 ; values [B,k] fp32 ; idx_out [B,k] int64 (flat positions: box = idx / C, label = idx % C) ; k <= 8192. */
int dtlr_topk_flat(const float *x, float *values, long *idx_out, int B, long n, int k, int apply_sigmoid, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DTLR_HIP_H */
