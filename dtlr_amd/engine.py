"""The MI355X forward of DTLR: ResNet-50 -> input_proj -> 6x deformable encoder -> two-stage query
selection -> 6x decoder -> heads.  Mirrors the reference call stack stage by stage (SURVEY.md
section 3.1; every method cites the reference lines it replaces) but is laid out for the GPU:

  * activations are NHWC / token-major ([B, H*W, C]) end to end, so a backbone map IS the token
    matrix the transformer consumes -- no flatten/transpose copies (deformable_transformer.py:278-285);
  * FrozenBatchNorm is folded into the conv weights/bias once at pack time (backbone.py:62-72);
  * sampling_offsets and attention_weights are one fused 256->384 projection (ms_deform_attn.py:97-98);
  * the six decoder cross-attention value projections of `memory` are batch-first and need no
    seq-first transposes (deformable_transformer.py:394-403 transposes everything);
  * position embeddings, reference points and proposals depend only on the padding masks and are
    computed once per forward (the reference recomputes reference points per encoder call);
  * aux heads for decoder layers 0..4 are skipped unless asked for (dino.py:339-354 computes all 6).

All compute goes through dtlr_amd.ops (HIP kernels / ROCm libraries on the GPU).  `dtype` is the
storage/compute type of activations and GEMM operands (float32 = parity path; bfloat16 / float16 = the 16-bit
MFMA paths, served by libdtlr_hip.so / libdtlr_hip_f16.so); selection scores, softmax, normalisation statistics and box arithmetic are always fp32.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch

from . import ops
from .config import DTLRConfig


def _fold_bn(sd, conv_key: str, bn_prefix: str):
    """conv -> FrozenBatchNorm2d (models/dino/backbone.py:62-72) as one conv with bias."""
    w = sd[conv_key].float()
    scale = sd[bn_prefix + ".weight"].float() * torch.rsqrt(sd[bn_prefix + ".running_var"].float() + 1e-5)
    bias = sd[bn_prefix + ".bias"].float() - sd[bn_prefix + ".running_mean"].float() * scale
    return w * scale.view(-1, 1, 1, 1), bias


class DTLREngine:
    def __init__(self, cfg: DTLRConfig, state_dict: Dict[str, torch.Tensor], device="cuda:0",
                 dtype: torch.dtype = torch.float32, split: bool = False):
        """`split=True` (with dtype float32): the parity-grade engine at the 16-bit matrix rate / 3 -- activations, residual streams,
        normalisation inputs and the box chain stay fp32 exactly as in the fp32 engine, but every GEMM / convolution weight is packed
        as a split fp16 hi + lo image (ops.split_pack) and multiplied by the DTLR_F32S kernels (three fp16 MFMAs per product, fp32
        accumulation: operands carried to 22 bits for |x| >= 2^-3 and to 2^-25 absolute below that -- ~2^-21 relative at the synthetic weights'
        scale, ~2^-18 at a trained checkpoint's ~1e-2 -- instead of the 16-bit engines' 2^-9 / 2^-12 per rounding point).
        Range guard: the fp16 hi halves saturate at 65504; the engine checks the backbone's output maps on its FIRST forward only
        (one host read; `check_activation_range()` re-arms it).  A later batch that overflows does not pass silently either: inf becomes NaN
        in the next normalisation, and the blank decoder flags every line with non-finite logits on the device (length -1 in the
        record, evaluation.records_to_lists raises) -- without a host synchronisation on the step."""
        cfg.validate()
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.split = bool(split)
        if self.split and dtype != torch.float32:
            raise ValueError("DTLREngine(split=True) is a float32 engine")
        if self.device.type != "cuda":
            raise RuntimeError("DTLREngine runs on the GPU only (no CPU path)")
        ops.require_cuda(torch.empty(0, device=self.device))
        self.w: Dict[str, torch.Tensor] = {}
        self._ffn_f32: Dict[str, torch.Tensor] = {}
        self._k256s_ok = set()                      # split engine: the [256, 256] encoder projections that run through dtlr_gemm_k256s
        self._pack(state_dict)
        self._shape_cache: Dict[tuple, dict] = {}
        self._level_cache: Dict[tuple, tuple] = {}
        self.use_lds_msda = True       # encoder MSDA with LDS-staged windows (False: gather kernel)
        self.use_fused_ffn = True      # bf16: linear1+ReLU+linear2+residual+LayerNorm in one kernel (False: two GEMMs + LN)
        self.use_k256 = True   # bf16: weight-resident streaming kernel for the K = 256 projections over all tokens
        self.use_pln_k256 = True
        self.use_k256s = True  # split: the same for the [256, 256] encoder projections (value_proj; output_proj + residual + norm1)
        self.use_ffn32 = True
        self.ffn32_tail = True          # the encoder FFN's last partial round of 256 workgroups on the 16x16x32 kernel (round 4); False: dtlr_ffn32_bf16 over all rows
        self.pln_k256_min_rows = 16384
        self.use_kres = True
        self.use_kres_narrow = True
        self.msda_auto = True      # per-layer choice LDS-window / gather kernel, calibrated once per canvas shape (see _msda_mode)
        # cost model of the encoder MSDA kernels, in units of the gather kernel's time (profiles/r03_msda_offset_halo_sweep_v1.json, B = 32,
        # 128x2048): LDS-window kernel with a halo of 8 / 16 / 24 columns = base + 4.8 sqrt(far fraction); gather kernel = 1
        self.msda_halo_base = {8: 0.51, 16: 0.635, 24: 0.71}
        self.msda_far_slope = 4.8
        self._msda_state = {}      # (layer, canvas shape) -> {"mode", "far"}; keyed by nothing that depends on the data or the call history
        self._msda_calibrating = None
        self.use_k256_small = True   # ... and for the encoder's output projection + LayerNorm
        self.use_stem_pool = True         # 16-bit: stem convolution + FrozenBN shift + ReLU + max-pool in one kernel
        self.use_dec_query_stage = True   # 16-bit: a decoder layer's query stage (sine, ref_point_head, q | k, v) in one launch
        self.use_l1_chain = True          # 16-bit: layer1's 1x1 convolutions chained (shortcut conv as extra K columns; tail + next conv1 in one launch)
        self.use_l1_chain_out = True      #         ... including the last tail -> layer2.0.conv1
        self.use_l2_cat = True            # 16-bit: layer2.0's strided shortcut convolution as extra K columns of its tail GEMM
        self.head_ts_min_classes = 1024   # 16-bit engines: class heads with at least this many classes run on the token-stationary kernel (dtlr_head_ts)
        self.head_ts_scores = True        # ... and the two-stage selection scores (row maximum: no logits leave the chip) for EVERY charset: 142 -> 74 us at 166 classes
        self.use_ow_resbcast = True       # fp32 / split: the encoder's [offsets | logits] projection as src W^T + (pos W^T + b), unpadded batches
        # split: value_proj(memory) of the six decoder layers as six slices of one launch (dtlr_gemm_k256s_multi): 474-534 us against 565
        # through the tiled GEMM.  The encoder form -- value_proj + [offsets | logits] of `src` as three slices of one launch -- is built and
        # tested but OFF: measured 302-314 us against 89 + 192 for the two launches it replaces (tools/experiments/k256s_multi_bench.py): the
        # slices' token tiles do NOT meet in the XCD's L2, so every slice streams `src` again (gemm_k256s.hip's notes).
        self.use_k256s_multi = True
        self.use_k256s_multi_enc = False
        self._range_check_pending = (dtype == torch.float16) or self.split      # engines whose operands are fp16: see forward()
        self._ws_streams = set()           # streams whose library workspace this engine has pre-sized (ops.workspace_reserve)
        # Round 6: independent launches on side HIP streams (fork / join with stream waits; buffers that cross streams are allocated on
        # the caller's stream BEFORE the fork and outlive the join, so the caching allocator never hands a block to another stream early):
        #   * input_proj + GroupNorm of level 0 (on C3) under layer3, of level 1 (on C4) under layer4;
        #   * value_proj(memory) of the six decoder layers under the two-stage selection (top-k: 32 workgroups; gathers; box MLP);
        #   * the shortcut (`downsample`) convolution of a ResNet layer's first bottleneck under its conv1 -> conv2.
        # Same kernels, same arguments, same per-kernel arithmetic: bit-identical to the one-stream schedule (GPU test, also under HIP-graph
        # capture).  MEASURED AND OFF: same-box A/B at B = 32 (profiles/r06_streams_ab_c4.txt): bf16 8.78 -> 8.84 ms, f32s 21.50 -> 21.68 ms
        # per step -- every kernel of the path already fills the chip (>= 256 workgroups, one or two per CU by LDS), so a side stream's
        # workgroups only queue behind the main stream's, and the five fork / join pairs cost more than the small grids (top-k, gathers) free.
        self.overlap_streams = False
        self._side_streams = [torch.cuda.Stream(device=self.device) for _ in range(3)]      # created (and their library workspace pre-sized) HERE:
        for st in self._side_streams:                                                       # nothing is created or allocated under a later capture
            with torch.cuda.stream(st):
                ops.workspace_reserve(self.dtype, 16 << 20)

    def check_activation_range(self) -> None:
        """re-arm the fp16-range check of the f16 / f32s engines for the next forward (it runs on the first forward only: a host sync)"""
        self._range_check_pending = (self.dtype == torch.float16) or self.split

    # ------------------------------------------------------------------------------ packing
    def _put(self, name, t, dtype=None):
        self.w[name] = t.to(device=self.device, dtype=dtype or self.dtype).contiguous()

    def _gw(self, t):
        """a GEMM / convolution weight in the form the engine's kernels multiply: the engine dtype, or its split image."""
        t = t.to(device=self.device, dtype=self.dtype).contiguous()
        return ops.split_pack(t) if self.split else t

    def _put_conv(self, name, w, b):
        elem = 2 if self.dtype in ops.H16 else 4
        if w.shape[2] == 1 and w.shape[3] == 1:      # 1x1 conv == linear over NHWC pixels: [Cout, Cin]
            self.w[name + ".w"] = self._gw(w.flatten(1))
        elif (w.shape[1] * elem) % 128 == 0:         # implicit-GEMM HIP kernel: [Cout, KH, KW, Cin]
            self.w[name + ".w"] = self._gw(w.permute(0, 2, 3, 1))
        else:
            raise ValueError(f"_put_conv({name}): Cin = {w.shape[1]} does not fit the implicit-GEMM kernel (the stem has its own)")
        self.w[name + ".b"] = b.to(device=self.device, dtype=torch.float32).contiguous()

    def _put_linear(self, name, w, b):
        if self.split and (name.endswith((".ff1", ".ff2", ".attn.out", ".sa.out")) or (name.startswith("enc") and name.endswith((".attn.value", ".attn.ow")))
                           or name in ("enc_output", "dec.value_all")):
            # the fused split FFN and the weight-resident K = 256 projections pack their own images from the fp32 weights
            self._ffn_f32[name] = w.to(device=self.device, dtype=torch.float32).contiguous()
            if not name.endswith((".ff1", ".ff2")) and tuple(w.shape) == (256, 256):
                self._k256s_ok.add(name)
        self.w[name + ".w"] = self._gw(w)
        self._put(name + ".b", b, torch.float32)          # biases enter the GEMM epilogue in fp32

    def _pack_swin(self, sd):
        """backbone.0.* of a Swin backbone (models/dino/swin_transformer.py): GEMM weights in the engine dtype, LayerNorm / bias /
        relative-position tensors fp32; the dense per-head bias of every block is built here, once."""
        cfg, f32 = self.cfg, torch.float32
        sp = cfg.swin_params()
        E, ws = sp["embed_dim"], sp["window_size"]
        # 16-bit engines: the MFMA GEMM consumes K in 128-byte slabs (64 elements).  swin_T's first stage has C = 96: its three K = 96
        # projections (qkv, proj, fc1) get zero-padded weight columns here and a zero-padded activation copy at run time (_padk).
        self._swin_kpad = (lambda k: -(-k // 64) * 64) if self.dtype in ops.H16 else (lambda k: k)
        b = "backbone.0."
        self._put("swin.pe.w", sd[b + "patch_embed.proj.weight"].float().reshape(E, 48).t(), f32)      # [48, E], k-major
        self._put("swin.pe.b", sd[b + "patch_embed.proj.bias"], f32)
        self._put("swin.pe.ln.w", sd[b + "patch_embed.norm.weight"], f32)
        self._put("swin.pe.ln.b", sd[b + "patch_embed.norm.bias"], f32)
        for i in range(4):
            for j in range(sp["depths"][i]):
                p, q = f"{b}layers.{i}.blocks.{j}.", f"swin.{i}.{j}."
                for nm in ("norm1", "norm2"):
                    self._put(q + nm + ".w", sd[p + nm + ".weight"], f32)
                    self._put(q + nm + ".b", sd[p + nm + ".bias"], f32)
                for nm, key in (("qkv", "attn.qkv"), ("proj", "attn.proj"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")):
                    wt = sd[p + key + ".weight"]
                    kp = self._swin_kpad(wt.shape[1])
                    if kp != wt.shape[1]:
                        wt = torch.nn.functional.pad(wt, (0, kp - wt.shape[1]))
                    self._put_linear(q + nm, wt, sd[p + key + ".bias"])
                self.w[q + "rpb"] = ops.swin_dense_bias(sd[p + "attn.relative_position_bias_table"].to(self.device), ws)
            if i < 3:
                p, q = f"{b}layers.{i}.downsample.", f"swin.{i}.merge."
                self._put(q + "ln.w", sd[p + "norm.weight"], f32)
                self._put(q + "ln.b", sd[p + "norm.bias"], f32)
                wt = sd[p + "reduction.weight"]
                kp = self._swin_kpad(wt.shape[1])
                self.w[q + "w"] = self._gw(wt if kp == wt.shape[1] else torch.nn.functional.pad(wt, (0, kp - wt.shape[1])))
            if i in cfg.return_interm_indices:
                self._put(f"swin.norm{i}.w", sd[f"{b}norm{i}.weight"], f32)
                self._put(f"swin.norm{i}.b", sd[f"{b}norm{i}.bias"], f32)

    def backbone_swin(self, x_nchw) -> List[torch.Tensor]:
        """SwinTransformer.forward (models/dino/swin_transformer.py:633-673) -> the NHWC maps of return_interm_indices."""
        cfg, w = self.cfg, self.w
        sp = cfg.swin_params()
        ws = sp["window_size"]
        x = ops.swin_patch_embed(x_nchw, w["swin.pe.w"], w["swin.pe.b"], w["swin.pe.ln.w"], w["swin.pe.ln.b"], self.dtype)

        def padk(t, wname):                       # zero-pad the last dim to the packed weight's K (16-bit engines, C = 96: see _pack_swin)
            kp = w[wname].shape[1]
            if kp == t.shape[-1]:
                return t
            tp = t.new_zeros(t.shape[:-1] + (kp,))
            tp[..., : t.shape[-1]] = t
            return tp
        outs = []
        for i in range(4):
            nh = sp["num_heads"][i]
            for j in range(sp["depths"][i]):
                q = f"swin.{i}.{j}."
                # x = x + proj(window_attention(norm1(x)))  (swin_transformer.py:191-243): shift / partition / padding / reverse are
                # index arithmetic inside the attention kernel; the residual add is the projection's epilogue
                y = ops.layernorm(x, w[q + "norm1.w"], w[q + "norm1.b"], 1e-5)
                qkv = ops.linear(padk(y, q + "qkv.w"), w[q + "qkv.w"], w[q + "qkv.b"])
                a = ops.swin_window_attn(qkv, w[q + "qkv.b"], w[q + "rpb"], nh, ws, 0 if j % 2 == 0 else ws // 2)
                x = ops.linear(padk(a, q + "proj.w"), w[q + "proj.w"], w[q + "proj.b"], residual=x)
                # x = x + fc2(gelu(fc1(norm2(x))))  (:244-247)
                y = ops.layernorm(x, w[q + "norm2.w"], w[q + "norm2.b"], 1e-5)
                hdn = ops.linear(padk(y, q + "fc1.w"), w[q + "fc1.w"], w[q + "fc1.b"], relu=3)
                x = ops.linear(padk(hdn, q + "fc2.w"), w[q + "fc2.w"], w[q + "fc2.b"], residual=x)
            if i in cfg.return_interm_indices:
                outs.append(ops.layernorm(x, w[f"swin.norm{i}.w"], w[f"swin.norm{i}.b"], 1e-5))
            if i < 3:
                q = f"swin.{i}.merge."
                x = ops.linear(padk(ops.swin_patch_merge(x, w[q + "ln.w"], w[q + "ln.b"]), q + "w"), w[q + "w"], None)
        return outs

    def _pack(self, sd):
        cfg, f32 = self.cfg, torch.float32
        if cfg.is_swin:
            self._pack_swin(sd)
        else:
            self._pack_resnet(sd)
        self._pack_rest(sd)

    def _pack_resnet(self, sd):
        cfg = self.cfg
        b = "backbone.0.body."
        w1, b1 = _fold_bn(sd, b + "conv1.weight", b + "bn1")
        self.w["conv1.b"] = b1.to(device=self.device, dtype=torch.float32).contiguous()
        if self.dtype in ops.H16:               # bf16 engine: MFMA stem kernel, weights as a fragment image in registers
            self.w["conv1.frag"] = ops.stem_pack_weights(w1, self.dtype).to(self.device)
        elif self.split:                               # split engine: the MFMA stem on fp16 hi + lo fragment images
            fh, fl = ops.stem_pack_weights_split(w1)
            self.w["conv1.fh"], self.w["conv1.fl"] = fh.to(self.device), fl.to(self.device)
        else:                                          # fp32 engine: exact-fp32 direct-convolution stem kernel, k-major weights
            self.w["conv1.wk"] = ops.stem_pack_weights_f32(w1).to(self.device)
        for li, nblocks in enumerate(cfg.backbone_blocks, start=1):
            for bi in range(nblocks):
                p = f"{b}layer{li}.{bi}."
                q = f"l{li}.{bi}."
                for c in (1, 2, 3):
                    self._put_conv(f"{q}c{c}", *_fold_bn(sd, f"{p}conv{c}.weight", f"{p}bn{c}"))
                if bi == 0:
                    self._put_conv(q + "ds", *_fold_bn(sd, p + "downsample.0.weight", p + "downsample.1"))

    def _pack_rest(self, sd):
        cfg, f32 = self.cfg, torch.float32
        for l in range(cfg.num_feature_levels):
            w, bias = sd[f"input_proj.{l}.0.weight"].float(), sd[f"input_proj.{l}.0.bias"].float()
            if l < cfg.num_feature_levels - 1:
                self._put_linear(f"ip{l}", w.flatten(1), bias)            # 1x1 conv == linear on NHWC tokens
            else:
                self._put_conv(f"ip{l}", w, bias)
            self._put(f"ip{l}.gn.w", sd[f"input_proj.{l}.1.weight"], f32)
            self._put(f"ip{l}.gn.b", sd[f"input_proj.{l}.1.bias"], f32)
        t = "transformer."
        self._put("level_embed", sd[t + "level_embed"], f32)

        def msda_pack(dst, src):
            self._put_linear(dst + ".ow", torch.cat([sd[src + ".sampling_offsets.weight"], sd[src + ".attention_weights.weight"]], 0),
                             torch.cat([sd[src + ".sampling_offsets.bias"], sd[src + ".attention_weights.bias"]], 0))
            self._put_linear(dst + ".value", sd[src + ".value_proj.weight"], sd[src + ".value_proj.bias"])
            self._put_linear(dst + ".out", sd[src + ".output_proj.weight"], sd[src + ".output_proj.bias"])

        def norm_pack(dst, src):
            self._put(dst + ".w", sd[src + ".weight"], f32)
            self._put(dst + ".b", sd[src + ".bias"], f32)

        for n in range(cfg.enc_layers):
            p, q = f"{t}encoder.layers.{n}.", f"enc{n}."
            msda_pack(q + "attn", p + "self_attn")
            norm_pack(q + "norm1", p + "norm1")
            norm_pack(q + "norm2", p + "norm2")
            self._put_linear(q + "ff1", sd[p + "linear1.weight"], sd[p + "linear1.bias"])
            self._put_linear(q + "ff2", sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        C = cfg.hidden_dim
        for n in range(cfg.dec_layers):
            p, q = f"{t}decoder.layers.{n}.", f"dec{n}."
            msda_pack(q + "attn", p + "cross_attn")
            for k in ("norm1", "norm2", "norm3"):
                norm_pack(q + k, p + k)
            W, Bi = sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]
            self._put_linear(q + "sa.qk", W[:2 * C], Bi[:2 * C])          # q,k share the input tgt+pos
            self._put_linear(q + "sa.v", W[2 * C:], Bi[2 * C:])
            self._put_linear(q + "sa.out", sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
            self._put_linear(q + "ff1", sd[p + "linear1.weight"], sd[p + "linear1.bias"])
            self._put_linear(q + "ff2", sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        self._put_linear("dec.value_all",
                         torch.cat([sd[f"{t}decoder.layers.{n}.cross_attn.value_proj.weight"] for n in range(cfg.dec_layers)], 0),
                         torch.cat([sd[f"{t}decoder.layers.{n}.cross_attn.value_proj.bias"] for n in range(cfg.dec_layers)], 0))
        norm_pack("dec.norm", t + "decoder.norm")
        for i in range(2):
            self._put_linear(f"dec.rph{i}", sd[f"{t}decoder.ref_point_head.layers.{i}.weight"], sd[f"{t}decoder.ref_point_head.layers.{i}.bias"])
        self._put("tgt_embed", sd[t + "tgt_embed.weight"])
        self._put_linear("enc_output", sd[t + "enc_output.weight"], sd[t + "enc_output.bias"])
        norm_pack("enc_output_norm", t + "enc_output_norm")
        # heads: selection scores and boxes stay fp32 (discrete selections amplify rounding)
        for i in range(3):
            self._put(f"enc_bbox{i}.w", sd[f"{t}enc_out_bbox_embed.layers.{i}.weight"], f32)
            self._put(f"enc_bbox{i}.b", sd[f"{t}enc_out_bbox_embed.layers.{i}.bias"], f32)
            self._put(f"bbox{i}.w", sd[f"bbox_embed.0.layers.{i}.weight"], f32)
            self._put(f"bbox{i}.b", sd[f"bbox_embed.0.layers.{i}.bias"], f32)
            if i < 2 and self.split:                   # the hidden layers are GEMMs; the 256 -> 4 layer is read by box_head_refine as fp32
                self.w[f"enc_bbox{i}.w"] = ops.split_pack(self.w[f"enc_bbox{i}.w"])
                self.w[f"bbox{i}.w"] = ops.split_pack(self.w[f"bbox{i}.w"])
            if i < 2 and self.dtype in ops.H16:
                # bf16 engine: the two hidden layers of the box MLPs run on the bf16 MFMA path (their input, the decoder
                # state, is bf16 already); accumulation, the 256->4 output layer and all box arithmetic stay fp32.
                # (As fp32-MFMA GEMMs these 16 launches of M = 28800 cost 43 us each: 0.7 ms of a 14.4 ms step.)
                self._put(f"bbox{i}.wh", sd[f"bbox_embed.0.layers.{i}.weight"])
                self._put(f"enc_bbox{i}.wh", sd[f"{t}enc_out_bbox_embed.layers.{i}.weight"])
                if i == 1:                             # second layer also chunk-major for the one-launch MLP kernel
                    self.w["bbox1.wp"] = ops.ffn_pack_w2(self.w["bbox1.wh"])
                    self.w["enc_bbox1.wp"] = ops.ffn_pack_w2(self.w["enc_bbox1.wh"])
        self._put("enc_class.w", sd[t + "enc_out_class_embed.weight"], f32)
        self._put("enc_class.b", sd[t + "enc_out_class_embed.bias"], f32)
        self._put("class.w", sd["class_embed.0.weight"], f32)
        self._put("class.b", sd["class_embed.0.bias"], f32)
        if self.split:
            self.w["enc_class.w"] = ops.split_pack(self.w["enc_class.w"])
            self.w["class.w"] = ops.split_pack(self.w["class.w"])
        self.num_classes = int(sd["class_embed.0.weight"].shape[0])

    # ------------------------------------------------------------------------------ stages
    def _conv(self, name, x, stride, padding, relu=False, residual=None):
        """NHWC convolution with the folded FrozenBN bias, optional residual add, then ReLU."""
        w = self.w[name + ".w"]
        if w.dim() == 2:                              # 1x1: the MFMA GEMM with the whole tail fused
            if stride != 1:                           # strided 1x1 (downsample): the implicit-GEMM kernel gathers the pixels itself
                return ops.conv2d_nhwc(x, w.view(w.shape[0], 1, 1, w.shape[1]), self.w[name + ".b"], stride, 0, relu, residual)
            M = x.numel() // x.shape[-1]
            if self.use_kres and ops.kres_supported(M, w.shape[0], w.shape[1], x.dtype) and (residual is None or w.shape[0] >= 256) \
                    and (w.shape[0] >= 256 or (self.use_kres_narrow and w.shape[1] == 256)):     # narrow outputs: the 256 -> 64 / 128 reductions only
                if name + ".wk" not in self.w:                # weight-resident streaming form (bottleneck tail / layer1 downsample)
                    self.w[name + ".wk"] = ops.kres_pack(w)
                return ops.gemm_kres(x, self.w[name + ".wk"], w.shape[0], self.w[name + ".b"], residual, relu=bool(relu))
            return ops.linear(x, w, self.w[name + ".b"], relu=(2 if relu else 0), residual=residual)
        return ops.conv2d_nhwc(x, w, self.w[name + ".b"], stride, padding, relu, residual)

    def _lin(self, name, x, relu=False, residual=None, a2=None, row_mask=None, out_dtype=None):
        w = self.w[name + ".w"]
        if self.use_k256_small and x.dtype in ops.H16 and not relu and residual is None and a2 is None and out_dtype in (None, x.dtype) \
                and w.shape[1] == 256 and w.shape[0] in (256, 384) and x.numel() // 256 >= 16384 and x.is_contiguous():
            return ops.gemm_k256(x, self._k256w(name), w.shape[0], self.w[name + ".b"], row_mask=row_mask)     # plain K = 256 projections
        return ops.linear(x, w, self.w[name + ".b"], relu, residual, a2, row_mask, out_dtype)

    def _ln(self, name, x, residual=None):
        return ops.layernorm(x, self.w[name + ".w"], self.w[name + ".b"], 1e-5, residual)

    def _proj_ln(self, proj, norm, a, residual):
        """output projection of an attention block + residual + post-norm (deformable_transformer.py:810-815, 847-870)."""
        w = self.w
        if self.split and self.use_k256s and proj in self._k256s_ok and a.shape[-1] == 256 and a.numel() // 256 >= self.pln_k256_min_rows:
            return ops.gemm_k256s(a, self._k256sw(proj), w[proj + ".b"], residual=residual, ln_w=w[norm + ".w"], ln_b=w[norm + ".b"])
        if self.use_pln_k256 and a.dtype in ops.H16 and a.shape[-1] == 256 and a.numel() // 256 >= self.pln_k256_min_rows:
            if proj + ".wk" not in w:                          # large M (the encoder): weight-resident streaming form
                w[proj + ".wk"] = ops.proj_ln_k256_pack(w[proj + ".w"])
            return ops.proj_ln_k256(a, w[proj + ".wk"], w[proj + ".b"], residual, w[norm + ".w"], w[norm + ".b"])
        if self.use_fused_ffn and a.dtype in ops.H16 and a.shape[-1] == 256:
            if proj + ".wp" not in w:                          # fragment-major copy of the projection weight, packed once
                w[proj + ".wp"] = ops.proj_pack_w(w[proj + ".w"])
            return ops.proj_ln(a, w[proj + ".wp"], w[proj + ".b"], residual, w[norm + ".w"], w[norm + ".b"])
        return self._ln(norm, self._lin(proj, a), residual=residual)

    def _head_ts(self, name):
        """(image, padded bias) of class head `name` for the token-stationary kernel, packed once (ops.head_ts_pack)"""
        key = name + ".ts"
        if key not in self.w:
            self.w[key] = ops.head_ts_pack(self.w[name + ".w"], self.w[name + ".b"], self.dtype)
        return self.w[key]

    def _class_head(self, hs):
        """class_embed on decoder states (models/dino/dino.py:349-352), fp32 logits.  bf16 engine: the states are exact bf16
        values, so [hs | hs] . [W_hi | W_lo]^T on the bf16 matrix cores is the fp32-weight product to ~2^-16 relative -- the
        fp32 MFMA path (a quarter of the rate, plus an fp32 copy of hs) is only used by the fp32 engine."""
        w = self.w
        if self.use_fused_ffn and hs.dtype in ops.H16:
            C = int(w["class.w"].shape[0])
            if C >= self.head_ts_min_classes and ops.head_ts_supported(C, "logits") and hs.shape[-1] == 256:
                # large charsets (Chinese: 7356 classes): tokens stationary in registers, the weight streamed once per 256 tokens --
                # the same two terms hs . W_hi + hs . W_lo (dtlr_head_ts; the tiled GEMM ran this head at 0.2 of the MFMA peak)
                img, bias = self._head_ts("class")
                return ops.head_ts(hs.contiguous(), img, bias, C, "logits")
            if "class.w2" not in w:
                wf = w["class.w"].float()
                hi = wf.to(hs.dtype)
                w["class.w2"] = torch.cat([hi, (wf - hi.float()).to(hs.dtype)], 1).contiguous()
            return ops.linear(torch.cat([hs, hs], -1), w["class.w2"], w["class.b"], out_dtype=torch.float32)
        return ops.linear(hs.float(), w["class.w"], w["class.b"])

    def _ffn(self, q, norm, x):
        """forward_ffn + post-norm (deformable_transformer.py:804-823, 876-880).  bf16 engine: one fused kernel, the
        d_ff-wide intermediate stays on chip; fp32 engine: two GEMMs + LayerNorm."""
        w = self.w
        if self.use_fused_ffn and self.use_ffn32 and x.dtype in ops.H16 and x.numel() // 256 >= 65536 and w[q + "ff1.w"].shape[0] % 32 == 0 \
                and 64 <= w[q + "ff1.w"].shape[0] <= 2048:
            if q + "ff.p32" not in w:                           # both weights in the 32x32 fragment order, packed once
                w[q + "ff.p32"] = ops.ffn32_pack(w[q + "ff1.w"], w[q + "ff2.w"])
            w1p, w2p = w[q + "ff.p32"]
            # whole rounds of 256 workgroups x 256 rows on the 32x32 kernel; a last partial round that fits 256 workgroups of the 16x16
            # kernel (192 or 128 rows each: a shorter round than a third full-length one) goes to that kernel
            M = x.numel() // 256
            rem = M % 65536
            if self.ffn32_tail and 0 < rem <= 49152 and ops.ffn_fused_supported(x, w[q + "ff1.w"]):
                if q + "ff2.wp" not in w:
                    w[q + "ff2.wp"] = ops.ffn_pack_w2(w[q + "ff2.w"])
                x2 = x.reshape(M, 256)
                y = torch.empty_like(x2)
                ops.ffn32(x2[:M - rem], w1p, w[q + "ff1.b"], w2p, w[q + "ff2.b"], w[q + norm + ".w"], w[q + norm + ".b"], out=y[:M - rem])
                ops.ffn_fused(x2[M - rem:], w[q + "ff1.w"], w[q + "ff1.b"], w[q + "ff2.wp"], w[q + "ff2.b"], w[q + norm + ".w"], w[q + norm + ".b"],
                              out=y[M - rem:])
                return y.view(x.shape)
            return ops.ffn32(x, w1p, w[q + "ff1.b"], w2p, w[q + "ff2.b"], w[q + norm + ".w"], w[q + norm + ".b"])
        if self.split and self.use_fused_ffn and x.shape[-1] == 256 and w[q + "ff1.b"].numel() % 32 == 0 and 32 <= w[q + "ff1.b"].numel() <= 2048:
            # split-fp32 engine: linear1 + ReLU + linear2 + residual + LayerNorm in one kernel on split operands (ffn_split.hip)
            if q + "ff.sp" not in w:
                w[q + "ff.sp"] = ops.ffn_split_pack(self._ffn_f32.pop(q + "ff1"), self._ffn_f32.pop(q + "ff2"))
            return ops.ffn_split(x, w[q + "ff.sp"], w[q + "ff1.b"], w[q + "ff2.b"], w[q + norm + ".w"], w[q + norm + ".b"])
        if self.use_fused_ffn and ops.ffn_fused_supported(x, w[q + "ff1.w"]):
            if q + "ff2.wp" not in w:                           # chunk-major copy of linear2.weight, packed once
                w[q + "ff2.wp"] = ops.ffn_pack_w2(w[q + "ff2.w"])
            return ops.ffn_fused(x, w[q + "ff1.w"], w[q + "ff1.b"], w[q + "ff2.wp"], w[q + "ff2.b"], w[q + norm + ".w"], w[q + norm + ".b"])
        h = self._lin(q + "ff1", x, relu=True)
        return self._ln(q + norm, self._lin(q + "ff2", h), residual=x)

    def backbone(self, x_nchw) -> List[torch.Tensor]:
        """torchvision resnet50 (v1.5) body with FrozenBN folded; returns layer2/3/4 maps, NHWC
        (models/dino/backbone.py:97-106,118-120)."""
        # stem: own kernels for both engines, reading the NCHW fp32 image directly (bf16: MFMA; fp32: exact direct convolution);
        # the folded-BN shift, the ReLU and the max-pool run as ONE pass over the full-resolution map
        # (16-bit engines: convolution + shift + ReLU + max-pool in one kernel)
        # (Round 5 tried the HBM-heavy front -- stem + layer1 + layer2, 268 MB maps at B = 32 -- per group of 4 / 8 / 16 images so that a
        # producer's map would still be in the 256 MiB Infinity Cache when its consumer reads it: 8.66 -> 9.74 / 9.08 / 8.86 ms per step;
        # the smaller launches lose more than the cache returns.  tools/experiments/gpu_calls/r05_call3.sh, profiles/r05_bb_group_c3.txt.)
        x = self._stem(x_nchw)
        return self._backbone_layers(x, 1, len(self.cfg.backbone_blocks))

    def _stem(self, x_nchw):
        if "conv1.frag" in self.w and self.use_stem_pool:
            return ops.stem_conv7x7_pool(x_nchw, self.w["conv1.frag"], self.w["conv1.b"], self.dtype)
        if "conv1.frag" in self.w:
            x = ops.stem_conv7x7(x_nchw, self.w["conv1.frag"], self.dtype)
        elif "conv1.fh" in self.w:
            x = ops.stem_conv7x7_f32s(x_nchw, self.w["conv1.fh"], self.w["conv1.fl"])
        else:
            x = ops.stem_conv7x7_f32(x_nchw, self.w["conv1.wk"])
        return ops.maxpool_nhwc(x, bias=self.w["conv1.b"], relu=True)

    def _backbone_layers(self, x, li_from, li_to):
        """bottleneck layers li_from..li_to (1-based, inclusive) on the NHWC map x; returns the maps of layers >= 2 among them"""
        outs = []
        pre = None                       # the NEXT bottleneck's conv1 output when the previous tail already computed it (layer1 chain)
        for li, nblocks in enumerate(self.cfg.backbone_blocks, start=1):
            if li < li_from or li > li_to:
                continue
            if li == 1 and self.use_l1_chain and x.dtype in ops.H16 and x.shape[-1] == 64 and x.numel() // 64 >= 16384 \
                    and self.w["l1.0.c3.w"].shape == (256, 64) and self.w["l1.0.ds.w"].shape == (256, 64):
                x, pre = self._layer1_chain(x, nblocks)
                continue
            for bi in range(nblocks):
                q = f"l{li}.{bi}."
                stride = 2 if (bi == 0 and li > 1) else 1
                cat = li == 2 and bi == 0 and self.use_l2_cat and x.dtype in ops.H16 and x.shape[-1] == 256 and self.w[q + "c3.w"].shape == (512, 128)
                join = None
                if cat:
                    idt = None
                elif bi != 0:
                    idt = x
                elif getattr(self, "overlap_streams", False) and x.is_cuda:
                    # the shortcut convolution of a layer's first bottleneck is independent of conv1 -> conv2: on a side stream under them
                    cur, join = self._side(2)
                    join.wait_stream(cur)
                    with torch.cuda.stream(join):
                        idt = self._conv(q + "ds", x, stride, 0)
                else:
                    idt = self._conv(q + "ds", x, stride, 0)
                o = pre if pre is not None else self._conv(q + "c1", x, 1, 0, relu=True)
                pre = None
                o = self._conv(q + "c2", o, stride, 1, relu=True)
                if join is not None:
                    cur.wait_stream(join)
                if cat:       # the strided shortcut convolution as K columns 128..383 of the tail GEMM: no shortcut map, no gather launch
                    if q + "cat.wk" not in self.w:
                        self.w[q + "cat.wk"] = ops.kres_pack(torch.cat([self.w[q + "c3.w"], self.w[q + "ds.w"]], 1).contiguous())
                        self.w[q + "cat.b"] = (self.w[q + "c3.b"].float() + self.w[q + "ds.b"].float()).contiguous()
                    x = ops.gemm_kres_cat_s2(o, x, self.w[q + "cat.wk"], self.w[q + "cat.b"], relu=True)
                else:
                    x = self._conv(q + "c3", o, 1, 0, relu=True, residual=idt)
            if li >= 2:
                outs.append(x)
        return outs

    def _layer1_chain(self, x0, nblocks):
        """layer1 (64-channel bottlenecks on the full-resolution pooled map: the HBM-heaviest part of the backbone) with the 1x1
        convolutions chained (ops.gemm_kres_chain): the first block's `downsample` shortcut is K columns 64..127 of its conv3 GEMM
        ([t | x] . [W3 | Wd]^T, bias b3 + bd: the 256-channel shortcut map is neither written nor read back), and every tail also
        produces the NEXT bottleneck's conv1 output from the tile it has on chip (the 256-channel map is written once and not re-read
        by conv1; the last tail feeds layer2.0.conv1).  torchvision Bottleneck.forward: out = relu(bn3(conv3(.)) + identity)."""
        w = self.w

        def wk(name):
            if name + ".wk" not in w:
                w[name + ".wk"] = ops.kres_pack(w[name + ".w"])
            return w[name + ".wk"]
        if "l1.0.cat.wk" not in w:
            w["l1.0.cat.wk"] = ops.kres_pack(torch.cat([w["l1.0.c3.w"], w["l1.0.ds.w"]], 1).contiguous())
            w["l1.0.cat.b"] = (w["l1.0.c3.b"].float() + w["l1.0.ds.b"].float()).contiguous()
        nxt = [f"l1.{bi + 1}.c1" for bi in range(nblocks - 1)] + (["l2.0.c1"] if self.use_l1_chain_out else [None])
        n2 = [64] * (nblocks - 1) + [128]
        o = self._conv("l1.0.c1", x0, 1, 0, relu=True)
        x = None
        for bi in range(nblocks):
            if o is None:                       # the previous tail could not produce this block's conv1
                o = self._conv(f"l1.{bi}.c1", x, 1, 0, relu=True)
            o = self._conv(f"l1.{bi}.c2", o, 1, 1, relu=True)
            nm = nxt[bi]
            if nm is not None and (w[nm + ".w"].shape != (n2[bi], 256) or (bi == 0 and n2[bi] != 64)):
                nm = None                       # a next conv1 the chain kernel has no form for (other widths; a one-block layer1): separate launch
            kw = dict(wp2=wk(nm), b2=w[nm + ".b"], n2=n2[bi]) if nm is not None else {}
            if bi == 0:
                x, o = ops.gemm_kres_chain(o, w["l1.0.cat.wk"], w["l1.0.cat.b"], x2=x0, relu=True, **kw)
            elif nm is not None:
                x, o = ops.gemm_kres_chain(o, wk(f"l1.{bi}.c3"), w[f"l1.{bi}.c3.b"], residual=x, relu=True, **kw)
            else:
                x, o = self._conv(f"l1.{bi}.c3", o, 1, 0, relu=True, residual=x), None
        return x, o

    def _geometry(self, mask, level_hw, has_padding=True):
        """Everything that depends only on the padding masks, ONE HIP launch per forward (ops.geometry): per-level masks
        (backbone.py:103, dino.py:304-307), pos + level embeds (position_encoding.py:79-108, deformable_transformer.py:281-285),
        valid ratios (:239-246), encoder reference points (:479-492), proposals and their validity (models/dino/utils.py:31-62)."""
        cfg, dev = self.cfg, mask.device
        level_hw = [(int(h), int(w)) for h, w in level_hw]
        g = ops.geometry(mask, level_hw, self.w["level_embed"], cfg.pe_temperatureH, cfg.pe_temperatureW, self.dtype)
        key = (dev, tuple(level_hw))
        if key not in self._level_cache:                              # int64 shapes / level starts for the B1 operator: once per shape
            shapes = torch.as_tensor(level_hw, dtype=torch.long, device=dev)
            lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
            self._level_cache[key] = (shapes, lsi, ops.msda_encoder_fits(level_hw, self.dtype))
        shapes, lsi, fits = self._level_cache[key]
        g.update(shapes=shapes, lsi=lsi, has_padding=has_padding, level_hw=level_hw, lds_msda_fits=fits)
        return g

    def _k256sw(self, name):
        """split engine: the resident-operand image of a [256, 256] projection for dtlr_gemm_k256s, packed once from the fp32 weight."""
        key = name + ".k256s"
        if key not in self.w:
            self.w[key] = ops.k256s_pack(self._ffn_f32.pop(name))
        return self.w[key]

    def _k256s_slices(self, name):
        """split engine: the weight [N, 256] of projection `name` (N a multiple of 128) as ceil(N / 256) resident-operand images of
        dtlr_gemm_k256s_multi, the last one zero-padded to 256 rows; packed once from the fp32 weight.  Returns [(image, first row, rows)]."""
        key = name + ".k256sm"
        if key not in self.w:
            wf = self._ffn_f32.pop(name)
            N = wf.shape[0]
            imgs = []
            for r0 in range(0, N, 256):
                n = min(256, N - r0)
                blk = wf[r0:r0 + n]
                if n < 256:
                    blk = torch.cat([blk, blk.new_zeros((256 - n, 256))], 0)
                imgs.append((ops.k256s_pack(blk.contiguous()), r0, n))
            self.w[key] = imgs
        return self.w[key]

    def _k256w(self, name):
        """fragment-order image of a [N, 256] projection weight for the weight-resident kernel, packed once."""
        key = name + ".k256"
        if key not in self.w:
            self.w[key] = ops.k256_pack(self.w[name + ".w"])
        return self.w[key]

    def _msda_mode(self, name, value_dtype, level_hw, ow, ref, n_heads):
        """'lds' or 'gather' for this encoder layer.  The LDS-window kernel fetches sampling points outside its staged columns through a
        global path that stalls a wave on 16 dependent loads per point: at ~1.2% of such points it is as slow as the gather kernel
        (tools/msda_sweep.py: 0.47 ms at 2.7% against 0.31 ms flat).  The fraction depends on the checkpoint's offset heads, so it is
        MEASURED -- but the two kernels are not bit-identical (packed-fp16 against fp32 accumulation), so the choice must not depend on
        the data or on the call history (round 2 probed the running batch every 256 calls: a result could depend on which batch had
        been probed, and data-parallel ranks could choose differently).  It is now a function of (weights, canvas shape) only: the
        first forward of a canvas shape runs ONE calibration pass of the encoder on a seeded noise batch of that shape
        (`_calibrate_msda`), probing every layer at window halos of 8 / 16 / 24 columns (dtlr_msda_encoder_far_samples: one small kernel
        + a 16-byte read-back each) and choosing the cheapest of {LDS kernel at one of those halos, gather kernel} under the cost model
        fitted to profiles/r03_msda_offset_halo_sweep_v1.json (a wider halo stages more columns -- 0.159 / 0.197 / 0.219 ms per call at
        no far samples -- but at sigma = 8 px offsets turns 0.365 ms into 0.234 ms, below the gather kernel's 0.307); every rank holds
        the same weights and generates the same noise, so every rank makes the same choice.  `msda_auto = False` pins the LDS
        kernel; `_msda_state[(layer, level_hw)] = {"mode": ...}` overrides a layer."""
        if not self.msda_auto:
            return "lds", None
        key = (name, tuple(level_hw))
        if self._msda_calibrating is not None:               # inside the calibration pass: measure, run the (always correct) LDS kernel
            best = {"mode": "gather", "halo": None, "cost": 1.0, "far": {}}
            for halo, base in self.msda_halo_base.items():
                if not ops.msda_encoder_fits(level_hw, value_dtype, halo):
                    continue
                far = ops.msda_encoder_far_fraction(value_dtype, level_hw, ow, ref, n_heads, halo)
                best["far"][halo] = far
                cost = base + self.msda_far_slope * math.sqrt(far)
                if cost < best["cost"]:
                    best.update(mode="lds", halo=halo, cost=cost)
            self._msda_calibrating[key] = best
            return "lds", None
        st = self._msda_state.get(key)
        return (st["mode"], st.get("halo")) if st is not None else ("lds", None)

    def _calibrate_msda(self, x_shape, level_hw):
        """One encoder pass on a seeded noise batch (2 lines of this canvas shape, unpadded) -> the per-layer kernel choice."""
        gen = torch.Generator().manual_seed(20260927)
        xc = torch.randn((2,) + tuple(x_shape[1:]), generator=gen).to(self.device)       # always 2 lines: the choice must not depend on the caller's batch size
        mc = torch.zeros((xc.shape[0],) + tuple(x_shape[2:]), dtype=torch.bool, device=self.device)
        self._msda_calibrating = {}
        try:
            feats, last, lhw = self.features(xc)
            g = self._geometry(mc, lhw, has_padding=True)          # not cached: the calibration batch has its own size
            self.encoder(self.tokens(feats, last, lhw), g)
            self._msda_state.update(self._msda_calibrating)
        finally:
            self._msda_calibrating = None
            # a shape the probe never reaches (window plan does not fit, L / P / head size other than 4 / 4 / 32) must not be
            # re-calibrated on every forward: record the (only possible) choice for it
            key0 = ("enc0.attn", tuple((int(h), int(w)) for h, w in level_hw))
            self._msda_state.setdefault(key0, {"mode": "gather", "halo": None})
            if len(self._msda_state) > 64 * max(1, self.cfg.enc_layers):      # evaluation over many canvas shapes: bounded state
                for k in list(self._msda_state)[: len(self._msda_state) // 2]:
                    if k != key0:
                        del self._msda_state[k]

    def _msda_module(self, name, query, query_pos, ref, value_src, g, n_points, value=None, ow_res=None):
        """MSDeformAttn.forward (ops/modules/ms_deform_attn.py:78-126) without the output
        projection (done by the caller, followed by the fused residual + LayerNorm).
        query + query_pos is formed in the GEMM prologue; the padding fill of `value` is its epilogue."""
        cfg = self.cfg
        B, Lq, C = query.shape
        S = value_src.shape[1]
        M, L, P = cfg.nheads, cfg.num_feature_levels, n_points
        k256 = self.use_k256 and query.dtype in ops.H16 and C == 256 and Lq == S
        k256s = self.split and self.use_k256s and C == 256 and Lq == S and (name + ".value") in self._k256s_ok
        ow = None
        if (k256s and self.use_k256s_multi_enc and value is None and ow_res is not None and not g["has_padding"] and value_src is query
                and query.is_contiguous() and ((name + ".ow") in self._ffn_f32 or (name + ".ow.k256sm") in self.w)
                and tuple(self.w[name + ".ow.w"].shape) == (384, 256) and ow_res.shape[-2] == S and S % 32 == 0):
            # split engine, unpadded batch: value_proj(src) and [offsets | logits](src + pos) = src W^T + (pos W^T + b) in ONE pass over src
            # (three slices: 256 | 256 | 128 channels; the position term is the row-broadcast residual of the last two)
            vimg = self._k256sw(name + ".value")
            oimgs = self._k256s_slices(name + ".ow")
            value = torch.empty((B, S, 256), dtype=torch.float32, device=query.device)
            ow = torch.empty((B, S, 384), dtype=torch.float32, device=query.device)
            sl = [dict(wp=vimg, out=value, bias=self.w[name + ".value.b"])]
            for img, r0, nr in oimgs:
                sl.append(dict(wp=img, out=ow[..., r0:r0 + nr], residual=ow_res[..., r0:r0 + nr]))
            ops.gemm_k256s_multi(query, sl, res_rows=S)
        if value is None:
            if k256s:
                value = ops.gemm_k256s(value_src, self._k256sw(name + ".value"), self.w[name + ".value.b"],
                                       row_mask=g["mask_flat"] if g["has_padding"] else None)
            elif k256:
                value = ops.gemm_k256(value_src, self._k256w(name + ".value"), 256, self.w[name + ".value.b"],
                                      row_mask=g["mask_flat"] if g["has_padding"] else None)
            else:
                value = self._lin(name + ".value", value_src, row_mask=g["mask_flat"] if g["has_padding"] else None)
        # the [offsets|logits] row stays in the activation dtype: in the bf16 engine its 2^-8 relative rounding
        # moves a sampling point by < 0.02 px, far below the bf16 noise of the sampled values themselves
        if ow is not None:
            pass
        elif k256 and ow_res is not None:
            # unpadded batch: (src + pos) W^T + b = src W^T + (pos W^T + b), and the second term is ONE [S, 384] matrix for every
            # image (L2-resident): the projection streams src alone and adds the row-broadcast term in its epilogue
            rows = ow_res.numel() // 384
            if self.use_kres and rows % 64 == 0 and (query.numel() // 256) % rows == 0 and self.w[name + ".ow.w"].shape[0] == 384:
                if name + ".ow.kb" not in self.w:               # residual tile DMA'd through LDS with the token tile (dtlr_gemm_kres_bcast384)
                    self.w[name + ".ow.kb"] = ops.kres_pack_bcast384(self.w[name + ".ow.w"])
                ow = ops.gemm_kres_bcast384(query, self.w[name + ".ow.kb"], ow_res)
            else:
                ow = ops.gemm_k256(query, self._k256w(name + ".ow"), 384, None, resid=ow_res)
        elif ow_res is not None:
            # fp32 / split engines, unpadded batch: the same identity through the tiled GEMM's row-broadcast residual epilogue (the A + A2
            # prologue variant keeps compiler-counted loads and half the occupancy: 212 us against the plain projection's ~140 at B = 32)
            ow = ops.linear_resbcast(query, self.w[name + ".ow.w"], ow_res)
        else:
            ow = self._lin(name + ".ow", query, a2=query_pos)
        if L == 4 and P == 4:
            if Lq == S and ref.shape[-1] == 2 and C // M == 32 and self.use_lds_msda and g["lds_msda_fits"]:   # encoder self-attention
                mode, halo = self._msda_mode(name, value.dtype, g["level_hw"], ow, ref, M)
                if mode == "lds":
                    return ops.msda_encoder(value.view(B, S, M, C // M), g["level_hw"], ow, ref, halo)
            return ops.msda_fused(value.unflatten(-1, (M, C // M)), g["shapes"], g["lsi"], ow, ref)
        ow = ow.float()
        off = ow[..., : M * L * P * 2].reshape(B, Lq, M, L, P, 2)
        aw = torch.softmax(ow[..., M * L * P * 2:].reshape(B, Lq, M, L * P), -1).reshape(B, Lq, M, L, P)
        if ref.shape[-1] == 2:
            normalizer = torch.stack([g["shapes"][:, 1], g["shapes"][:, 0]], -1).float()
            loc = ref[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
        else:
            loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
        return ops.msda(value.view(B, S, M, C // M), g["shapes"], g["lsi"], loc.contiguous(), aw.contiguous())

    def encoder(self, src, g):
        """TransformerEncoder.forward + DeformableTransformerEncoderLayer.forward
        (deformable_transformer.py:494-580, 804-823)."""
        pos = g["pos"]                                   # already in the engine dtype, level_embed added
        ow_res = [None] * self.cfg.enc_layers
        if not g["has_padding"]:
            pos = pos[0]                                 # unpadded batch: one [S, 256] matrix for every image (L2-resident A2 operand)
            if (self.use_k256 and src.dtype in ops.H16) or (self.use_ow_resbcast and src.dtype == torch.float32):
                if "enc_ow_res" not in g:                # pos W^T + b per layer, [S, 384] in the engine dtype: computed once per shape (g is cached)
                    g["enc_ow_res"] = [self._lin(f"enc{n}.attn.ow", pos) for n in range(self.cfg.enc_layers)]
                ow_res = g["enc_ow_res"]
        for n in range(self.cfg.enc_layers):
            q = f"enc{n}."
            a = self._msda_module(q + "attn", src, pos, g["enc_ref"], src, g, self.cfg.enc_n_points, ow_res=ow_res[n])
            src = self._proj_ln(q + "attn.out", q + "norm1", a, src)
            src = self._ffn(q, "norm2", src)
        return src

    def two_stage(self, memory, g, forced_topk=None):
        """deformable_transformer.py:320-363 with gen_encoder_output_proposals (utils.py:15-64).
        The box MLP runs only on the selected rows (selection uses class scores only)."""
        cfg = self.cfg
        w = self.w
        if self.use_fused_ffn and memory.dtype in ops.H16 and cfg.hidden_dim == 256:
            # bf16 engine: ONE kernel masks, projects and normalises, and writes output_memory as [hi | lo | hi] bf16; the class
            # head then runs on the bf16 matrix cores against [W_hi | W_hi | W_lo] (three-term split product, ~2^-16 relative:
            # selection scores as good as the fp32 MFMA path at a third of its time), and output_memory of the 900 selected rows
            # is rebuilt as hi + lo.
            if "enc_output.wp" not in w:
                w["enc_output.wp"] = ops.proj_pack_w(w["enc_output.w"])
                w["enc_class.w3"], w["enc_class.b3"] = ops.split_head_weight(w["enc_class.w"], w["enc_class.b"], dtype=memory.dtype)
            om = ops.proj_ln_split(memory, w["enc_output.wp"], w["enc_output.b"], g["keep"], w["enc_output_norm.w"], w["enc_output_norm.b"])
            # only max_c of the class head feeds the top-k: the GEMM's row-max epilogue (no [T, C] matrix, no reduction pass)
            # (Round 3 measured a two-pass form for the 7356-class head -- hi-only product over all tokens, exact three-term product over
            # the 1536 best candidates per line: 18.22 -> 17.56 ms per cfg5 step only, because the K = 256 pass is epilogue-bound (one
            # row-max epilogue per 4 k-slabs instead of 12), and the cut "no outsider can reach the exact top 900" could not be PROVEN
            # from the weights at that budget: the max over 7356 classes concentrates the token scores, ~600 tokens per line lie within
            # the 2 x 0.059 bound of the 900-th.  Dropped; the lever for that head is the large-N GEMM itself.)
            C_enc = int(w["enc_class.w"].shape[0])
            if (self.head_ts_scores or C_enc >= self.head_ts_min_classes) and ops.head_ts_supported(C_enc, "rowmax"):
                # round 5: for a large charset the tiled GEMM re-reads its token rows once per 128-channel tile (58 times for 7356 classes:
                # 4.9 of the Chinese step's 17 ms at 0.2 of the MFMA peak); the token-stationary kernel streams the weight instead
                img, bias = self._head_ts("enc_class")
                scores = ops.head_ts(om, img, bias, C_enc, "rowmax", 0, 256)
            else:
                scores = ops.linear_rowmax(om, w["enc_class.w3"], w["enc_class.b3"])
        else:
            if self.split and self.use_k256s and "enc_output" in self._k256s_ok and memory.shape[-1] == 256:
                # split engine: masking, projection and LayerNorm in one streaming pass (dtlr_gemm_k256s, LN form with a row mask)
                if "drop_rows" not in g:
                    g["drop_rows"] = (~g["keep"].bool()).contiguous()
                om = ops.gemm_k256s(memory, self._k256sw("enc_output"), w["enc_output.b"], row_mask=g["drop_rows"],
                                    ln_w=w["enc_output_norm.w"], ln_b=w["enc_output_norm.b"])
            else:
                om = memory * g["keep"].unsqueeze(-1).to(memory.dtype)
                # selection scores are computed in fp32: the projection writes fp32 straight from its accumulators
                om = self._ln("enc_output_norm", self._lin("enc_output", om, out_dtype=torch.float32))
            scores = ops.linear_rowmax(om, w["enc_class.w"], w["enc_class.b"])
        idx = ops.topk_rows(scores, cfg.num_queries) if forced_topk is None else forced_topk
        sel_raw, sel_x, prop_sel, init_box = ops.two_stage_gather(om, g["proposals"], idx)      # one launch for all the gathers
        ref_unsig = self._box_mlp("enc_bbox", sel_raw if sel_x is None else sel_x, prop_sel, mode=1)
        ts = dict(topk_idx=idx, topk_scores=scores, ref_unsig=ref_unsig, init_box=init_box)
        if sel_x is None:
            ts["hs_enc"] = sel_raw
        else:
            ts["hs_enc3"] = sel_raw
        return ts

    @staticmethod
    def _inverse_sigmoid(x, eps=1e-3):
        """util/misc.py:575-579."""
        x = x.clamp(min=0, max=1)
        return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))

    @staticmethod
    def _sine_embed(boxes):
        """gen_sineembed_for_position (models/dino/utils.py:141-167): [B,nq,4] -> [B,nq,512] (y|x|w|h)."""
        dim_t = torch.arange(128, dtype=torch.float32, device=boxes.device)
        dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / 128)
        e = (boxes[..., None] * (2 * math.pi)) / dim_t                 # [B,nq,4,128]
        e = torch.stack((e[..., 0::2].sin(), e[..., 1::2].cos()), dim=4).flatten(3)
        return torch.cat((e[:, :, 1], e[:, :, 0], e[:, :, 2], e[:, :, 3]), dim=2)

    def _box_mlp_hidden(self, name, x):
        """First two layers (Linear+ReLU, Linear+ReLU) of a 3-layer box MLP (models/dino/dino.py MLP) -> fp32 [.., 256]."""
        w = self.w
        if name + "0.wh" in w:                         # bf16 engine: bf16 MFMA, fp32 accumulate, fp32 result
            h = ops.linear(x.to(self.dtype), w[name + "0.wh"], w[name + "0.b"], relu=True)
            return ops.linear(h, w[name + "1.wh"], w[name + "1.b"], relu=True, out_dtype=torch.float32)
        h = ops.linear(x.float(), w[name + "0.w"], w[name + "0.b"], relu=True)
        return ops.linear(h, w[name + "1.w"], w[name + "1.b"], relu=True)

    def _box_mlp(self, name, x, ref, mode):
        """3-layer box MLP + consumer: mode 0 sigmoid(mlp(x) + inverse_sigmoid(ref)), mode 1 mlp(x) + ref."""
        w = self.w
        if name + "1.wp" in w:                         # bf16 engine: one launch
            return ops.box_mlp_refine(x.to(self.dtype), w[name + "0.wh"], w[name + "0.b"], w[name + "1.wp"], w[name + "1.b"],
                                      w[name + "2.w"], w[name + "2.b"], ref, mode)
        return ops.box_head_refine(self._box_mlp_hidden(name, x), w[name + "2.w"], w[name + "2.b"], ref, mode=mode)

    def _refine(self, x, ref):
        """sigmoid(bbox_embed(x) + inverse_sigmoid(ref)) (deformable_transformer.py:734-756; dino.py:339-354)."""
        return self._box_mlp("bbox", x, ref, mode=0)

    def _side(self, i: int):
        """(current stream, side stream i).  One set of side streams per engine: two forwards of one engine issued from two caller streams
        at once share them -- that only adds ordering between the two, never a race (every use is bracketed by stream waits)."""
        return torch.cuda.current_stream(self.device), self._side_streams[i]

    def _value_all_alloc(self, memory):
        Nall = self.w["dec.value_all.w"].shape[0]
        return torch.empty(memory.shape[:-1] + (Nall,), dtype=memory.dtype, device=memory.device)

    def _value_all_into(self, memory, g, vall) -> bool:
        """value_proj(memory) of ALL decoder layers into the caller's [B, S, layers x 256] buffer (same input, N = layers x 256: memory is
        read once instead of once per layer; layer n samples its column slice through the strided MSDA entry point).  False: no
        out-of-place form for this engine / shape (the caller runs ops.linear)."""
        C = self.cfg.hidden_dim
        rmask = g["mask_flat"] if g["has_padding"] else None
        Nall = self.w["dec.value_all.w"].shape[0]
        if self.use_k256 and memory.dtype in ops.H16 and C == 256 and Nall % 384 == 0:
            # weight-resident streaming kernel, 384 output channels per launch, written as column slices of one [B, S, N] buffer
            for j in range(Nall // 384):
                key = f"dec.value_all.k256.{j}"
                if key not in self.w:
                    self.w[key] = ops.k256_pack(self.w["dec.value_all.w"][384 * j:384 * (j + 1)])
                ops.gemm_k256(memory, self.w[key], 384, self.w["dec.value_all.b"][384 * j:384 * (j + 1)], row_mask=rmask,
                              out=vall[..., 384 * j:384 * (j + 1)])
            return True
        if (self.split and self.use_k256s_multi and C == 256 and Nall % 256 == 0 and Nall // 256 <= 8 and memory.is_contiguous()
                and ("dec.value_all" in self._ffn_f32 or "dec.value_all.k256sm" in self.w)):
            # split engine: the six value projections as six slices of ONE pass over memory (dtlr_gemm_k256s_multi)
            bias = self.w["dec.value_all.b"]
            ops.gemm_k256s_multi(memory, [dict(wp=img, out=vall[..., r0:r0 + nr], bias=bias[r0:r0 + nr]) for img, r0, nr in self._k256s_slices("dec.value_all")],
                                 row_mask=rmask)
            return True
        return False

    def decoder(self, memory, ts, g, want_aux=False, dbg=None, vall=None):
        """TransformerDecoder.forward + DeformableTransformerDecoderLayer
        (deformable_transformer.py:652-766, 882-997), batch-first.  vall: value_proj(memory) of all layers when the caller already
        computed it (forward() does, on a side stream under the two-stage selection)."""
        cfg = self.cfg
        B = memory.shape[0]
        ref = ts["ref_unsig"].sigmoid()
        # (Measured and dropped in round 3: walking the queries in cx order of their reference points instead of the reference's score
        # order -- every decoder operator is per-query or permutation-equivariant -- makes the cross-attention gather 6% faster (74.9 ->
        # 70.5 us per call: neighbouring quads then share cache lines), but the permutation's own gathers cost more than the 26 us it
        # saves per step.)
        tgt = self.w["tgt_embed"][None].expand(B, -1, -1).contiguous()
        refs = [ref]
        hs = []
        C = cfg.hidden_dim
        if vall is None:
            vall = self._value_all_alloc(memory)
            if not self._value_all_into(memory, g, vall):
                vall = ops.linear(memory, self.w["dec.value_all.w"], self.w["dec.value_all.b"], row_mask=g["mask_flat"] if g["has_padding"] else None)
        for n in range(cfg.dec_layers):
            q = f"dec{n}."
            w = self.w
            if self.use_dec_query_stage and tgt.dtype in ops.H16 and C == 256 and w["dec.rph0.w"].shape == (256, 512):
                for nm in ("dec.rph0", "dec.rph1", q + "sa.qk", q + "sa.v"):
                    if nm + ".dq" not in w:                    # fragment-order images, packed once
                        w[nm + ".dq"] = ops.dq_pack(w[nm + ".w"])
                ref_in, qpos, qk, v = ops.dec_query_stage(ref, g["valid_ratios"], tgt, w["dec.rph0.dq"], w["dec.rph0.b"], w["dec.rph1.dq"],
                                                          w["dec.rph1.b"], w[q + "sa.qk.dq"], w[q + "sa.qk.b"], w[q + "sa.v.dq"], w[q + "sa.v.b"])
            else:
                ref_in, sine = ops.decoder_query_prep(ref, g["valid_ratios"], self.dtype)      # [B,nq,L,4], [B,nq,512]
                qpos = self._lin("dec.rph1", self._lin("dec.rph0", sine, relu=True))
                # self attention (q = k = tgt + query_pos, v = tgt)
                qk = self._lin(q + "sa.qk", tgt, a2=qpos)
                v = self._lin(q + "sa.v", tgt)
            a = ops.mha(qk, v, cfg.nheads, split=self.split)
            tgt = self._proj_ln(q + "sa.out", q + "norm2", a, tgt)
            # deformable cross attention
            a = self._msda_module(q + "attn", tgt, qpos, ref_in, memory, g, cfg.dec_n_points, value=vall[..., n * C:(n + 1) * C])
            tgt = self._proj_ln(q + "attn.out", q + "norm1", a, tgt)
            # ffn
            tgt = self._ffn(q, "norm3", tgt)
            # iterative box refinement (734-756)
            ref = self._refine(tgt, ref)
            refs.append(ref)
            if dbg is not None:                                        # tools/error_budget.py: per-layer state
                dbg.setdefault("tgt", []).append(tgt)
            if want_aux or n == cfg.dec_layers - 1:
                hs.append(self._ln("dec.norm", tgt))
            else:
                hs.append(None)
        return hs, refs

    def features(self, x):
        """backbone maps (NHWC) + the extra stride-2 level's convolution (dino.py:290-311) and the level sizes."""
        feats = self.backbone_swin(x.float()) if self.cfg.is_swin else self.backbone(x.float())
        level_hw = [(f.shape[1], f.shape[2]) for f in feats]
        last = self._conv(f"ip{len(feats)}", feats[-1], 2, 1)
        level_hw.append((last.shape[1], last.shape[2]))
        return feats, last, level_hw

    def _features_tokens_overlapped(self, x):
        """features() + tokens() of the ResNet path with input_proj + GroupNorm of level 0 / 1 on side streams under layer3 / layer4
        (dino.py:290-311 consumes C3 / C4 / C5 only after the whole backbone; nothing orders level 0's projection after layer3).  The token
        matrix is allocated before the forks, on the caller's stream; the side streams write disjoint row ranges of it."""
        x1 = self._stem(x.float())
        c3 = self._backbone_layers(x1, 1, 2)[-1]
        B, h, w_ = c3.shape[0], c3.shape[1], c3.shape[2]
        nlev = self.cfg.num_feature_levels
        level_hw = [(h, w_)]
        for _ in range(nlev - 1):                                   # 3x3 / stride 2 / pad 1 stages: layer3, layer4, input_proj[3]
            h, w_ = (h - 1) // 2 + 1, (w_ - 1) // 2 + 1
            level_hw.append((h, w_))
        if nlev != 4 or len(self.cfg.backbone_blocks) != 4:
            raise RuntimeError("overlap_streams: the overlapped schedule is written for the 4-level ResNet configuration")
        T = [a * b for a, b in level_hw]
        src = torch.empty((B, sum(T), self.cfg.hidden_dim), dtype=self.dtype, device=c3.device)

        def proj_norm(l, f, off):
            t = self._lin(f"ip{l}", f.flatten(1, 2)) if l < nlev - 1 else f.flatten(1, 2)
            ops.groupnorm_tokens(t, 32, self.w[f"ip{l}.gn.w"], self.w[f"ip{l}.gn.b"], out=src[:, off:off + T[l]])

        cur, s0 = self._side(0)
        s0.wait_stream(cur)
        with torch.cuda.stream(s0):
            proj_norm(0, c3, 0)
        c4 = self._backbone_layers(c3, 3, 3)[-1]
        _, s1 = self._side(1)
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            proj_norm(1, c4, T[0])
        c5 = self._backbone_layers(c4, 4, 4)[-1]
        last = self._conv(f"ip{nlev - 1}", c5, 2, 1)
        assert [(f.shape[1], f.shape[2]) for f in (c3, c4, c5, last)] == level_hw
        proj_norm(2, c5, T[0] + T[1])
        proj_norm(3, last, T[0] + T[1] + T[2])
        cur.wait_stream(s0)
        cur.wait_stream(s1)
        return [c3, c4, c5], last, level_hw, src

    def geometry_for(self, x, mask, level_hw, has_padding=True):
        # geometry depends only on the canvas shape and the padding masks: for an unpadded batch it is
        # the same for every forward of that shape -> cached (legal savings)
        gkey = (tuple(x.shape), tuple(level_hw)) if not has_padding else None
        g = self._shape_cache.get(gkey) if gkey is not None else None
        if g is None:
            g = self._geometry(mask, level_hw, has_padding)
            if gkey is not None:
                if len(self._shape_cache) > 8:
                    self._shape_cache.clear()
                self._shape_cache[gkey] = g
        return g

    def tokens(self, feats, last, level_hw):
        """input_proj + GroupNorm of every level (dino.py:115-136), each level normalised straight into its rows of the
        concatenated token matrix (no torch.cat pass)."""
        B = last.shape[0]
        S_tot = sum(h * w for h, w in level_hw)
        src = torch.empty((B, S_tot, self.cfg.hidden_dim), dtype=self.dtype, device=last.device)
        off = 0
        for l, f in enumerate(list(feats) + [last]):
            t = self._lin(f"ip{l}", f.flatten(1, 2)) if l < len(feats) else f.flatten(1, 2)
            T_l = t.shape[1]
            ops.groupnorm_tokens(t, 32, self.w[f"ip{l}.gn.w"], self.w[f"ip{l}.gn.b"], out=src[:, off:off + T_l])
            off += T_l
        return src

    def heads(self, hs, refs, ts, want_aux=False):
        """DINO.forward's tail (models/dino/dino.py:339-415): class / box heads of the last (and, on request, every) decoder layer
        and the two-stage intermediate outputs."""
        cfg = self.cfg
        n = cfg.dec_layers - 1
        out = {
            "pred_logits": self._class_head(hs[n]),
            "pred_boxes": self._refine(hs[n], refs[n]),
        }
        if want_aux:
            out["aux_outputs"] = [{"pred_logits": self._class_head(hs[i]), "pred_boxes": self._refine(hs[i], refs[i])}
                                  for i in range(n)]
        if "hs_enc3" in ts:                                        # bf16 engine: the two-stage head on its split images
            C = int(self.w["enc_class.w"].shape[0])     # the two-stage head's OWN class count (--fix_enc_out_class keeps the old one)
            if C >= self.head_ts_min_classes and ops.head_ts_supported(C, "logits"):
                img, bias = self._head_ts("enc_class")
                interm_class = ops.head_ts(ts["hs_enc3"].contiguous(), img, bias, C, "logits", 0, 256)
            else:
                interm_class = ops.linear(ts["hs_enc3"], self.w["enc_class.w3"][:C], self.w["enc_class.b3"][:C], out_dtype=torch.float32)
        else:
            interm_class = ops.linear(ts["hs_enc"], self.w["enc_class.w"], self.w["enc_class.b"])
        out["interm_outputs"] = {"pred_logits": interm_class, "pred_boxes": ts["ref_unsig"].sigmoid()}
        out["interm_outputs_for_matching_pre"] = {"pred_logits": interm_class, "pred_boxes": ts["init_box"]}
        out["dn_meta"] = None
        return out

    # ------------------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x: torch.Tensor, mask: torch.Tensor, forced_topk: Optional[torch.Tensor] = None,
                want_aux: bool = False, return_debug: bool = False, has_padding: bool = True) -> Dict[str, torch.Tensor]:
        """x [B,3,H,W] fp32 (zero-padded), mask [B,H,W] bool (True = padding)  ->  DINO.forward's
        dict (models/dino/dino.py:270-415): pred_logits [B,nq,C] raw, pred_boxes [B,nq,4] cxcywh."""
        ops.require_cuda(x, "images")
        cfg = self.cfg
        B = x.shape[0]
        st = torch.cuda.current_stream(self.device).cuda_stream
        if st not in self._ws_streams and not torch.cuda.is_current_stream_capturing():
            # the library's split-K / hidden-split scratch of THIS stream, sized once before anything can be captured on it: a captured
            # forward then allocates nothing and dispatches exactly as an eager one (dtlr_hip.h, dtlr_workspace_reserve)
            ops.workspace_reserve(self.dtype)
            self._ws_streams.add(st)
        overlap = self.overlap_streams and not cfg.is_swin
        src = None
        if overlap:
            feats, last, level_hw, src = self._features_tokens_overlapped(x)
        else:
            feats, last, level_hw = self.features(x)
        if self._range_check_pending:
            # fp16 storage / split fp16 operands saturate at 65504 (the conversions do not clamp: a larger backbone activation becomes inf and
            # poisons a whole GroupNorm group).  No trained checkpoint ships with the reference, so the range assumption is CHECKED on the
            # first forward of an engine instead of trusted: one host read, once.
            self._range_check_pending = False
            peak = max(float(f.float().abs().max()) for f in list(feats) + [last])
            if not (peak < 6.0e4):
                raise RuntimeError(f"DTLREngine({'f32s' if self.split else 'float16'}): backbone activations reach {peak:.3g}, beyond fp16's range "
                                   "(65504) -- run this checkpoint on the bfloat16 or the exact float32 engine")
        g = self.geometry_for(x, mask, level_hw, has_padding)
        if src is None:
            src = self.tokens(feats, last, level_hw)
        if self.msda_auto and self.use_lds_msda and ("enc0.attn", tuple((int(h), int(w)) for h, w in level_hw)) not in self._msda_state:
            self._calibrate_msda(x.shape, level_hw)            # once per canvas shape
        memory = self.encoder(src, g)
        vall = None
        if overlap and memory.is_contiguous():
            # value_proj(memory) of the six decoder layers on a side stream, under the two-stage selection (buffer allocated HERE, on this stream)
            vall = self._value_all_alloc(memory)
            cur, side = self._side(0)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                done = self._value_all_into(memory, g, vall)
            if not done:
                vall = None
        ts = self.two_stage(memory, g, forced_topk)
        if vall is not None:
            cur.wait_stream(side)
        hs, refs = self.decoder(memory, ts, g, want_aux, vall=vall)
        out = self.heads(hs, refs, ts, want_aux)
        if return_debug:
            out["_debug"] = dict(memory=memory, topk_idx=ts["topk_idx"], topk_scores=ts["topk_scores"], src=src,
                                 feats=feats, geometry=g)
        return out
