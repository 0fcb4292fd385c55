"""Deterministic, name-seeded synthetic weights + checkpoint ingestion.

No checkpoint ships with the reference tree (README.md:63-71 are Drive links), so parity and
benchmarks run on weights produced here: every tensor of the reference state-dict schema
(keys as produced by models/dino/dino.py:49-245 + backbone.py:36-128) is
drawn from a numpy PCG64 stream seeded by crc32(canonical tensor name) ^ seed.  The same call
yields identical weights in the authoring container (where they are loaded into the imported
reference to make the golden fixtures) and on the GPU box.

A real checkpoint (`torch.load(path)["model"]`, evaluation.py:55-56) goes through
`load_checkpoint_state_dict` and is used unchanged.
"""
from __future__ import annotations

import math
import re
import zlib
from collections import OrderedDict
from typing import Dict

import numpy as np
import torch

from .config import DTLRConfig

# Version 2 (round 2): heads with TRAINED-LIKE MARGINS.  Version 1 drew every head tensor i.i.d.: all 900 queries decoded to a
# non-blank class on top-1/top-2 margins of ~0, so any rounding flipped labels and decoded-string comparisons said nothing.
# Version 2 keeps every backbone / encoder / attention tensor of version 1 and changes only
#   * the decoder's residual branches (self-attn out_proj, cross-attn output_proj, linear2: x 0.3) so that a query's content
#     embedding survives the 18 post-norm sublayers next to the image-dependent signal,
#   * tgt_embed: a weak i.i.d. part + beta_q * sqrt(d) * (the code vector of the query's designated class, or the blank code),
#   * class_embed: row c = g * code_c - g_b * code_blank + small noise, bias -4.6 (the reference's init, dino.py:164-166),
# so that ~8% of the queries carry a character (designated-class logit ~ +2..+9, image-dependent spread ~ +-1) and the rest
# are blank (every class logit << 0), like the output of a trained recogniser.  beta_q varies per query, so margins range from
# a few tenths of a logit to many logits.
GENERATOR_VERSION = 2
# Version 3 (round 3, `synthetic_state_dict(..., version=3)`: a STRESS set, not the parity reference).  The round-2 review objected that
# the x 0.3 damping shrinks exactly the error path the parity gates measure.  v3 removes it and makes explicit what it stood in for: a
# trained decoder keeps a query's identity across its 18 post-norm sublayers because its weights keep re-writing it, while an i.i.d.
# decoder at natural branch gain (each sublayer adds a branch of the stream's own norm, the LayerNorm rescales) washes any planted
# content out as 2^(-18/2).  v3 gives every decoder FFN a small PROTOTYPE MEMORY (the key-value structure trained FFNs are known to
# hold): for each code a query is designated to carry (its class code, the blank code), V3_MEM_COPIES hidden units whose linear1 row
# is that code (bias = -threshold: the unit fires only for a stream pointing along the code) and whose linear2 column writes the code
# back; every other hidden unit, attention projection and output projection is i.i.d. at NATURAL gain.  Measured (MI355X, round 3,
# profiles/r03_error_budget_*_v3w.json): the content survives (99.4% of the queries decode to their designation, margins of the
# character queries: median 3.3 logits), mean 16-bit errors are BELOW v2's (the memory re-writes the content cleanly), but the threshold
# units make the network a strong error amplifier for the few queries near a unit's firing threshold: the exact-fp32 HIP engine and the
# reference's own CPU forward -- two correct fp32 evaluations -- already differ by 1.0e-3 in the logits (3e-5 on v2).  A network
# that amplifies fp32 summation-order noise to the north-star tolerance cannot be the yardstick for "logits within 1e-3", so v2 stays
# the generator of the goldens; v3 is exercised by tools/error_budget.py --weights 3 and a GPU stress test.


# Version 4 (round 5, `synthetic_state_dict(..., version=4)`: the FREE-RUNNING parity set; v2 stays the generator of the goldens).  In v2 a
# character is planted in tgt_embed, i.e. it belongs to a selection RANK (deformable_transformer.py:354-355: content query i = row i): two
# tokens whose two-stage scores are closer than an implementation's score error trade ranks, and with them their characters trade positions
# on the line.  Any two fp32 implementations differ that way (the oracle against ITSELF at 1e-5 score noise: 7 of 16 strings survive), and a
# 16-bit engine's free-running CER against the oracle is ~90% on v2 -- a property of the planting, not of the arithmetic.  A trained
# recogniser reads the character from the IMAGE at the query's position, so its decode is invariant to rank swaps.  v4 restates that:
#   * tgt_embed: every row the same vector (beta * sqrt(d) * blank code): a rank swap is a pure permutation of identical content queries;
#   * characters come from the image: the LAST decoder layer's FFN holds V4_DETECTORS one-shot detector units -- unit k fires on the stream
#     entering the FFN (post-norm1: cross-attention of the image memory at the query's reference box) beyond the (1 - V4_TAIL) quantile of
#     a fixed random direction u_k, and writes V4_GAIN * exceedance (in standard deviations) * code(class_k) into the stream; everything else
#     (backbone, encoder, attention, damping, heads) is v2's.  No feedback (one layer, one shot): the v3 bistability does not arise.
#   * the quantile and the spread of <u_k, x> are not computable from the weights alone: they are CALIBRATED once on a seeded batch through the
#     CPU oracle (tools/calibrate_generator_v4.py) and frozen below -- a constant of the generator like V3_MEM_THETA, Latin config, seed 0.
# Measured with the oracle (4 bench lines): ~60 characters per line over 16 classes, character margins median 4 logits, 3 of 3600 queries
# within 0.1 of their decision boundary; the oracle against itself at 1e-5 / 5e-5 score noise (62 / 265 rank slots changed): all strings
# identical; at 4e-2 (a bf16 engine's score error: ~50 of the 900 selected tokens differ per line) ~8% CER from the changed SET.
V4_DETECTORS = 16
V4_TAIL = 0.015
V4_GAIN = 30.0
V4_BETA = 0.75
V4_CALIBRATION: Dict[tuple, Dict[str, list]] = {}          # (num_classes, backbone, seed) -> {"theta": [...], "sigma": [...]}; filled below
_V4_ALLOW_UNCALIBRATED = False                               # set by tools/calibrate_generator_v4.py only: v4 without its detector bank
# tools/calibrate_generator_v4.py: 7200 queries (8 noise lines 128x2048, seed 4242), V4_TAIL = 0.015; detector classes
# [118, 104, 19, 78, 0, 54, 61, 159, 106, 73, 79, 135, 82, 1, 128, 21]
V4_CALIBRATION[(166, "resnet50", 0)] = {
    "theta": [-0.604932, 0.664480, 1.016210, 1.452146, 1.277894, 0.748140, -0.607717, -0.081358, 1.195282, -0.562536, 0.078129, 1.041693, -0.568296, -1.365318, 2.034528, -0.728936],
    "sigma": [0.125409, 0.139103, 0.220382, 0.146720, 0.212711, 0.087712, 0.092358, 0.141598, 0.151736, 0.095402, 0.074732, 0.097516, 0.108304, 0.228227, 0.072932, 0.063551],
}


def v4_detectors(cfg: DTLRConfig, seed: int):
    """(directions [K, d] unit rows orthogonal to the blank code, class id per detector): name-seeded, independent of the calibration"""
    d, C = cfg.hidden_dim, cfg.num_classes
    code, _, _ = _codes(cfg, seed, 2)
    r = _rng("v4.detectors", seed)
    U = r.standard_normal((V4_DETECTORS, d)).astype(np.float32)
    cb = code[C]
    U = U - (U @ cb)[:, None] * cb[None]
    U /= np.linalg.norm(U, axis=1, keepdims=True)
    classes = r.choice(C, V4_DETECTORS, replace=False)
    return U.astype(np.float32), classes


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF))


def _normal(name, seed, shape, std):
    return torch.from_numpy((_rng(name, seed).standard_normal(shape) * std).astype(np.float32))


def _uniform(name, seed, shape, lo, hi):
    return torch.from_numpy(_rng(name, seed).uniform(lo, hi, shape).astype(np.float32))


def _linear(sd, name, out_f, in_f, seed, gain=1.0, bias_std=0.02, bias_const=None):
    sd[name + ".weight"] = _normal(name + ".weight", seed, (out_f, in_f), gain / math.sqrt(in_f))
    if bias_const is not None:
        sd[name + ".bias"] = torch.full((out_f,), float(bias_const))
    else:
        sd[name + ".bias"] = _normal(name + ".bias", seed, (out_f,), bias_std)


def _norm(sd, name, n, seed):
    sd[name + ".weight"] = _uniform(name + ".weight", seed, (n,), 0.8, 1.2)
    sd[name + ".bias"] = _normal(name + ".bias", seed, (n,), 0.05)


def _frozen_bn(sd, name, n, seed, scale=1.0):
    # FrozenBatchNorm2d buffers (models/dino/backbone.py:45-49)
    sd[name + ".weight"] = _uniform(name + ".weight", seed, (n,), 0.8 * scale, 1.2 * scale)
    sd[name + ".bias"] = _normal(name + ".bias", seed, (n,), 0.05)
    sd[name + ".running_mean"] = _normal(name + ".running_mean", seed, (n,), 0.05)
    sd[name + ".running_var"] = _uniform(name + ".running_var", seed, (n,), 0.6, 1.4)


def _conv(sd, name, cout, cin, k, seed, gain=math.sqrt(2.0)):
    sd[name] = _normal(name, seed, (cout, cin, k, k), gain / math.sqrt(cin * k * k))


def _msda(sd, p, cfg: DTLRConfig, n_points: int, seed: int):
    d, M, L, P = cfg.hidden_dim, cfg.nheads, cfg.num_feature_levels, n_points
    # offsets: small data-dependent part + the ring-shaped bias pattern the reference initialises
    # (ops/modules/ms_deform_attn.py:62-70): head h points along angle 2*pi*h/M, point p at radius p+1
    sd[p + ".sampling_offsets.weight"] = _normal(p + ".sampling_offsets.weight", seed, (M * L * P * 2, d), 0.5 / math.sqrt(d))
    ang = np.arange(M, dtype=np.float64) * (2.0 * math.pi / M)
    dirs = np.stack([np.cos(ang), np.sin(ang)], -1)
    dirs = dirs / np.abs(dirs).max(-1, keepdims=True)
    grid = np.tile(dirs[:, None, None, :], (1, L, P, 1)) * np.arange(1, P + 1, dtype=np.float64)[None, None, :, None]
    sd[p + ".sampling_offsets.bias"] = torch.from_numpy(grid.reshape(-1).astype(np.float32))
    _linear(sd, p + ".attention_weights", M * L * P, d, seed, gain=1.0)
    _linear(sd, p + ".value_proj", d, d, seed)
    _linear(sd, p + ".output_proj", d, d, seed)


V3_MEM_COPIES = 6          # hidden units per stored code
V3_MEM_THETA = (3.5, 4.0, 4.5, 5.0, 5.5, 6.0)      # firing thresholds of the copies (projection of a unit-variance stream on a unit code ~ N(0,1))
V3_MEM_GAIN = 12.0         # write-back strength: out = gain * sum_r relu(<x, code> - theta_r) / copies * code


def _prototype_memory(sd, cfg: DTLRConfig, code, q_cls, d: int, ff: int):
    """Generator v3: overwrite the first (#stored codes x V3_MEM_COPIES) hidden units of every decoder FFN with code detectors /
    emitters (see GENERATOR_VERSION).  Stored codes = the blank code and the class code of every designated character query."""
    C = cfg.num_classes
    stored = [C] + sorted(set(int(c) for c in q_cls if c >= 0))
    n_units = len(stored) * V3_MEM_COPIES
    if n_units > ff:
        raise ValueError(f"generator v3: {len(stored)} stored codes x {V3_MEM_COPIES} copies do not fit dim_feedforward = {ff}")
    K = torch.from_numpy(code[stored].astype(np.float32))                       # [n_codes, d], unit rows
    rows = K.repeat_interleave(V3_MEM_COPIES, 0)                                # detector rows (a post-norm stream has ~unit variance per channel)
    theta = torch.tensor(V3_MEM_THETA[:V3_MEM_COPIES], dtype=torch.float32).repeat(len(stored))
    for n in range(cfg.dec_layers):
        p = f"transformer.decoder.layers.{n}."
        sd[p + "linear1.weight"][:n_units] = rows
        sd[p + "linear1.bias"][:n_units] = -theta
        sd[p + "linear2.weight"][:, :n_units] = (V3_MEM_GAIN / V3_MEM_COPIES) * rows.t()


def _codes(cfg: DTLRConfig, seed: int, version: int = GENERATOR_VERSION):
    """Generator v2/v3: unit-norm sign codes, one per class plus the blank code, and the per-query designation
    (class id or -1 = blank, strength beta; v3 starts every query at beta >= 0.8: near the edge of the memory's capture range a
    query's content decays layer by layer and an arbitrarily small perturbation decides whether it survives -- a bistability a
    trained query does not have, and one that turns rounding errors of 1e-4 into logit differences of several units)."""
    d, C, nq = cfg.hidden_dim, cfg.num_classes, cfg.num_queries
    r = _rng("v2.codes", seed)
    code = (r.integers(0, 2, (C + 1, d)).astype(np.float32) * 2 - 1) / math.sqrt(d)      # rows 0..C-1: classes, row C: blank
    r = _rng("v2.designation", seed)
    is_char = r.random(nq) < 0.08
    cls = np.where(is_char, r.integers(0, C, nq), -1)
    beta = (np.where(is_char, r.uniform(0.15, 1.0, nq), r.uniform(0.5, 1.0, nq)) if version == 2 else
            np.where(is_char, r.uniform(0.85, 1.0, nq), r.uniform(0.8, 1.0, nq))).astype(np.float32)
    return code, cls, beta


def synthetic_state_dict(cfg: DTLRConfig, seed: int = 0, version: int = GENERATOR_VERSION) -> "OrderedDict[str, torch.Tensor]":
    """fp32 CPU tensors keyed exactly like the reference's `model.state_dict()` (generator version GENERATOR_VERSION; version=2
    reproduces round 2's damped-branch weights for A/B studies)."""
    cfg.validate()
    if version not in (2, 3, 4):
        raise ValueError(f"synthetic_state_dict: unknown generator version {version}")
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    d, C, ff = cfg.hidden_dim, cfg.num_classes, cfg.dim_feedforward

    if cfg.is_swin:
        _swin(sd, cfg, seed)
    # ---- backbone.0.body.* : ResNet-50 v1.5 with FrozenBN -----------------------------------
    b = "backbone.0.body."
    if not cfg.is_swin:
        _resnet(sd, cfg, seed)
    _rest(sd, cfg, seed, d, C, ff, version)
    return sd


def _swin(sd, cfg: DTLRConfig, seed: int):
    """backbone.0.* of a Swin backbone (models/dino/swin_transformer.py:435-555: patch_embed, layers.{i}.blocks.{j}, downsample,
    norm{i} for the returned stages; `relative_position_index` is a registered buffer and part of the state dict)."""
    sp = cfg.swin_params()
    E, depths, heads, ws = sp["embed_dim"], sp["depths"], sp["num_heads"], sp["window_size"]
    b = "backbone.0."
    _conv(sd, b + "patch_embed.proj.weight", E, 3, 4, seed, gain=1.0)
    sd[b + "patch_embed.proj.bias"] = _normal(b + "patch_embed.proj.bias", seed, (E,), 0.02)
    _norm(sd, b + "patch_embed.norm", E, seed)
    coords = np.stack(np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij")).reshape(2, -1)
    rel = (coords[:, :, None] - coords[:, None, :]).transpose(1, 2, 0).copy()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    rel_index = torch.from_numpy(rel.sum(-1).astype(np.int64))
    for i in range(4):
        C = E << i
        for j in range(depths[i]):
            p = f"{b}layers.{i}.blocks.{j}."
            _norm(sd, p + "norm1", C, seed)
            sd[p + "attn.relative_position_bias_table"] = _normal(p + "attn.relative_position_bias_table", seed, ((2 * ws - 1) ** 2, heads[i]), 0.5)
            sd[p + "attn.relative_position_index"] = rel_index.clone()
            _linear(sd, p + "attn.qkv", 3 * C, C, seed, gain=1.4)
            _linear(sd, p + "attn.proj", C, C, seed, gain=0.5)
            _norm(sd, p + "norm2", C, seed)
            _linear(sd, p + "mlp.fc1", 4 * C, C, seed, gain=math.sqrt(2.0))
            _linear(sd, p + "mlp.fc2", C, 4 * C, seed, gain=0.5)
        if i < 3:
            p = f"{b}layers.{i}.downsample."
            _norm(sd, p + "norm", 4 * C, seed)
            sd[p + "reduction.weight"] = _normal(p + "reduction.weight", seed, (2 * C, 4 * C), 1.0 / math.sqrt(4 * C))
        if i in cfg.return_interm_indices:
            _norm(sd, f"{b}norm{i}", C, seed)


def _resnet(sd, cfg: DTLRConfig, seed: int):
    b = "backbone.0.body."
    _conv(sd, b + "conv1.weight", 64, 3, 7, seed)
    _frozen_bn(sd, b + "bn1", 64, seed)
    inplanes = 64
    for li, (planes, nblocks) in enumerate(zip((64, 128, 256, 512), cfg.backbone_blocks), start=1):
        for bi in range(nblocks):
            p = f"{b}layer{li}.{bi}."
            _conv(sd, p + "conv1.weight", planes, inplanes, 1, seed)
            _frozen_bn(sd, p + "bn1", planes, seed)
            _conv(sd, p + "conv2.weight", planes, planes, 3, seed)
            _frozen_bn(sd, p + "bn2", planes, seed)
            _conv(sd, p + "conv3.weight", planes * 4, planes, 1, seed)
            _frozen_bn(sd, p + "bn3", planes * 4, seed, scale=0.4)   # damp the residual branch
            if bi == 0:
                _conv(sd, p + "downsample.0.weight", planes * 4, inplanes, 1, seed, gain=1.0)
                _frozen_bn(sd, p + "downsample.1", planes * 4, seed)
            inplanes = planes * 4



def _rest(sd, cfg: DTLRConfig, seed: int, d: int, C: int, ff: int, version: int = GENERATOR_VERSION):
    # ---- input_proj (models/dino/dino.py:115-136) ---------------------------------------------
    for l, cin in enumerate(cfg.backbone_channels):
        _conv(sd, f"input_proj.{l}.0.weight", d, cin, 1, seed, gain=1.0)
        sd[f"input_proj.{l}.0.bias"] = _normal(f"input_proj.{l}.0.bias", seed, (d,), 0.02)
        _norm(sd, f"input_proj.{l}.1", d, seed)
    l = len(cfg.backbone_channels)
    _conv(sd, f"input_proj.{l}.0.weight", d, cfg.backbone_channels[-1], 3, seed, gain=1.0)
    sd[f"input_proj.{l}.0.bias"] = _normal(f"input_proj.{l}.0.bias", seed, (d,), 0.02)
    _norm(sd, f"input_proj.{l}.1", d, seed)

    # ---- transformer ------------------------------------------------------------------------------
    t = "transformer."
    sd[t + "level_embed"] = _normal(t + "level_embed", seed, (cfg.num_feature_levels, d), 1.0)
    for n in range(cfg.enc_layers):
        p = f"{t}encoder.layers.{n}."
        _msda(sd, p + "self_attn", cfg, cfg.enc_n_points, seed)
        _norm(sd, p + "norm1", d, seed)
        _linear(sd, p + "linear1", ff, d, seed, gain=math.sqrt(2.0))
        _linear(sd, p + "linear2", d, ff, seed)
        _norm(sd, p + "norm2", d, seed)
    for n in range(cfg.dec_layers):
        p = f"{t}decoder.layers.{n}."
        _msda(sd, p + "cross_attn", cfg, cfg.dec_n_points, seed)
        _norm(sd, p + "norm1", d, seed)
        sd[p + "self_attn.in_proj_weight"] = _normal(p + "self_attn.in_proj_weight", seed, (3 * d, d), 1.0 / math.sqrt(d))
        sd[p + "self_attn.in_proj_bias"] = _normal(p + "self_attn.in_proj_bias", seed, (3 * d,), 0.02)
        _linear(sd, p + "self_attn.out_proj", d, d, seed)
        _norm(sd, p + "norm2", d, seed)
        _linear(sd, p + "linear1", ff, d, seed, gain=math.sqrt(2.0))
        _linear(sd, p + "linear2", d, ff, seed)
        _norm(sd, p + "norm3", d, seed)
        if version in (2, 4):
            for k in ("self_attn.out_proj.weight", "cross_attn.output_proj.weight", "linear2.weight"):
                sd[p + k] = sd[p + k] * 0.3                  # v2: damped residual branches (see GENERATOR_VERSION)
    _norm(sd, t + "decoder.norm", d, seed)
    _linear(sd, t + "decoder.ref_point_head.layers.0", d, 2 * d, seed, gain=math.sqrt(2.0))
    _linear(sd, t + "decoder.ref_point_head.layers.1", d, d, seed)
    code, q_cls, q_beta = _codes(cfg, seed, version)
    # a character query points along (its class code + 0.7 blank code): the blank part pushes every OTHER class down, as a trained
    # head does (sum of the sigmoids stays below 1, the blank channel of the decoders is 1 - sum); a blank query along the blank code
    kappa = 0.7
    q_code = np.where((q_cls >= 0)[:, None], (code[np.maximum(q_cls, 0)] + kappa * code[cfg.num_classes]) / math.sqrt(1 + kappa * kappa),
                      code[cfg.num_classes][None]).astype(np.float32)
    sd[t + "tgt_embed.weight"] = (_normal(t + "tgt_embed.weight", seed, (cfg.num_queries, d), 0.5)
                                  + torch.from_numpy(q_beta[:, None] * math.sqrt(d) * q_code))
    if version == 3:
        _prototype_memory(sd, cfg, code, q_cls, d, ff)
    if version == 4:
        # every content query the same vector; the characters come from the detector bank of the last decoder layer's FFN
        sd[t + "tgt_embed.weight"] = torch.from_numpy(np.tile((V4_BETA * math.sqrt(d) * code[cfg.num_classes])[None], (cfg.num_queries, 1)).astype(np.float32))
        cal = V4_CALIBRATION.get((cfg.num_classes, cfg.backbone, seed))
        if cal is not None:                                        # (None: the calibration pass itself -- detectors not installed yet)
            U, classes = v4_detectors(cfg, seed)
            p5 = f"{t}decoder.layers.{cfg.dec_layers - 1}."
            for k in range(V4_DETECTORS):
                sd[p5 + "linear1.weight"][k] = torch.from_numpy(U[k] / np.float32(cal["sigma"][k]))
                sd[p5 + "linear1.bias"][k] = -float(cal["theta"][k]) / float(cal["sigma"][k])
                sd[p5 + "linear2.weight"][:, k] = V4_GAIN * torch.from_numpy(code[classes[k]])
        elif not _V4_ALLOW_UNCALIBRATED:
            raise ValueError("generator v4 is calibrated for the Latin config (C = 166, resnet50), seed 0 only: run tools/calibrate_generator_v4.py "
                             "and add the constants to V4_CALIBRATION")
    _linear(sd, t + "enc_output", d, d, seed)
    _norm(sd, t + "enc_output_norm", d, seed)
    # Tokens whose proposal is invalid/padded have their memory row zeroed (models/dino/utils.py:
    # 58-62), so they all share ONE two-stage score.  With zero biases here that score is the bare
    # class bias (-6), i.e. below every real token, as a trained model arranges; otherwise the
    # top-k would contain a block of exactly tied scores whose order is implementation-defined
    # even inside the reference (CPU vs CUDA torch.topk).
    sd[t + "enc_output.bias"] = torch.zeros(d)
    sd[t + "enc_output_norm.bias"] = torch.zeros(d)
    # two-stage heads own their tensors (two_stage_*_embed_share=False, Latin_CTC.py:67-68)
    _linear(sd, t + "enc_out_bbox_embed.layers.0", d, d, seed, gain=math.sqrt(2.0))
    _linear(sd, t + "enc_out_bbox_embed.layers.1", d, d, seed, gain=math.sqrt(2.0))
    _linear(sd, t + "enc_out_bbox_embed.layers.2", 4, d, seed, gain=0.3)
    # The reference initialises class biases to -log(99) = -4.6 (models/dino/dino.py:164-166); a
    # TRAINED head is far more bimodal (most queries blank).  -6.0 with a wider weight spread gives
    # synthetic outputs where both branches of the blank decoder (sum p < 1-eps / >= 1-eps) and
    # the score>TH filter of the NMS decoder are exercised.
    cls_bias = -6.0
    _linear(sd, t + "enc_out_class_embed", C, d, seed, gain=1.0, bias_const=cls_bias)

    # ---- shared decoder heads: ONE MLP / ONE Linear aliased 2 x dec_layers times -----------------
    shared: Dict[str, torch.Tensor] = OrderedDict()
    _linear(shared, "bbox_embed.layers.0", d, d, seed, gain=math.sqrt(2.0))
    _linear(shared, "bbox_embed.layers.1", d, d, seed, gain=math.sqrt(2.0))
    _linear(shared, "bbox_embed.layers.2", 4, d, seed, gain=0.3)
    # v2 class head: row c = g * code_c - g_b * code_blank + noise; bias = the reference's -log(99)
    g_blank = 1.0
    # v3: per-class gains (a trained head is more confident about some characters than others) -> a spread of decision margins
    g_cls = 2.5 if version in (2, 4) else _rng("v3.class_gain", seed).uniform(1.4, 2.6, (C, 1)).astype(np.float32)
    shared["class_embed.weight"] = (torch.from_numpy(g_cls * code[:C] - g_blank * code[C:C + 1])
                                    + _normal("class_embed.weight", seed, (C, d), 0.25 / math.sqrt(d)))
    # bias: the reference's init -log(99) = -4.6 (dino.py:164-166), lowered by log(C / 166) for larger charsets so that the summed
    # sigmoid of the "other" classes stays below 1 as in a trained head (7356 classes: -8.4)
    shared["class_embed.bias"] = torch.full((C,), -4.6 - max(0.0, math.log(C / 166.0)))
    for n in range(cfg.dec_layers):
        for k, v in shared.items():
            head, rest = k.split(".", 1) if k.startswith("bbox_embed") else ("class_embed", k[len("class_embed."):])
            key = f"{head}.{n}.{rest}"
            sd[key] = v
            sd[t + "decoder." + key] = v
    sd["label_enc.weight"] = _normal("label_enc.weight", seed, (cfg.dn_labelbook_size + 1, d), 1.0)


# --------------------------------------------------------------------------------------------------
_ALIAS = re.compile(r"^(transformer\.decoder\.)?(class_embed|bbox_embed)\.(\d+)\.")


def load_checkpoint_state_dict(path: str, trust_pickle: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """`checkpoint["model"]` as evaluation.py:55-56 reads it; `module.` prefixes stripped
    (util/misc.py:581-586).

    Checkpoints are third-party files (the DTLR weights are Drive downloads): they are read with
    `weights_only=True`, which only admits tensors and plain containers.  The reference's own
    checkpoints also pickle an argparse `args` namespace (finetuning.py:709-721); that one global is
    allow-listed.  Anything else needs the explicit `trust_pickle=True` opt-in (env
    DTLR_TRUST_CHECKPOINT=1), which runs the unrestricted unpickler -- only for files you trust."""
    import argparse
    import os
    import pickle
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if not (trust_pickle or os.environ.get("DTLR_TRUST_CHECKPOINT") == "1"):
            raise RuntimeError(f"{path}: not loadable with weights_only=True ({e}); pass trust_pickle=True / set "
                               "DTLR_TRUST_CHECKPOINT=1 only if the file comes from a trusted source") from e
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    out = OrderedDict()
    for k, v in sd.items():
        out[k[7:] if k.startswith("module.") else k] = v
    return out


def num_classes_of(sd: Dict[str, torch.Tensor]) -> int:
    """Class count comes from the checkpoint, not a constant (SURVEY.md section 8 preamble)."""
    return int(sd["class_embed.0.weight"].shape[0])
