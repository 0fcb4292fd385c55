"""Model hyper-parameters of the DTLR inference path.

The reference keeps these in mmcv-style python config files (config/Latin_CTC.py:24-105,
config/Chinese.py:3, config/coco_transformer.py); only the keys the inference forward reads are
kept here, as data.  `DTLRConfig.from_reference_file` parses a reference config file (plain
assignments + `_base_`) so a user holding a reference checkout can point at it directly.
"""
from __future__ import annotations

import dataclasses
import os
from typing import List


@dataclasses.dataclass
class DTLRConfig:
    num_classes: int = 166            # config/Latin_CTC.py:3 ; Chinese.py:3 -> 7356
    hidden_dim: int = 256             # Latin_CTC.py:40
    nheads: int = 8                   # :42
    enc_layers: int = 6               # :35
    dec_layers: int = 6               # :36
    dim_feedforward: int = 2048       # :39
    num_feature_levels: int = 4       # :56
    enc_n_points: int = 4             # :57
    dec_n_points: int = 4             # :58
    num_queries: int = 900            # :43
    num_select: int = 300             # :72
    nms_iou_threshold: float = -1     # :94
    pe_temperatureH: int = 20         # :31
    pe_temperatureW: int = 20         # :32
    dn_labelbook_size: int = 167      # :105 (num_classes + 1)
    backbone: str = "resnet50"        # :26
    return_interm_indices: tuple = (1, 2, 3)   # :33
    two_stage_type: str = "standard"  # :64
    embed_init_tgt: bool = True       # :104
    dec_pred_class_embed_share: bool = True    # :97
    dec_pred_bbox_embed_share: bool = True     # :96
    two_stage_bbox_embed_share: bool = False   # :67
    two_stage_class_embed_share: bool = False  # :68
    decoder_sa_type: str = "sa"       # :91
    decoder_module_seq: tuple = ("sa", "ca", "ffn")  # :93
    transformer_activation: str = "relu"  # :73
    dropout: float = 0.0              # :41
    # bottleneck blocks per ResNet stage; (3,4,6,3) is resnet50, fewer only in tiny test configs
    backbone_blocks: tuple = (3, 4, 6, 3)
    # Swin backbones (models/dino/backbone.py:172-205 -> swin_transformer.py:683-720): the named variants below, or
    # "swin_custom" with swin_embed_dim / swin_depths / swin_num_heads / swin_window (test-sized networks)
    swin_embed_dim: int = 0
    swin_depths: tuple = ()
    swin_num_heads: tuple = ()
    swin_window: int = 0

    @property
    def head_dim(self) -> int:
        return self.hidden_dim // self.nheads

    SWIN_VARIANTS = {   # swin_transformer.py:686-717
        "swin_T_224_1k": dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=7),
        "swin_B_224_22k": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=7),
        "swin_B_384_22k": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=12),
        "swin_L_224_22k": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), window_size=7),
        "swin_L_384_22k": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), window_size=12),
    }

    @property
    def is_swin(self) -> bool:
        return self.backbone.startswith("swin_")

    def swin_params(self) -> dict:
        """embed_dim / depths / num_heads / window_size of the configured Swin backbone."""
        if self.backbone == "swin_custom":
            return dict(embed_dim=self.swin_embed_dim, depths=tuple(self.swin_depths), num_heads=tuple(self.swin_num_heads),
                        window_size=self.swin_window)
        return dict(self.SWIN_VARIANTS[self.backbone])

    @property
    def backbone_channels(self) -> List[int]:
        if self.is_swin:                                              # num_features[1:] (backbone.py:204, swin_transformer.py:521)
            e = self.swin_params()["embed_dim"]
            return [2 * e, 4 * e, 8 * e]
        return [512, 1024, 2048]   # layer2/3/4 outputs (models/dino/backbone.py:125-126)

    def validate(self) -> None:
        """Reject configurations outside the hot path."""
        if self.backbone != "resnet50" and not (self.backbone in self.SWIN_VARIANTS or self.backbone == "swin_custom"):
            raise NotImplementedError(f"backbone {self.backbone!r}: resnet50 (every shipped reference config, config/*.py:26) and the "
                                      "Swin variants of models/dino/backbone.py:172 are built; resnet101 / convnext are not")
        if self.is_swin:
            sp = self.swin_params()
            if len(sp["depths"]) != 4 or len(sp["num_heads"]) != 4 or sp["window_size"] <= 0 or sp["embed_dim"] <= 0:
                raise ValueError("swin backbone: 4 stages with depths / heads / window / embed_dim set")
            if any((sp["embed_dim"] << i) % h or (sp["embed_dim"] << i) // h != 32 for i, h in enumerate(sp["num_heads"])):
                raise NotImplementedError("swin backbone: head_dim must be 32 (true of every reference variant)")
        if self.two_stage_type != "standard" or not self.embed_init_tgt:
            raise NotImplementedError("only two_stage_type='standard' with embed_init_tgt=True")
        if tuple(self.decoder_module_seq) != ("sa", "ca", "ffn") or self.decoder_sa_type != "sa":
            raise NotImplementedError("decoder_module_seq must be ['sa','ca','ffn'] with sa_type 'sa'")
        if tuple(self.return_interm_indices) != (1, 2, 3) or self.num_feature_levels != 4:
            raise NotImplementedError("4 feature levels from layer2/3/4 + one extra stride-2 level")
        if self.transformer_activation != "relu" or self.dropout != 0.0:
            raise NotImplementedError("relu FFN without dropout")
        if not (self.dec_pred_class_embed_share and self.dec_pred_bbox_embed_share):
            raise NotImplementedError("decoder heads are shared across layers in every reference config")
        if self.hidden_dim != 256 or self.nheads != 8:
            raise NotImplementedError("hidden_dim=256 / nheads=8 (the reference decoder hard-codes "
                                      "128 sine frequencies, models/dino/utils.py:145)")

    # ---- presets -------------------------------------------------------------------------
    @staticmethod
    def latin() -> "DTLRConfig":
        return DTLRConfig()

    @staticmethod
    def chinese() -> "DTLRConfig":
        return DTLRConfig(num_classes=7356, dn_labelbook_size=7357)

    @staticmethod
    def tiny(num_classes: int = 23) -> "DTLRConfig":
        """A reduced network with the same topology, for KB-sized golden fixtures.  hidden_dim
        stays 256: the reference hard-codes 128 sine frequencies per coordinate
        (models/dino/utils.py:145) so its decoder only works at d_model=256."""
        return DTLRConfig(num_classes=num_classes, enc_layers=2, dec_layers=2,
                          dim_feedforward=512, num_queries=30, num_select=20,
                          dn_labelbook_size=num_classes + 1, backbone_blocks=(1, 1, 1, 1))

    @staticmethod
    def from_reference_file(path: str) -> "DTLRConfig":
        """Parse a reference config (`config/*.py`): plain assignments with `_base_` inheritance
        (util/slconfig.py:192-195 does this through addict/yapf, which the path does not need).
        Like the reference's loader this EXECUTES the file (configs are Python): only pass configs you trust."""
        def load(p, seen):
            ns: dict = {}
            with open(p) as f:
                exec(compile(f.read(), p, "exec"), ns)   # noqa: S102 - plain assignment files
            merged: dict = {}
            for b in ns.get("_base_", []) or []:
                bp = os.path.join(os.path.dirname(p), b)
                if bp not in seen:
                    merged.update(load(bp, seen | {bp}))
            merged.update({k: v for k, v in ns.items() if not k.startswith("_")})
            return merged
        raw = load(path, {path})
        fields = {f.name for f in dataclasses.fields(DTLRConfig)}
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in raw.items() if k in fields}
        cfg = DTLRConfig(**kw)
        cfg.validate()
        return cfg
